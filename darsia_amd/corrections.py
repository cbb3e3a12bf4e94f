"""Shape corrections on the GPU: ``BaseCorrection`` (corrections/basecorrection.py:15-139), the dense gather
``map_coordinates`` and ``CurvatureCorrection`` (corrections/shape/curvature.py:166-985), plus
``extract_quadrilateral_ROI`` (image/subregions.py:31-176) which the curvature set-up uses for its crop step.

What runs where: the per-image work of ``CurvatureCorrection`` is ``_transform_image`` -- one
``scipy.ndimage.map_coordinates(channel, grid, order=1)`` per channel over the cached coordinate grid
(curvature.py:921-957); that gather runs on the device for all channels in one pass and is bit-identical to
SciPy's (orders 0 and 1, float32 / float64 / uint8 / uint16).  The one-time set-up (``_transform_coordinates``:
a handful of float32 NumPy expressions over the pixel mesh) stays on the host exactly as in the reference, but its
intermediate gathers use the device as well.  The interactive calibration helpers (``pre_bulge_correction``,
``bulge_correction``, ``stretch_correction``, ``compute_bulge`` ...), TOML parsing and plotting are out of scope.
"""
from __future__ import annotations

import json
import math
from abc import ABC, abstractmethod
from pathlib import Path
from typing import Optional, Union
from warnings import warn

import numpy as np

from . import _lib
from .geometry import VoxelArray
from .image import Image  # noqa: F401
from .device import DeviceArray, host_empty
from .registration import _opencv_block_width, get_perspective_transform, warp_patches

_MC_DTYPES = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.uint8): 2, np.dtype(np.uint16): 3}


def map_coordinates(img, grid, order: int = 1, cval: float = 0.0):
    """``scipy.ndimage.map_coordinates(img[..., c], grid, order=order)`` (mode "constant") for every channel of an
    (H, W) or (H, W, C) array; returns (n,) or (n, C) in ``img``'s dtype.  ``img`` may be a ``DeviceArray`` (the result is
    one, too); ``grid`` is a (2, n) array of rows and columns or a pair of device arrays ``(rows, cols)`` -- a grid that
    is used for many frames is best kept on the device."""
    on_device = isinstance(img, DeviceArray)
    if not (isinstance(img, np.ndarray) or on_device) or img.ndim not in (2, 3):
        raise ValueError("img must be an (H, W) or (H, W, C) array")
    if np.dtype(img.dtype) not in _MC_DTYPES:
        raise TypeError(f"dtype {img.dtype} not supported on the device (float32, float64, uint8, uint16)")
    if order not in (0, 1):
        raise NotImplementedError("interpolation orders 0 and 1 are available on the device")
    if isinstance(grid, (tuple, list)) and len(grid) == 2 and all(isinstance(g, DeviceArray) for g in grid):
        rows, cols = grid
        if rows.dtype != cols.dtype or rows.dtype not in (np.float32, np.float64) or rows.size != cols.size:
            raise ValueError("device grids must be two float32 / float64 arrays of the same size")
        coord_dtype = rows.dtype
    else:
        grid = np.asarray(grid)
        if grid.ndim != 2 or grid.shape[0] != 2:
            raise ValueError("grid must have shape (2, n): rows, columns")
        if grid.dtype not in (np.float32, np.float64):
            grid = grid.astype(np.float64)
        rows, cols = np.ascontiguousarray(grid[0]), np.ascontiguousarray(grid[1])
        coord_dtype = grid.dtype
    _lib.require_gpu()
    src = img if on_device else np.ascontiguousarray(img)
    chans = 1 if src.ndim == 2 else src.shape[2]
    n = int(rows.size)
    out_shape = (n,) if src.ndim == 2 else (n, chans)
    out = DeviceArray(out_shape, src.dtype, src.device) if on_device else host_empty(out_shape, src.dtype)
    _lib.check(_lib.load().dxh_map_coordinates(
        src.device if on_device else _lib.default_device(), _MC_DTYPES[np.dtype(src.dtype)], src.shape[0], src.shape[1], chans,
        _lib.ptr(src), 0 if coord_dtype == np.float32 else 1, _lib.ptr(rows), _lib.ptr(cols), n, int(order), float(cval),
        _lib.ptr(out)))
    return out


def extract_quadrilateral_ROI(img_src: np.ndarray, pts_src=None, indexing: str = "reverse matrix",
                              interpolation: str = "inter_linear", **kwargs) -> np.ndarray:
    """Perspective extraction of a quadrilateral (image/subregions.py:31-176: ``cv2.getPerspectiveTransform`` +
    ``cv2.warpPerspective(img.astype(float32), P, (width, height), INTER_LINEAR)``, cast back to the input dtype).
    The warp runs on the device with OpenCV's float32 fixed-point bilinear arithmetic (unpinned)."""
    if interpolation not in ("inter_linear", "inter_area"):
        # "inter_area" never sets the flag in the reference (subregions.py:157-158 compares instead of assigning), so
        # OpenCV's default, INTER_LINEAR, is what it has always run with
        raise NotImplementedError("only 'inter_linear' is available on the device")
    scalar = img_src.ndim == 2
    original_shape = img_src.shape[:2]
    if "width" in kwargs and "height" in kwargs:
        aspect_ratio = kwargs["width"] / kwargs["height"]
        original_height, original_width = original_shape
        width = min(original_width, int(aspect_ratio * float(original_height)))
        height = min(original_height, int(1.0 / aspect_ratio * float(original_width)))
    else:
        height, width = kwargs.get("shape", original_shape)
    if pts_src is None:
        pts_src = np.array([[0, 0], [original_shape[0], 0], [original_shape[0], original_shape[1]], [0, original_shape[1]]])
        if indexing == "reverse matrix":
            pts_src = np.fliplr(pts_src)
    else:
        if isinstance(pts_src, VoxelArray):  # voxels use matrix indexing by definition
            pts_src = np.fliplr(np.asarray(pts_src))
        else:
            pts_src = np.asarray(pts_src)
            if indexing == "matrix":
                pts_src = np.fliplr(pts_src)
    if "pts_dst" in kwargs:
        pts_dst = kwargs["pts_dst"]
        if isinstance(pts_dst, VoxelArray):
            pts_dst = np.fliplr(np.asarray(pts_dst))
        else:
            pts_dst = np.asarray(pts_dst)
            if indexing == "matrix":
                pts_dst = np.fliplr(pts_dst)
    else:
        pts_dst = np.array([[0, 0], [0, height - 1], [width - 1, height - 1], [width - 1, 0]])
    assert pts_src.shape[0] == pts_dst.shape[0]
    p = get_perspective_transform(pts_src.astype(np.float32), pts_dst.astype(np.float32))
    chans = 1 if scalar else img_src.shape[2]
    if chans not in (1, 3):
        raise NotImplementedError("scalar or 3-channel images")
    h, w = original_shape
    rec = _lib.WarpPatch()
    rec.minv[:] = list(np.linalg.inv(p).ravel())
    rec.sr0, rec.sc0, rec.sh, rec.sw = 0, 0, h, w
    rec.ox, rec.oy, rec.w_eff, rec.h_eff = 0, 0, int(width), int(height)
    rec.nx, rec.ny = int(width), int(height)
    rec.bw0 = _opencv_block_width(int(width), int(height))
    src = img_src
    if isinstance(img_src, DeviceArray):
        # an image that is in HBM already is warped into a new device array; the result has the source's shape there,
        # so only a target of that shape (the default) stays device-resident
        if (int(height), int(width)) != (h, w):
            raise NotImplementedError("device-resident extract_quadrilateral_ROI: target shape = source shape")
        return warp_patches(img_src, [rec])
    if height > h or width > w:
        # the kernel writes a canvas of the source's shape: a target larger than the source gets both embedded in their
        # common bounding shape, the source rectangle (sh, sw) still ends where the image ends
        src = np.zeros((max(h, int(height)), max(w, int(width))) + img_src.shape[2:], img_src.dtype)
        src[:h, :w] = img_src
    return np.ascontiguousarray(warp_patches(src, [rec])[:height, :width])


class BaseCorrection(ABC):
    """Abstract correction (basecorrection.py:15-139): ``__call__(image, overwrite=False)`` on arrays and Images."""

    # corrections whose correct_array never writes into its argument set this and skip the defensive host copies
    # of basecorrection.py:38,44 (a 384 MB copy costs ~40 ms)
    _pure = False

    def __call__(self, image, overwrite: bool = False):
        if isinstance(image, DeviceArray):
            return self.correct_array(image)   # a device-resident frame: the correction returns a new device array
        if isinstance(image, np.ndarray):
            return self.correct_array(image if (overwrite or self._pure) else image.copy())
        if hasattr(image, "img") and hasattr(image, "metadata"):
            img = image.img if (overwrite or self._pure) else image.img.copy()
            if getattr(image, "series", False) and hasattr(self, "correct_array_series"):
                img = self.correct_array_series(img)
            elif getattr(image, "series", False):
                slices = [self.correct_array(image.img[..., t] if image.scalar else image.img[..., t, :])
                          for t in range(image.time_num)]
                img = np.stack(slices, axis=image.space_dim)
            else:
                img = self.correct_array(img)
            meta_update = self.correct_metadata(image.metadata())
            if overwrite:
                image.img = img
                if hasattr(image, "update_metadata"):
                    image.update_metadata(meta_update)
                else:
                    for k, v in meta_update.items():
                        setattr(image, k, v)
                return image
            meta = image.metadata()
            meta.update(meta_update)
            return type(image)(img, **meta)
        raise TypeError(f"Input type {type(image)} not supported.")

    @abstractmethod
    def correct_array(self, image: np.ndarray) -> np.ndarray:
        ...

    def correct_metadata(self, metadata: dict = {}) -> dict:
        return {}

    @abstractmethod
    def save(self, path) -> None:
        ...

    @abstractmethod
    def load(self, path) -> None:
        ...


class DeformationCorrection(BaseCorrection):
    """Local deformation correction: ``ImageRegistration`` against a baseline wrapped as a correction
    (corrections/shape/deformation.py:15-61).  ``config`` holds the registration's keywords (``N_patches``, ``rel_overlap``,
    ``max_features``, ``tol``, ``mask_dst``, ``verbosity`` ...) and ``active`` (default True); ``None`` switches the
    correction off and arrays pass through untouched.  As it stands the reference's class cannot be instantiated (it leaves
    BaseCorrection's abstract ``save`` / ``load`` undefined: TypeError); here they store / restore the configuration."""

    def __init__(self, base, config: Optional[dict]) -> None:
        self.base = base
        self._configure(config)

    def _configure(self, config: Optional[dict]) -> None:
        from .imageregistration import ImageRegistration

        self.config = None if config is None else dict(config)
        self.active = False if config is None else config.get("active", True)
        if self.active:
            self.image_registration = ImageRegistration(self.base, **config)

    def save(self, path) -> None:
        np.savez(path, class_name=type(self).__name__, config=np.array(self.config, dtype=object))

    def load(self, path) -> None:
        data = np.load(path, allow_pickle=True)
        if "config" not in data:
            raise ValueError("Invalid file format.")
        self._configure(data["config"].item())

    def correct_array(self, img: np.ndarray) -> np.ndarray:
        """The array registered onto the baseline; it is read with the baseline's metadata (deformation.py:55-58)."""
        if not self.active:
            return img
        return self.image_registration(Image(img, **self.base.metadata())).img


_CURVATURE_SECTIONS = {
    # section -> (key, default) the reference keeps (curvature.py:44-93 and 113-162); everything else is dropped
    "init": (("horizontal_bulge", 0.0), ("vertical_bulge", 0.0)),
    "crop": (("pts_src", None), ("width", 1.0), ("height", 1.0), ("in meters", True)),
    "bulge": (("horizontal_bulge", 0.0), ("horizontal_center_offset", 0), ("vertical_bulge", 0.0), ("vertical_center_offset", 0)),
    "stretch": (("horizontal_stretch", 0.0), ("horizontal_center_offset", 0), ("vertical_stretch", 0.0), ("vertical_center_offset", 0)),
}


def load_curvature_correction_config_from_dict(sec: dict, where: str = "config") -> dict:
    """The normal form the reference gives a dictionary configuration (curvature.py:97-164): only the four sections, only
    their known keys with the defaults filled in, and the crop corners as a ``VoxelArray`` -- i.e. ``pts_src`` of a dictionary
    (or TOML) configuration is read as (row, col), whereas a JSON file, which is taken as it stands, keeps
    ``extract_quadrilateral_ROI``'s (col, row) reading of plain lists."""
    from .geometry import make_voxel

    config: dict = {}
    for name, keys in _CURVATURE_SECTIONS.items():
        if name not in sec:
            warn(f"No 'curvature.{name}' section found in {where}.")
            continue
        if sec[name] is None:
            continue
        config[name] = {k: (make_voxel(sec[name].get(k, [])) if k == "pts_src" else sec[name].get(k, d)) for k, d in keys}
    return config


def load_curvature_correction_config_from_toml(path: Path) -> dict:
    """curvature.py:20-94: the ``[curvature]`` table of a TOML file, normalised as above."""
    try:
        import tomllib
    except ImportError:  # Python < 3.11
        raise NotImplementedError("TOML configurations need Python >= 3.11 (tomllib)") from None
    data = tomllib.loads(Path(path).read_text())
    if "curvature" not in data:
        warn(f"No 'curvature' section found in {path}.")
        return {}
    return load_curvature_correction_config_from_dict(data["curvature"], str(path))


class CurvatureCorrection(BaseCorrection):
    """Polynomial curvature correction (curvature.py:166-985): config sections ``init`` / ``crop`` / ``bulge`` /
    ``stretch`` applied in that order; the transformed coordinate grid is cached per object."""

    _pure = True

    def __init__(self, config: Optional[Union[dict, str, Path, list]] = None, **kwargs) -> None:
        self.setup_config(config)
        if "image" in kwargs:
            im_source = kwargs["image"]
            if not isinstance(im_source, np.ndarray):
                raise NotImplementedError("provide the reference image as an array (image files need OpenCV)")
            self.reference_image = im_source
            self.current_image = np.copy(im_source)
            self.dtype = self.current_image.dtype
            self.in_meters = kwargs.get("in_meters", True)
            self.width = kwargs.get("width", 1.0)
            self.height = kwargs.get("height", 1.0)
        else:
            warn("No image provided. Please provide an image or a config file.")
        self.resize_factor = kwargs.get("resize_factor", 1.0)
        if not math.isclose(self.resize_factor, 1.0):
            self._adapt_config()
        self.cache: dict = {}
        self.use_cache = bool(self.config.get("use_cache", False))
        if self.use_cache:
            self.cache_path = Path(self.config.get("cache", "./cache/curvature_transformation.npy"))
            self.cache_path.parents[0].mkdir(parents=True, exist_ok=True)
        self.interpolation_order: int = kwargs.get("interpolation_order", 1)

    def setup_config(self, config=None) -> None:
        """curvature.py:264-288: a dictionary is normalised, a JSON file is taken as it stands, a list of files is merged."""
        def read(path: Path) -> dict:
            if path.suffix == ".toml":
                return load_curvature_correction_config_from_toml(path)
            return json.loads(path.read_text())

        if isinstance(config, dict):
            self.config = load_curvature_correction_config_from_dict(config)
        elif isinstance(config, (str, Path)):
            self.config = read(Path(config))
        elif isinstance(config, list):
            self.config = {}
            for path in config:
                self.config.update(read(Path(path)))
        else:
            self.config = {}

    def write_config_to_file(self, path) -> None:
        """curvature.py:292-302."""
        with open(Path(path), "w") as outfile:
            json.dump(self.config, outfile, indent=4)

    def read_config_from_file(self, path) -> None:
        with open(Path(path), "r") as openfile:
            self.config = json.load(openfile)

    # ---- persistence (curvature.py:312-350: npz with the class name and the config) --------------------------
    def save(self, path) -> None:
        np.savez(path, class_name=type(self).__name__, config=self.config)

    def load(self, path) -> None:
        data = np.load(path, allow_pickle=True)
        if "config" not in data:
            raise ValueError("Invalid file format.")
        self.config = data["config"].item()
        self.cache = {}

    # ---- application ---------------------------------------------------------------------------------------
    def correct_array(self, img: np.ndarray) -> np.ndarray:
        """curvature.py:690-730."""
        if not (self.use_cache and self.cache_path.exists()) and "grid" not in self.cache:
            self._precompute_transformed_coordinates(img)
            if self.use_cache:
                np.save(self.cache_path, self.cache)
        elif self.use_cache and self.cache_path.exists():
            self.cache = np.load(self.cache_path, allow_pickle=True).item()
        return self._transform_image(img, self.cache["grid"], self.cache["shape"])

    def simple_curvature_correction(self, img: np.ndarray, **kwargs) -> np.ndarray:
        """curvature.py:734-771."""
        ny, nx = img.shape[:2]
        X, Y = np.meshgrid(np.arange(nx, dtype=np.float32), np.arange(ny, dtype=np.float32))
        X, Y = self._transform_coordinates(X, Y, **kwargs)
        grid = np.array([Y.ravel(), X.ravel()])
        return self._transform_image(img, grid, X.shape[:2])

    def _precompute_transformed_coordinates(self, img: np.ndarray) -> None:
        """curvature.py:775-825: the coordinate images pass through the configured steps themselves."""
        ny, nx = img.shape[:2]
        X, Y = np.meshgrid(np.arange(nx, dtype=np.float32), np.arange(ny, dtype=np.float32))
        coords = {"X": X, "Y": Y}
        for key, pixels in coords.items():
            if "init" in self.config:
                pixels = self.simple_curvature_correction(pixels, **self.config["init"])
            if "crop" in self.config:
                pixels = extract_quadrilateral_ROI(pixels, **self.config["crop"])
            if "bulge" in self.config:
                pixels = self.simple_curvature_correction(pixels, **self.config["bulge"])
            if "stretch" in self.config:
                pixels = self.simple_curvature_correction(pixels, **self.config["stretch"])
            coords[key] = pixels
        X, Y = coords["X"], coords["Y"]
        self.cache["grid"] = np.array([Y.ravel(), X.ravel()])
        self.cache["shape"] = X.shape[:2]

    def _adapt_config(self) -> None:
        """curvature.py:827-855."""
        for mainkey in ("init", "bulge"):
            if mainkey in self.config:
                for key in ("horizontal_bulge", "vertical_bulge", "horizontal_center_offset", "vertical_center_offset"):
                    if key in self.config[mainkey]:
                        self.config[mainkey][key] *= self.resize_factor
        if "crop" in self.config:
            self.config["crop"]["pts_src"] = (self.resize_factor * np.array(self.config["crop"]["pts_src"])).tolist()
        if "stretch" in self.config:
            for key in ("horizontal_stretch", "vertical_stretch", "horizontal_center_offset", "vertical_center_offset"):
                self.config["stretch"][key] *= self.resize_factor

    @staticmethod
    def _transform_coordinates(X: np.ndarray, Y: np.ndarray, **kwargs):
        """Bulge and stretch maps of the pixel mesh (curvature.py:857-919); float32 in, float32 out."""
        horizontal_bulge = kwargs.get("horizontal_bulge", 0.0)
        horizontal_stretch = kwargs.get("horizontal_stretch", 0.0)
        horizontal_center_offset = kwargs.get("horizontal_center_offset", 0)
        vertical_bulge = kwargs.get("vertical_bulge", 0.0)
        vertical_stretch = kwargs.get("vertical_stretch", 0.0)
        vertical_center_offset = kwargs.get("vertical_center_offset", 0)
        ny, nx = X.shape[:2]
        image_center = [round(nx / 2) + horizontal_center_offset, round(ny / 2) + vertical_center_offset]
        X -= image_center[0]
        Y -= image_center[1]
        x_lo, x_hi, y_lo, y_hi = np.min(X), np.max(X), np.min(Y), np.max(Y)
        Xmod = X + horizontal_bulge * np.multiply(X, (y_hi - Y) * (Y - y_lo)) + horizontal_stretch * X * (x_hi - X) * (X - x_lo)
        Ymod = Y + vertical_bulge * np.multiply(Y, (x_hi - X) * (X - x_lo)) + vertical_stretch * Y * (y_hi - Y) * (Y - y_lo)
        Xmod += image_center[0]
        Ymod += image_center[1]
        return Xmod, Ymod

    def _device_grid(self, grid: np.ndarray):
        """The cached coordinate grid as a pair of device arrays, uploaded once per grid (256 MB for 4000 x 8000)."""
        held = getattr(self, "_grid_dev", None)
        if held is None or held[0] is not grid:
            g = np.asarray(grid)
            if g.dtype not in (np.float32, np.float64):
                g = g.astype(np.float64)
            self._grid_dev = (grid, DeviceArray.from_host(np.ascontiguousarray(g[0])), DeviceArray.from_host(np.ascontiguousarray(g[1])))
        return self._grid_dev[1], self._grid_dev[2]

    def _transform_image(self, img, grid: np.ndarray, shape):
        """curvature.py:921-957 -- all channels in one device gather.  The object's cached grid stays on the device; a
        ``DeviceArray`` image gives a ``DeviceArray`` result."""
        dev_grid = self._device_grid(grid) if grid is self.cache.get("grid") else grid
        if isinstance(img, DeviceArray):
            img3 = img if img.ndim == 3 else img.reshape(*img.shape, 1)
            out = map_coordinates(img3, dev_grid, order=self.interpolation_order)
            out_shape = tuple(n for n in (*shape, img3.shape[2]) if n != 1) or (1,)
            return out.reshape(out_shape)
        img3 = np.atleast_3d(img)
        out = map_coordinates(np.ascontiguousarray(img3), dev_grid, order=self.interpolation_order)
        return np.squeeze(out.reshape(*shape, img3.shape[2]))

    def correct_metadata(self, metadata: dict = {}) -> dict:
        """curvature.py:959-985."""
        meta = {}
        crop = getattr(self, "config", {}).get("crop", {})
        if all(key in crop for key in ("width", "height")):
            meta["dimensions"] = [crop["height"], crop["width"]]
            meta["origin"] = np.array([0, crop["height"]])
        return meta
