"""Image registration façade on top of the GPU warp and thin-plate-spline kernels
(reference: multi_image_analysis/translationanalysis.py:20-833,
multi_image_analysis/imageregistration.py:17-306, 490-659).

What runs where:

* per-patch translation estimation -- injectable.  The reference uses ORB features + RANSAC
  homographies from OpenCV (corrections/shape/translation.py:36-107), which is sparse, OpenCV-internal
  work and out of scope.  Any object with the reference estimator's
  ``find_effective_translation(img_src, img_dst, roi_src, roi_dst, mask_src=, mask_dst=, plot_matches=)``
  method can be plugged in (``darsia.TranslationEstimator`` included).  The default here is a
  phase-correlation estimator that runs on the GPU for all patches of the image pair in one call
  (``dxh_phase_correlation``); it is NOT the reference's estimator and produces different translations.
* thin-plate-spline fit on ~N_patches + boundary points -- HOST, SciPy, as in the reference.
* dense evaluation of the displacement field and the piecewise perspective warp -- GPU
  (``darsia_amd.registration``).

Single-level (``DiffeomorphicImageRegistration``) and multilevel (``MultiscaleDiffeomorphicImageRegistration``,
imageregistration.py:309-484: single levels composed with ``add`` / ``deduct``) registration and the ``ImageRegistration`` façade.
Images whose pixels are a ``DeviceArray`` stay in HBM through estimator and warp.
"""
from __future__ import annotations

import logging
from typing import Optional, Union

import numpy as np
from scipy.interpolate import RBFInterpolator

from . import _lib
from .image import Image
from .registration import Patches, PiecewisePerspectiveTransform, evaluate_rbf

logger = logging.getLogger(__name__)

_GPU_EVAL_THRESHOLD = 4096  # below this many points SciPy on the host is faster than a launch


_PC_DTYPES = {np.dtype(np.float32): _lib.DXH_F32, np.dtype(np.float64): _lib.DXH_F64, np.dtype(np.uint8): 2, np.dtype(np.uint16): 3}


class PhaseCorrelationEstimator:
    """Whole-patch translation by FFT phase correlation with a parabolic sub-pixel fit, on the device
    (``dxh_phase_correlation``; the algorithm is stated in oracle/phase_correlation_numpy.py).  Stand-in for
    ``darsia.TranslationEstimator`` (ORB + RANSAC, OpenCV) with the same call signature; ``find_effective_translations`` takes
    all patches of an image pair in one call, which is what ``TranslationAnalysis`` uses when an estimator offers it."""

    def __init__(self, max_shift_fraction: float = 0.25, min_peak_ratio: float = 4.0, min_active_fraction: float = 0.5) -> None:
        """``min_peak_ratio``: an estimate is kept only if its correlation peak reaches this multiple of the level unrelated
        patterns reach (``dxh_phase_correlation_q``): flat, noise-only and non-finite patches are reported as not intact, as
        the reference's feature matcher does, and the spline interpolates over them.  ``min_active_fraction``: with masks, a
        patch is estimated only if at least this fraction of its pixels is active in BOTH masks (the reference restricts
        the matched pixels to the masks; here a mostly masked patch is left to the spline instead)."""
        self.max_shift_fraction = max_shift_fraction
        self.min_peak_ratio = min_peak_ratio
        self.min_active_fraction = min_active_fraction
        self.last_peaks = None      # correlation peak per patch of the last call (diagnostics)

    def find_effective_translations(self, img_src, img_dst, rois, masks_src=None, masks_dst=None) -> list:
        """``[(translation, intact)]`` per patch; ``rois``: pairs of slices or rows (r0, r1, c0, c1) -- the same rectangle of both
        images; ``masks_*``: full-size boolean arrays, a patch without a single active pixel in either is not estimated."""
        from .device import DeviceArray

        on_device = isinstance(img_src, DeviceArray) or isinstance(img_dst, DeviceArray)
        if on_device:
            # frames that are in HBM already are read where they lie; a host partner of the same dtype is uploaded by the call
            a, b = (x if isinstance(x, DeviceArray) else np.ascontiguousarray(x) for x in (img_src, img_dst))
            if np.dtype(a.dtype) != np.dtype(b.dtype) or np.dtype(a.dtype) not in _PC_DTYPES:
                raise TypeError("device-resident frames must share one of the dtypes float32 / float64 / uint8 / uint16")
        else:
            a, b = np.asarray(img_src), np.asarray(img_dst)
        boxes = np.array([[r[0].start, r[0].stop, r[1].start, r[1].stop] if isinstance(r[0], slice) else list(r) for r in rois],
                         dtype=np.int32).reshape(-1, 4)
        if len(boxes) == 0:
            return []
        if a.shape != b.shape or a.ndim not in (2, 3) or (a.ndim == 3 and a.shape[2] != 3):
            return [(None, False)] * len(boxes)
        boxes[:, 1] = np.minimum(boxes[:, 1], a.shape[0])
        boxes[:, 3] = np.minimum(boxes[:, 3], a.shape[1])
        if not on_device:
            if a.dtype != b.dtype or a.dtype not in _PC_DTYPES:
                a, b = a.astype(np.float64), b.astype(np.float64)
            a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
        _lib.require_gpu()
        n = len(boxes)
        shifts = np.empty((n, 2), np.float64)
        ok = np.empty(n, np.int32)
        device = a.device if isinstance(a, DeviceArray) else (b.device if isinstance(b, DeviceArray) else _lib.default_device())
        peaks = np.empty(n, np.float64)
        _lib.check(_lib.load().dxh_phase_correlation_q(device, a.shape[0], a.shape[1], 1 if a.ndim == 2 else 3,
                                                        _PC_DTYPES[np.dtype(a.dtype)], _lib.ptr(a), _lib.ptr(b), n, _lib.ptr(boxes),
                                                        float(self.max_shift_fraction), float(self.min_peak_ratio),
                                                        _lib.ptr(shifts), _lib.ptr(ok), _lib.ptr(peaks)))
        self.last_peaks = peaks

        def enough(mask, r0, r1, c0, c1):
            m = mask[r0:r1, c0:c1]
            need = self.min_active_fraction * m.size
            return np.count_nonzero(m) >= max(need, 1)

        out = []
        for k, (r0, r1, c0, c1) in enumerate(boxes):
            active = ok[k] and bool(np.isfinite(shifts[k]).all()) and (
                masks_src is None or masks_dst is None or (enough(masks_src, r0, r1, c0, c1) and enough(masks_dst, r0, r1, c0, c1)))
            out.append((np.array([[1.0, 0.0, shifts[k, 0]], [0.0, 1.0, shifts[k, 1]]]), True) if active else (None, False))
        return out

    def find_effective_translation(self, img_src, img_dst, roi_src=None, roi_dst=None, mask_src=None, mask_dst=None,
                                   plot_matches: bool = False):
        a = np.asarray(img_src)
        return self.find_effective_translations(img_src, img_dst, [(0, a.shape[0], 0, a.shape[1])], mask_src, mask_dst)[0]


class TranslationAnalysis:
    """Patch-wise translation between a test image and a baseline image, interpolated to a continuous
    displacement map and applied as a piecewise perspective warp (translationanalysis.py:20-833)."""

    def __init__(self, base, N_patches, rel_overlap: float, translation_estimator, mask=None) -> None:
        self.N_patches = list(N_patches)
        self.rel_overlap = rel_overlap
        self.update_base(base)
        self.translation_estimator = translation_estimator
        self.translation = lambda arg: np.zeros((2, np.atleast_2d(arg).shape[0]))
        self.have_translation = np.zeros(tuple(self.N_patches), dtype=bool)
        # masks nobody gave are all True: they are made when somebody looks at them (32 MB and 3 ms per 4000 x 8000 image)
        self._mask_base = None if mask is None else mask.copy()
        self._all_true_base = None
        self.interpolator_translation_x = None
        self.interpolator_translation_y = None

    # ---- masks (translationanalysis.py:60-75: all True unless given) ----------------------------------
    def _is_all_true(self, which: str) -> bool:
        held = getattr(self, "_mask_" + which, None)
        return held is None or held.img is getattr(self, "_all_true_" + which, None)

    def _default_mask(self, which: str, like):
        mask = Image(np.ones(like.img.shape[:2], dtype=bool), dimensions=like.dimensions)
        setattr(self, "_mask_" + which, mask)
        setattr(self, "_all_true_" + which, mask.img)
        return mask

    @property
    def mask_base(self):
        return self._mask_base if self._mask_base is not None else self._default_mask("base", self.base)

    @mask_base.setter
    def mask_base(self, mask) -> None:
        self._mask_base, self._all_true_base = mask, None

    @property
    def mask_img(self):
        return self._mask_img if self._mask_img is not None else self._default_mask("img", self.img)

    @mask_img.setter
    def mask_img(self, mask) -> None:
        self._mask_img, self._all_true_img = mask, None

    # ---- set-up ----------------------------------------------------------------------------------
    def update_params(self, N_patches=None, rel_overlap=None) -> None:
        changed = False
        if N_patches is not None and list(N_patches) != self.N_patches:
            self.N_patches, changed = list(N_patches), True
        if rel_overlap is not None and rel_overlap != self.rel_overlap:
            self.rel_overlap, changed = rel_overlap, True
        if changed:
            self.update_base_patches()

    def update_base(self, base) -> None:
        self.base = base
        self._drop_device_base()
        self.update_base_patches()

    # ---- the baseline's pixels in HBM, for the device estimator ---------------------------------------
    def _drop_device_base(self) -> None:
        held = getattr(self, "_base_dev", None)
        if held is not None:
            held[1].close()
        self._base_dev = None

    def _stages_on_device(self, img) -> bool:
        """Host frames the device estimator and the warp take as stored: worth one upload for both."""
        a, b = img.img, self.base.img
        return (isinstance(self.translation_estimator, PhaseCorrelationEstimator) and isinstance(a, np.ndarray)
                and isinstance(b, np.ndarray) and a.dtype == b.dtype and a.dtype in _PC_DTYPES and a.shape == b.shape
                and a.ndim in (2, 3) and (a.ndim == 2 or a.shape[2] == 3))

    def refresh(self) -> None:
        """Drop the baseline's device copy (after an in-place edit of ``base.img``; ``update_base`` does it by itself)."""
        self._drop_device_base()

    def _device_base(self):
        """The baseline's pixels as a device array, uploaded once and kept while the host array is the same object with the
        same fingerprint (``concentration._fingerprint``: a safety net -- in-place edits are outside the contract, see
        ``refresh``).  ``self.base`` keeps the array alive, so its id() cannot be recycled under the mark."""
        from .concentration import _fingerprint
        from .device import DeviceArray

        mark = (id(self.base.img), _fingerprint(self.base.img))
        held = getattr(self, "_base_dev", None)
        if held is None or held[0] != mark:
            self._drop_device_base()
            self._base_dev = held = (mark, DeviceArray.from_host(self.base.img))
        return held[1]

    def update_base_patches(self) -> None:
        self.patches_base = Patches(self.base, self.N_patches, rel_overlap=self.rel_overlap)

    def load_image(self, img, mask=None) -> None:
        self.img = img
        self._mask_img = None if mask is None else mask.copy()
        self._all_true_img = None     # the array known to be all True, if a default mask was made and is still in place

    # ---- boundary conditions (translationanalysis.py:310-404, pixel units) ----------------------------
    def bc_x(self, units=("pixel", "pixel")):
        assert units[0] == "pixel", "metric units are not mirrored"
        ys = np.linspace(0, self.base.num_voxels[0], self.N_patches[0] + 1)
        pts = [np.array([0, y]) for y in ys] + [np.array([self.base.num_voxels[1], y]) for y in ys]
        return pts, len(pts) * [0.0]

    def bc_y(self, units=("pixel", "pixel")):
        assert units[0] == "pixel", "metric units are not mirrored"
        # NOTE: the reference places the bottom boundary at y = num_voxels[1] (the image WIDTH),
        # translationanalysis.py:394-398; kept.
        pts = [np.array([x, self.base.num_voxels[1]]) for x in np.linspace(0, self.base.num_voxels[1], self.N_patches[1] + 1)]
        return pts, len(pts) * [0.0]

    def _patch_centers_pixel(self):
        """Patch centres as (x, y) pixels: metric centre -> voxel (floor) -> flipped (translationanalysis.py:249-255)."""
        return np.asarray(self.patches_base.global_centers_voxels)[..., ::-1].astype(float)   # Patches holds exactly these voxels

    def _set_interpolators(self, coords, tx, ty) -> None:
        ex, etx = self.bc_x()
        ey, ety = self.bc_y()
        self.interpolator_translation_x = RBFInterpolator(coords + ex, tx + etx)
        self.interpolator_translation_y = RBFInterpolator(coords + ey, ty + ety)

        def translation_callable(arg):
            arg = np.atleast_2d(np.asarray(arg, dtype=np.float64))
            if arg.shape[0] >= _GPU_EVAL_THRESHOLD:
                return np.array([evaluate_rbf(self.interpolator_translation_x, arg),
                                 evaluate_rbf(self.interpolator_translation_y, arg)])
            return np.array([self.interpolator_translation_x(arg), self.interpolator_translation_y(arg)])

        self.translation = translation_callable

    # ---- analysis -----------------------------------------------------------------------------------
    def find_translation(self, units=("pixel", "pixel")):
        """translationanalysis.py:149-308 (pixel units)."""
        assert list(units) == ["pixel", "pixel"], "metric units are not mirrored"
        batch = getattr(self.translation_estimator, "find_effective_translations", None)
        same_grid = self.img.img.shape[:2] == self.base.img.shape[:2] and list(self.img.dimensions) == list(self.base.dimensions)
        # the patch rectangles depend on shape, dimensions and the patch grid only: the baseline's serve an image of its geometry
        patches_img = self.patches_base if (batch is not None and same_grid) else Patches(self.img, self.N_patches, rel_overlap=self.rel_overlap)
        if batch is None:
            patches_mask_base = Patches(self.mask_base, self.N_patches, rel_overlap=self.rel_overlap)
            patches_mask_img = Patches(self.mask_img, self.N_patches, rel_overlap=self.rel_overlap)
        have = np.zeros(tuple(self.N_patches), dtype=bool)
        centers = self._patch_centers_pixel()
        coords, tx, ty = [], [], []
        if batch is not None:
            # all patches of the pair in one call (the device estimator): same rectangles, same masks as the loop below
            from .device import DeviceArray

            # a frame staged in HBM (__call__) meets the baseline there: neither crosses PCIe again
            base_pixels = self._device_base() if isinstance(self.img.img, DeviceArray) and self._stages_on_device(self.base) else self.base.img
            flat = batch(self.img.img, base_pixels, [patches_img.rois[i][j] for i in range(self.N_patches[0]) for j in range(self.N_patches[1])],
                         *((None, None) if (self._is_all_true("img") and self._is_all_true("base"))
                           else (self.mask_img.img, self.mask_base.img)))
        for i in range(self.N_patches[0]):
            for j in range(self.N_patches[1]):
                if batch is not None:
                    translation, intact = flat[i * self.N_patches[1] + j]
                else:
                    translation, intact = self.translation_estimator.find_effective_translation(
                        patches_img(i, j).img, self.patches_base(i, j).img, None, None,
                        mask_src=patches_mask_img(i, j).img, mask_dst=patches_mask_base(i, j).img, plot_matches=False)
                if intact:
                    have[i, j] = True
                    disp = np.asarray(translation)[:, -1]
                    coords.append(centers[i, j])
                    tx.append(float(disp[0]))
                    ty.append(float(disp[1]))
        self._set_interpolators(coords, tx, ty)
        self.have_translation = have.copy()
        return self.translation, bool(have.any())

    def translate_image(self, reverse: bool = True):
        """translationanalysis.py:616-646: piecewise perspective warp of the loaded image (GPU)."""
        # the zero-overlap patch grid depends on shape, dimensions, origin and patch counts only: one per geometry, re-pointed at
        # the image to warp (the grid holds no pixels)
        import copy as _copy

        key = (tuple(self.img.img.shape[:2]), tuple(self.img.dimensions), tuple(np.asarray(self.img.origin, dtype=float)), tuple(self.N_patches))
        held = getattr(self, "_warp_grid", None)
        if held is None or held[0] != key:
            grid = Patches(self.img, self.N_patches, rel_overlap=0.0)
            grid.base = None                      # geometry only: the kept grid must not pin an image
            self._warp_grid = held = (key, grid)
        patches = _copy.copy(held[1])
        patches.base = self.img
        return PiecewisePerspectiveTransform().find_and_warp(patches, self.translation, reverse)

    def __call__(self, img, reverse: bool = False, mask=None):
        if not self._stages_on_device(img):
            self.load_image(img, mask)
            self.find_translation()
            return self.translate_image(reverse)
        # a host frame the device stages take as stored: ONE upload serves the estimator and the warp, one download brings
        # the result back (otherwise the frame crosses PCIe once per stage)
        from .device import DeviceArray
        from .image import with_array

        staged = with_array(img, DeviceArray.from_host(img.img))
        try:
            self.load_image(staged, mask)
            self.find_translation()
            warped = self.translate_image(reverse)
            out = with_array(warped, warped.img.get())
            warped.img.close()
        finally:
            self.img = img          # the caller's image is the loaded one, as after a plain load_image
            staged.img.close()
        return out

    def return_patch_translation(self, reverse: bool = True, units: str = "metric") -> np.ndarray:
        """Deformation at the patch centres in (row, col) components, metric or pixel units, shape (*N_patches, 2)
        (translationanalysis.py:406-447)."""
        patch_centers = self.patches_base.global_centers_voxels
        shape = patch_centers.shape
        t = np.array(self.translation(np.fliplr(patch_centers.reshape((-1, 2)))), dtype=float)
        if reverse:
            t *= -1.0
        patch_translation = np.vstack((t[1], t[0])).T   # standard matrix indexing
        if units == "metric":
            patch_translation = self.base.coordinatesystem.coordinate_vector(patch_translation)
        return patch_translation.reshape(shape)

    # ---- composition (translationanalysis.py:676-833) ----------------------------------------------
    def deduct_translation_analysis(self, other) -> None:
        centers = self._patch_centers_pixel().reshape(-1, 2)
        d = other.translation(centers)
        self._set_interpolators([c for c in centers], [float(v) for v in d[0]], [float(v) for v in d[1]])

    def add_translation_analysis(self, other) -> None:
        centers = self._patch_centers_pixel().reshape(-1, 2)
        d0 = self.translation(centers)
        displaced = centers + np.stack([d0[0], d0[1]], axis=-1)
        d1 = other.translation(displaced)
        total = d0 + d1
        self._set_interpolators([c for c in centers], [float(v) for v in total[0]], [float(v) for v in total[1]])


class DiffeomorphicImageRegistration:
    """Single-level registration (imageregistration.py:17-306)."""

    def __init__(self, img_dst, **kwargs) -> None:
        self.translation_estimator = kwargs.get("translation_estimator") or PhaseCorrelationEstimator()
        self.N_patches = kwargs.get("N_patches", [1, 1])
        self.rel_overlap = kwargs.get("rel_overlap", 0.0)
        self.dst = img_dst
        self.translation_analysis = TranslationAnalysis(
            img_dst, N_patches=self.N_patches, rel_overlap=self.rel_overlap,
            translation_estimator=self.translation_estimator, mask=kwargs.get("mask_dst", None))

    def update_dst(self, img_dst) -> None:
        self.dst = img_dst
        self.translation_analysis.update_base(img_dst)

    def deduct(self, other: "DiffeomorphicImageRegistration") -> None:
        self.translation_analysis.deduct_translation_analysis(other.translation_analysis)

    def add(self, other: "DiffeomorphicImageRegistration") -> None:
        self.translation_analysis.add_translation_analysis(other.translation_analysis)

    def __call__(self, img, mask=None, return_transformed_dst: bool = False):
        transformed = self.translation_analysis(img, mask=mask)
        if return_transformed_dst:
            return transformed, self.apply(self.dst, reverse=True)
        return transformed

    def call_with_output(self, img, plot_patch_translation: bool = False, return_patch_translation: bool = False,
                         mask=None):
        """Register ``img`` and optionally return the deformation at the patch centres, in the sense img -> dst
        (imageregistration.py:127-164).  Plotting is not mirrored (matplotlib, out of scope)."""
        if plot_patch_translation:
            raise NotImplementedError("plot_translation is not mirrored")
        transformed = self.translation_analysis(img, mask=mask)
        if return_patch_translation:
            return transformed, self.translation_analysis.return_patch_translation(False)
        return transformed

    def evaluate(self, coords: Union[np.ndarray, Patches], reverse: bool = False, units: str = "metric") -> np.ndarray:
        """Deformation vectors at arbitrary points (imageregistration.py:166-217)."""
        base = self.translation_analysis.base
        if isinstance(coords, Patches):
            coords = coords.global_centers_cartesian
        assert coords.shape[-1] == 2
        shape = coords.shape
        coords = coords.reshape(-1, 2)
        if units == "metric":
            pixel_coords = np.fliplr(np.asarray(base.coordinatesystem.voxel(coords)))
        elif units == "pixel":
            pixel_coords = coords
        else:
            raise ValueError
        translation = self.translation_analysis.translation(pixel_coords)
        if reverse:
            translation = -1.0 * translation
        displacement = np.transpose(np.vstack((translation[1], translation[0])))
        if units == "metric":
            displacement = base.coordinatesystem.coordinate_vector(displacement)
        return displacement.reshape(shape)

    def apply(self, img, reverse: bool = False):
        self.translation_analysis.load_image(img)
        return self.translation_analysis.translate_image(reverse)

    def displacement(self) -> np.ndarray:
        """Displacement in metric units on all pixels, shape (H*W, 2) (imageregistration.py:265-306).
        The two thin-plate splines are evaluated on the GPU over the pixel grid."""
        from .registration import evaluate_rbf_pair_on_grid

        ta = self.translation_analysis
        if ta.interpolator_translation_x is None:
            raise RuntimeError("no translation has been determined yet")
        ny, nx = ta.base.img.shape[:2]
        # the reference stacks (ty, tx) in (row, col) order and converts with coordinate_vector, which gives
        # (x, y) = (tx * voxel width, -ty * voxel height): both factors are applied as the field is written
        vs = ta.base.coordinatesystem.voxel_size
        return evaluate_rbf_pair_on_grid(ta.interpolator_translation_x, ta.interpolator_translation_y, (ny, nx),
                                         out_scale=(vs["x"], -vs["y"]))


class MultiscaleDiffeomorphicImageRegistration:
    """Multilevel registration for larger deformations (imageregistration.py:309-484): every level
    registers the CURRENTLY warped image against the destination on its own patch grid; the levels are
    composed into one translation map (level 0: deduct, later levels: add) and the ORIGINAL image is
    re-warped with the combined map."""

    def __init__(self, img_dst, config, mask_dst=None, total_config=None, **kwargs) -> None:
        assert isinstance(config, list) and all(isinstance(c, dict) for c in config)
        self.img_dst = img_dst
        self.dst = img_dst
        self.mask_dst = mask_dst
        self.config = config
        self.num_levels = len(config)
        self.total_config = config[-1] if total_config is None else total_config
        self.verbosity = kwargs.get("verbosity", 0)

    def _level_registration(self, level: int) -> "DiffeomorphicImageRegistration":
        """The single-level registration of a level against the (fixed) destination.  The reference builds a new one per call
        (imageregistration.py:382-398); nothing in it depends on the image to register, so it is kept -- with its patch grid and
        the device copy of the destination's pixels."""
        if not hasattr(self, "_levels"):
            self._levels = {}
        if level not in self._levels:
            self._levels[level] = DiffeomorphicImageRegistration(self.img_dst, mask_dst=self.mask_dst, **self.config[level])
        return self._levels[level]

    def _single_level_iteration(self, img, mask, config):
        registration = DiffeomorphicImageRegistration(self.img_dst, mask_dst=self.mask_dst, **config)
        transformed, _ = registration.call_with_output(img, return_patch_translation=True, mask=mask)
        return transformed, registration

    def __call__(self, img, mask=None, return_transformed_dst: bool = False):
        """imageregistration.py:400-452.  Same maps and same result as the reference's loop; what differs is where the pixels
        are: a host image the device stages take as stored is uploaded once, every level's estimator and every re-warp of the
        original read it in HBM, and only the final image comes back.  (The warped image a level's own registration would
        return is never used by the reference either: only its translation map is, so only that is computed.)"""
        from .device import DeviceArray
        from .image import with_array

        first = self._level_registration(0).translation_analysis
        staged = with_array(img, DeviceArray.from_host(img.img)) if first._stages_on_device(img) else None
        work = img if staged is None else staged
        try:
            transformed_img = work
            transformed_mask = mask.copy() if mask is not None else None
            self.combined_image_registration = DiffeomorphicImageRegistration(work, **self.total_config)
            for level in range(self.num_levels):
                registration = self._level_registration(level)
                registration.translation_analysis.load_image(transformed_img, transformed_mask)
                registration.translation_analysis.find_translation()
                if level == 0:
                    self.combined_image_registration.deduct(registration)
                else:
                    self.combined_image_registration.add(registration)
                if transformed_mask is not None:
                    warped = registration.apply(Image(transformed_mask.img.astype(np.float32), **transformed_mask.metadata()))
                    transformed_mask = Image(warped.img > 0.5, **transformed_mask.metadata())
                previous = transformed_img
                transformed_img = self.combined_image_registration.apply(work)
                if staged is not None and previous is not work:
                    previous.img.close()
                registration.translation_analysis.img = img   # the level does not keep a staged frame alive
            if staged is not None:
                result = with_array(transformed_img, transformed_img.img.get())
                transformed_img.img.close()
                transformed_img = result
        finally:
            if staged is not None:
                # the combined map's geometry comes from the image: point it back at the caller's (same shape, same metadata)
                ta = self.combined_image_registration.translation_analysis
                ta.base = ta.img = ta.patches_base.base = self.combined_image_registration.dst = img
                staged.img.close()
        if return_transformed_dst:
            return transformed_img, self.combined_image_registration.apply(self.dst, reverse=True)
        return transformed_img

    def apply(self, img, reverse: bool = False):
        if not hasattr(self, "combined_image_registration"):
            raise ValueError("Construct the deformation first.")
        return self.combined_image_registration.apply(img, reverse)

    def evaluate(self, coords, reverse: bool = False, units: str = "metric"):
        return self.combined_image_registration.evaluate(coords, reverse, units)

    def displacement(self):
        return self.combined_image_registration.displacement()


class ImageRegistration:
    """Façade (imageregistration.py:490-659): one level for a flat ``N_patches``, multilevel for a list of patch grids.
    Keyword handling as in the reference: ``max_features`` (int), ``tol`` (float) and ``rel_overlap`` (float) are broadcast
    over the levels or given as lists of one entry per level -- anything else is "not compatible"; with
    ``method="onelevel"`` the first entry of every list is taken.  ``translation_estimator`` (not in the reference, whose
    estimator is OpenCV's ORB / RANSAC) injects the per-patch estimator, one object or one per level."""

    def __init__(self, img_dst, method: Optional[str] = None, **kwargs) -> None:
        assert method in [None, "multilevel", "onelevel"]
        n_patches = kwargs.get("N_patches", [1, 1])
        max_features = kwargs.get("max_features", 200)
        tol = kwargs.get("tol", 0.05)
        rel_overlap = kwargs.get("rel_overlap", 0.0)
        mask_dst = kwargs.get("mask_dst", None)
        verbosity = kwargs.get("verbosity", 0)
        estimator = kwargs.get("translation_estimator")
        nested = isinstance(n_patches, list) and all(isinstance(n, list) for n in n_patches)
        if method is None:
            method = "multilevel" if nested else "onelevel"
        self.method = method
        if method == "multilevel":
            if not isinstance(n_patches, list):
                n_patches = [n_patches]
            levels = len(n_patches)
            compatible = True
            if isinstance(max_features, int):
                max_features = levels * [max_features]
            else:
                compatible = compatible and isinstance(max_features, list) and len(max_features) == levels
            if isinstance(tol, float):
                tol = levels * [tol]
            else:
                compatible = compatible and isinstance(tol, list) and len(tol) == levels
            if isinstance(rel_overlap, float):
                rel_overlap = levels * [rel_overlap]
            else:
                compatible = compatible and isinstance(rel_overlap, list) and len(rel_overlap) == levels
            if not compatible:
                raise ValueError("Input for the multilevel image registration is not compatible.")
            estimators = estimator if isinstance(estimator, list) else levels * [estimator]
            if len(estimators) != levels:
                raise ValueError("translation_estimator must be one object or one per level")
            config = [{"N_patches": n_patches[i], "max_features": max_features[i], "tol": tol[i], "rel_overlap": rel_overlap[i],
                       "translation_estimator": estimators[i]} for i in range(levels)]
            self.config = config
            self.image_registration = MultiscaleDiffeomorphicImageRegistration(img_dst, config, mask_dst=mask_dst, verbosity=verbosity)
        else:
            if nested:
                n_patches = n_patches[0]
            if isinstance(max_features, list):
                max_features = max_features[0]
            if isinstance(tol, list):
                tol = tol[0]
            if isinstance(rel_overlap, list):
                rel_overlap = rel_overlap[0]
            if isinstance(estimator, list):
                estimator = estimator[0]
            config = {"N_patches": n_patches, "max_features": max_features, "tol": tol, "rel_overlap": rel_overlap,
                      "mask_dst": mask_dst, "verbosity": verbosity, "translation_estimator": estimator}
            self.config = config
            self.image_registration = DiffeomorphicImageRegistration(img_dst, **config)
        self.registration = self.image_registration   # earlier name, kept

    def __call__(self, img, mask=None, return_transformed_dst: bool = False):
        return self.registration(img, mask=mask, return_transformed_dst=return_transformed_dst)

    def apply(self, img, reverse: bool = False):
        return self.registration.apply(img, reverse)

    def evaluate(self, coords, reverse: bool = False, units: str = "metric"):
        return self.registration.evaluate(coords, reverse, units)

    def displacement(self):
        return self.registration.displacement()
