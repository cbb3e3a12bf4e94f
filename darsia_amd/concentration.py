"""``ConcentrationAnalysis`` with its elementwise stages fused around the device-resident TV
restoration (reference: multi_image_analysis/concentrationanalysis.py:20-421,
signals/reduction/monochromatic.py:12-95, signals/models/linearmodel.py:14-149,
clipmodel.py:10-106, combinedmodel.py:13-52).

A frame goes to the GPU once (colour), ``diff -> reduce -> clean -> [model] -> restoration -> [model]``
run there, and the concentration comes back once (scalar).  The fused path needs stages it can
express on the device:

* difference: the four ``diff option`` values;
* reduction: ``MonochromaticReduction`` with colour "red", "green", "blue", "red+green", "gray",
  "negative-key" (or no reduction for scalar frames);
* cleaning: threshold filter;  balancing: none;
* restoration: ``darsia_amd.TVD(method="heterogeneous bregman")`` (iterates inside the context), the default
  ``darsia_amd.TVD()`` (chambolle) and the two scikit-image split-Bregman methods (the operator runs on the device array between the two
  elementwise stages), or none;
* model: ``ScalingModel``, ``LinearModel``, ``ClipModel`` or a ``CombinedModel`` of them.

Anything else (callables, other reductions, other restorations ...) is applied stage by stage through
the objects' own ``__call__`` exactly as the reference does -- the device operators still do their part,
only the array makes more trips.

The small model / reduction classes are cheap per-pixel maps (out of scope as
stand-alone kernels); their ``__call__`` is plain NumPy like the reference's.
"""
from __future__ import annotations

import copy
import ctypes as C
import logging
import os
from typing import Optional
from warnings import warn

import numpy as np

from . import _lib
from .device import DeviceArray, host_empty
from .image import ScalarImage
from .restoration.split_bregman_tvd import DeviceTVD
from .restoration.tvd import TVD
from .utils.linear_solvers.jacobi import Jacobi


# ---- per-pixel maps (host semantics = reference) ---------------------------------------------------------
class MonochromaticReduction:
    """signals/reduction/monochromatic.py:12-95 (without the hsv branch, which needs skimage.color)."""

    _CODES = {"red": 1, "green": 2, "blue": 3, "red+green": 4, "gray": 5, "negative-key": 6}

    def __init__(self, **kwargs) -> None:
        self.color = kwargs.get("color", "gray")
        self.verbosity = kwargs.get("verbosity", 0)

    def __call__(self, img: np.ndarray) -> np.ndarray:
        c = self.color
        if c == "gray":  # cv2.cvtColor(img.astype(float32), COLOR_RGB2GRAY)
            f = img.astype(np.float32)
            return f[:, :, 0] * np.float32(0.299) + f[:, :, 1] * np.float32(0.587) + f[:, :, 2] * np.float32(0.114)
        if c == "red":
            return img[:, :, 0]
        if c == "green":
            return img[:, :, 1]
        if c == "blue":
            return img[:, :, 2]
        if c == "red+green":
            return img[:, :, 0] + img[:, :, 1]
        if c == "negative-key":
            return 1 - np.min(1 - img, axis=2)
        if callable(c):
            return c(img)
        if c == "":
            return img
        raise ValueError(f"Mono-colored space {c} not supported.")


class ScalingModel:
    def __init__(self, key: str = "", **kwargs) -> None:
        self._scaling = kwargs.get(key + "scaling", 1.0)
        self.num_parameters = 1

    def update(self, scaling: Optional[float] = None) -> None:
        if scaling is not None:
            self._scaling = scaling

    def __call__(self, img: np.ndarray) -> np.ndarray:
        return img if np.isclose(self._scaling, 1.0) else self._scaling * img


class LinearModel:
    def __init__(self, key: str = "", **kwargs) -> None:
        self._scaling = kwargs.get(key + "scaling", 1.0)
        self._offset = kwargs.get(key + "offset", 0.0)
        self.num_parameters = 2

    def update(self, scaling: Optional[float] = None, offset: Optional[float] = None) -> None:
        if scaling is not None:
            self._scaling = scaling
        if offset is not None:
            self._offset = offset

    def __call__(self, img: np.ndarray) -> np.ndarray:
        return self._scaling * img + self._offset


class ClipModel:
    def __init__(self, min_value=None, max_value=None, key=None, **kwargs) -> None:
        if key is None:
            self._min_value, self._max_value = min_value, max_value
        else:
            self._min_value = kwargs.get(key + "_min_value", None)
            self._max_value = kwargs.get(key + "_max_value", None)
        if self._min_value is None and self._max_value is None:
            raise ValueError("at least one of min_value or max_value must be provided")
        self.num_parameters = 2

    def update(self, min_value=None, max_value=None) -> None:
        if min_value is not None:
            self._min_value = min_value
        if max_value is not None:
            self._max_value = max_value

    def __call__(self, img):
        if isinstance(img, np.ndarray):
            return np.clip(img, self._min_value, self._max_value)
        if hasattr(img, "img") and hasattr(img, "copy"):
            result = img.copy()
            result.img = np.clip(result.img, self._min_value, self._max_value)
            return result
        raise ValueError("invalid input type")


class CombinedModel:
    """Chain of models; a model whose ``__call__`` takes more than (self, img) receives the extra
    positional arguments (combinedmodel.py:29-52)."""

    def __init__(self, models: list) -> None:
        self.models = models
        self.num_parameters = sum(getattr(m, "num_parameters", 0) for m in models)

    def __call__(self, img: np.ndarray, *args) -> np.ndarray:
        result = img.copy()
        for model in self.models:
            nargs = model.__call__.__code__.co_argcount
            result = model(result) if nargs == 2 else model(result, *args[: nargs - 2])
        return result

    def __getitem__(self, pos: int):
        return self.models[pos]


def _model_ops(model):
    """Model -> list of device ops, or None if it cannot be expressed."""
    if model is None:
        return []
    if isinstance(model, CombinedModel):
        ops = []
        for m in model.models:
            sub = _model_ops(m)
            if sub is None:
                return None
            ops += sub
        return ops if len(ops) <= 8 else None
    if isinstance(model, LinearModel):
        return [(0, 0, float(model._scaling), float(model._offset))]
    if isinstance(model, ScalingModel):
        return [] if np.isclose(model._scaling, 1.0) else [(1, 0, float(model._scaling), 0.0)]
    if isinstance(model, ClipModel):
        lo, hi = model._min_value, model._max_value
        return [(2, (1 if lo is not None else 0) | (2 if hi is not None else 0), float(lo or 0.0), float(hi or 0.0))]
    return None


def _ops_array(ops):
    arr = (_lib.ModelOp * max(1, len(ops)))()
    for i, (kind, flags, a, b) in enumerate(ops):
        arr[i].kind, arr[i].flags, arr[i].a, arr[i].b = kind, flags, a, b
    return arr


_DIFF = {"plain": 0, "positive": 1, "negative": 2, "absolute": 3}


_FRAME_DTYPES = {np.dtype(np.float32): _lib.DXH_F32, np.dtype(np.float64): _lib.DXH_F64, np.dtype(np.uint8): 2, np.dtype(np.uint16): 3}


logger = logging.getLogger(__name__)


def _float_dtype(dt) -> np.dtype:
    """The float type a probing frame is subtracted in: float frames as they are, integer frames as float64
    (concentrationanalysis.py:241-246: ``img_as(float)``)."""
    dt = np.dtype(dt)
    return dt if dt.kind == "f" else np.dtype(np.float64)


def _fingerprint(value):
    """Identity of a host array for the device-side caches (coefficient planes of a TV context, the registration's
    baseline): the object, its shape / dtype and the sum of a strided sample of ~4096 values.  The sample is a SAFETY NET,
    not a guarantee: an in-place edit that misses every sampled element (a region of up to thousands of pixels on a large
    array) or cancels in the sum goes unseen.  The contract is therefore: arrays handed to these objects are not edited in
    place; after such an edit call ``refresh()`` (ConcentrationAnalysis, TranslationAnalysis / ImageRegistration), or pass
    a new array.  Holders keep a reference to the array next to the mark, so ``id()`` cannot be recycled under them."""
    if isinstance(value, np.ndarray) and value.ndim > 0:
        flat = value.reshape(-1) if value.flags.c_contiguous else value.ravel()
        step = max(1, flat.size // 4096)
        return ("array", id(value), value.shape, value.dtype.str, float(np.sum(flat[::step], dtype=np.float64)))
    if isinstance(value, DeviceArray):
        # a device-resident array: identity only (its contents cannot be sampled without a download; same contract)
        return ("device", id(value), value.shape, value.dtype.str, value.ptr)
    return ("scalar", None if value is None else float(value))


class ConcentrationAnalysis:
    """Concentration/saturation profiles from image comparison (concentrationanalysis.py:20-421)."""

    def __init__(self, base=None, signal_reduction=None, balancing=None, restoration=None, model=None, labels=None,
                 **kwargs) -> None:
        self.base = None
        self._base_collection = []
        if base is not None:
            if not isinstance(base, list):
                base = [base]
            # baselines whose pixels are in HBM (e.g. out of the device corrections) come to the host once: the cleaning filter
            # is calibrated there, and the device contexts take their own copy when they are made
            from .image import with_array

            base = [with_array(b, b.img.get()) if isinstance(b.img, DeviceArray) else b for b in base]
            if any(img.img.dtype not in (np.float32, np.float64) for img in base):
                raise TypeError("baseline images must be float32/float64 (convert before, e.g. img / 255.)")
            self.base = base[0].copy()
            self._base_collection = base
            if getattr(self.base, "space_dim", 2) != 2:
                raise NotImplementedError
        self.signal_reduction = signal_reduction
        self.balancing = balancing
        self.model = model
        self.restoration = restoration
        self.labels = labels
        self._diff_option = kwargs.get("diff option", "absolute")
        self.first_restoration_then_model = kwargs.get("restoration -> model", False)
        self.verbosity = kwargs.get("verbosity", 0)
        # device contexts: [0] on this thread's stream, [1] on the uploader thread's, [2] on a helper thread's (map)
        self._ctxs = [None, None, None]
        self._ctx_keys = [None, None, None]
        self._uploader = None
        self._slot2_thread = None
        self.find_cleaning_filter()
        self.mask = None if self.base is None else np.ones(self.base.img.shape[:2], dtype=bool)

    # ---- stages on host arrays (used for the cleaning filter and the non-fusable path) -------------------
    def _subtract_background(self, img) -> np.ndarray:
        opt = self._diff_option
        if opt not in _DIFF:
            raise ValueError(f"Diff option {opt} not supported")
        a = img.img
        if self.base is None:
            return {"positive": lambda: np.clip(a, 0, None), "negative": lambda: np.clip(-a, 0, None),
                    "absolute": lambda: np.absolute(a), "plain": lambda: a}[opt]()
        b = self.base.img
        return {"positive": lambda: np.clip(a - b, 0, None), "negative": lambda: np.clip(b - a, 0, None),
                "absolute": lambda: np.abs(a - b), "plain": lambda: a - b}[opt]()

    def _reduce_signal(self, img: np.ndarray) -> np.ndarray:
        return img if self.signal_reduction is None else self.signal_reduction(img)

    def _clean_signal(self, img: np.ndarray) -> np.ndarray:
        return img if self.threshold_cleaning_filter is None else np.clip(img - self.threshold_cleaning_filter, 0, None)

    def _balance_signal(self, img: np.ndarray) -> np.ndarray:
        return img if self.balancing is None else self.balancing(img)

    def _restore_signal(self, signal: np.ndarray) -> np.ndarray:
        return signal if self.restoration is None else self.restoration(signal)

    def _convert_signal(self, signal: np.ndarray, diff: np.ndarray) -> np.ndarray:
        return signal if self.model is None else self.model(signal)

    def find_cleaning_filter(self, baseline_images=None) -> None:
        """Structural noise threshold from a series of baseline images (concentrationanalysis.py:167-200)."""
        if baseline_images is None and self.base is not None:
            baseline_images = self._base_collection[1:] or None
        self.threshold_cleaning_filter = None
        if baseline_images is not None:
            self.threshold_cleaning_filter = np.zeros(self.base.img.shape[:2], dtype=float)
            for img in baseline_images:
                mono = self._reduce_signal(self._subtract_background(img.copy()))
                self.threshold_cleaning_filter = np.maximum(self.threshold_cleaning_filter, mono)
        self._ctx_keys = [None, None, None]  # device copies are stale

    def update(self, base=None, mask=None) -> None:
        if base is not None:
            self.base = base.copy()
            self._ctx_keys = [None, None, None]
        if mask is not None:
            self.mask = mask

    # ---- fused device path ----------------------------------------------------------------------------------
    def _fusable(self, probe: np.ndarray):
        """The fused device plan for this frame, or None -- then ``self._why_generic`` says what keeps the frame on the
        stage-by-stage chain (host NumPy for the elementwise stages, the restoration operator on the device)."""
        plan, self._why_generic = self._fusable_why(probe)
        return plan

    def _fusable_why(self, probe: np.ndarray):
        if self.balancing is not None:
            return None, "a balancing stage is set"
        if self._diff_option not in _DIFF:
            return None, f"difference option {self._diff_option!r} has no device stage"
        if probe.ndim == 3 and probe.shape[2] == 3:
            red = self.signal_reduction
            if not isinstance(red, MonochromaticReduction) or red.color not in MonochromaticReduction._CODES:
                return None, "the signal reduction is not a MonochromaticReduction the device knows"
            reduction, channels = MonochromaticReduction._CODES[red.color], 3
            if reduction == 5:
                # cv2's gray conversion works in float32 AFTER the (possibly float64) background subtraction: the fused
                # float32 pipeline reproduces that only when nothing in front of it is wider than float32
                wide = _float_dtype(probe.dtype) if self.base is None else np.result_type(_float_dtype(probe.dtype), self.base.img.dtype)
                if wide != np.float32:
                    return None, "gray reduction behind a float64 difference"
        elif probe.ndim == 2 and self.signal_reduction is None:
            reduction, channels = 0, 1
        else:
            return None, f"frame of shape {probe.shape} with signal reduction {type(self.signal_reduction).__name__}"
        rest = self.restoration
        mode = None if rest is None else "hb"
        if isinstance(rest, TVD) and rest.method in ("chambolle", "anisotropic bregman", "isotropic bregman") and not rest.kwargs:
            # the default TVD() and scikit-image's split-Bregman methods: the operator runs on the device between the two
            # elementwise stages (device array in, device array out)
            mode, solver, iso = "op", None, False
        elif rest is not None:
            if not (isinstance(rest, TVD) and rest.method == "heterogeneous bregman"):
                return None, f"restoration {type(rest).__name__} is not a TVD operator"
            kw = dict(rest.kwargs)
            solver = kw.pop("solver", None)
            bad = (kw.pop("dim", 2) != 2 or kw.pop("x0", None) is not None or kw.pop("adaptive", None) is not None
                   or kw.pop("verbose", False) or (solver is not None and not isinstance(solver, Jacobi)))
            if bad:
                return None, "TVD keyword arguments (dim / x0 / adaptive / verbose / solver) outside the fused pipeline"
            iso = bool(kw.pop("isotropic", False))
            kw.pop("promotion", None)
            if kw:
                return None, f"unknown TVD keyword arguments {sorted(kw)}"
            for coef in (rest.weight, rest.omega, rest.regularization):
                if isinstance(coef, np.ndarray) and coef.ndim > 0 and coef.shape != probe.shape[:2]:
                    return None, "coefficient array of another shape than the frame"
            # a float32 signal: the operator itself would run the reference's own dtype flow (promotion "numpy2",
            # restoration/split_bregman_tvd.py) -- the scalar context of the pipeline then IS a numpy2 context (mode "hb2":
            # the signal is formed in float32 and written into its float64 planes, k_signal<float, .., double>; the model
            # reads the float64 iterate rounded to float32, k_model<float, double>), the iterations run where the signal lies
            from .restoration.split_bregman_tvd import choose_promotion, native_is_the_reference_flow, numpy2_unsupported
            sig = np.empty((1, 1), self._signal_dtype(probe, reduction))
            jac = solver if solver is not None else Jacobi()
            why = numpy2_unsupported(sig, rest.weight, rest.omega, rest.regularization, 2, None, iso, jac, None, True)
            want = rest.kwargs.get("promotion")
            if want in (None, "numpy2") and native_is_the_reference_flow(sig, rest.weight, rest.omega, rest.regularization):
                want = "native"
            if choose_promotion(want, sig.dtype, (rest.weight, rest.omega, rest.regularization), why) == "numpy2":
                mode = "hb2"
        else:
            solver, iso = None, False
        ops = _model_ops(self.model)
        if ops is None:
            return None, f"model {type(self.model).__name__} has no device stage"
        return (reduction, channels, ops, solver, iso, mode), None

    def _signal_dtype(self, probe: np.ndarray, reduction: int):
        # dtype as NumPy would promote (concentrationanalysis.py:328-395): probe - baseline promotes to the wider of the
        # two, a (float64) cleaning filter promotes the signal; gray is float32 in cv2
        dt = _float_dtype(probe.dtype) if self.base is None else np.result_type(_float_dtype(probe.dtype), self.base.img.dtype)
        if self.threshold_cleaning_filter is not None:
            dt = np.result_type(dt, np.float64)
        elif reduction == 5:
            dt = np.dtype(np.float32)
        return np.dtype(dt)

    def _ctx_for(self, slot: int, key, channels: int) -> DeviceTVD:
        """Scalar TV context of a slot with the baseline and the cleaning filter resident.  A context takes the stream of
        the thread that creates it: slot 0 belongs to the caller's thread, slot 1 to the uploader's, slot 2 to a helper
        thread that does nothing else (map)."""
        if self._ctxs[slot] is None or self._ctx_keys[slot] != key:
            if self._ctxs[slot] is not None:
                self._ctxs[slot].close()
                self._ctxs[slot] = None
            h, w, dts = key[:3]
            dt = np.dtype(dts)
            st = DeviceTVD((h, w), dt, dim=2, promotion="numpy2" if (len(key) > 3 and key[3] == "numpy2") else "native")
            # the baseline stays in its own float type (probe - baseline is evaluated in NumPy's promotion of the two on the
            # device): a float32 baseline under a float64 cleaning filter is half the bytes, and no host conversion
            base = None
            if self.base is not None:
                own = self.base.img.dtype in (np.float32, np.float64) and self.base.img.dtype.itemsize <= dt.itemsize
                base = np.ascontiguousarray(self.base.img, dtype=None if own else dt)
            _lib.check(st.lib.dxh_tvd_set_base_typed(st._h, _lib.ptr(base), channels, _lib.dtype_code(dt if base is None else base.dtype)))
            filt = None if self.threshold_cleaning_filter is None else np.ascontiguousarray(self.threshold_cleaning_filter, dtype=dt)
            _lib.check(st.lib.dxh_tvd_set_cleaning_filter(st._h, _lib.ptr(filt)))
            self._ctxs[slot], self._ctx_keys[slot] = st, key
        return self._ctxs[slot]

    def refresh(self) -> None:
        """Drop every device-side copy (baseline, cleaning filter, coefficient planes): the next frame uploads them
        again.  Needed after an IN-PLACE edit of the baseline image or of a coefficient array of the restoration (new
        objects are noticed by themselves, ``_fingerprint``)."""
        self._ctx_keys = [None, None, None]
        for st in getattr(self, "_ctxs", []):
            if st is not None:
                st._coef_mark = None

    def close(self) -> None:
        """Release the device contexts and the uploader thread (both come back on demand)."""
        for name in ("_uploader", "_slot2_thread"):
            if getattr(self, name, None) is not None:
                getattr(self, name).shutdown(wait=True)
                setattr(self, name, None)
        for i, st in enumerate(getattr(self, "_ctxs", [])):
            if st is not None:
                st.close()
                self._ctxs[i] = None

    def _upload_frame(self, st: DeviceTVD, probe: np.ndarray, plan) -> None:
        reduction, channels, ops, solver, iso, mode = plan
        rest = self.restoration
        like = np.empty(st.shape, st.dtype)
        coefs = (rest.weight, rest.omega, rest.regularization) if mode in ("hb", "hb2") else (1.0, 1.0, None)
        # the coefficients stay in the context from frame to frame: a weight map (porosity) is 128 - 256 MB of upload and
        # three derived planes per frame otherwise.  "Unchanged" = the same objects with the same fingerprint (_fingerprint:
        # in-place edits are outside the contract, refresh() drops the copies)
        mark = tuple(_fingerprint(v) for v in coefs)
        if getattr(st, "_coef_mark", None) != mark:
            st.set_coeffs(*coefs, like)
            st._coef_mark = mark
            st._coef_refs = coefs      # keeps the arrays alive: their id() stays theirs
        pre = ops if not self.first_restoration_then_model else []
        # the frame goes over as stored (float32 / float64 / uint8 / uint16); the device widens where NumPy would.  A frame that
        # is in HBM already (DeviceArray: e.g. out of the colour or shape correction) is read where it lies.
        if isinstance(probe, DeviceArray):
            if np.dtype(probe.dtype) not in _FRAME_DTYPES:
                raise TypeError(f"device-resident frames must be float32 / float64 / uint8 / uint16, not {probe.dtype}")
            src = probe
        else:
            src = np.ascontiguousarray(probe, dtype=None if probe.dtype in _FRAME_DTYPES else st.dtype)
        _lib.check(st.lib.dxh_tvd_upload_signal_typed(st._h, _lib.ptr(src), _FRAME_DTYPES[np.dtype(src.dtype)], channels,
                                                       _DIFF[self._diff_option], reduction,
                                                       C.cast(_ops_array(pre), C.c_void_p), len(pre)))

    def _finish_frame(self, st: DeviceTVD, plan, on_device: bool = False):
        self._start_frame(st, plan)
        return self._fetch_frame(st, plan, on_device)

    def _start_frame(self, st: DeviceTVD, plan) -> None:
        """The restoration of the frame that is in ``st``: enqueued only where nothing has to be read back (no stopping
        criteria), so that the caller can fetch the previous frame / send the next one meanwhile (``map``)."""
        reduction, channels, ops, solver, iso, mode = plan
        rest = self.restoration
        if mode == "op":
            # signal out of the context, through the operator, back in: three copies inside HBM (~0.1 ms each)
            signal = st.download(out=DeviceArray(st.shape, st.dtype, st.device))
            restored = rest(signal)
            signal.close()
            st.upload(restored)
            restored.close()
        elif mode in ("hb", "hb2"):
            maxiter, tol = (1, None) if solver is None else (solver.maxiter, solver.tol)
            # nothing to read back without stopping criteria: enqueue only, the download below maps the pages of the
            # result array meanwhile
            flags = _lib.FLAG_ASYNC if rest.eps is None and tol is None else 0
            st.run(rest.max_num_iter, maxiter, tol, eps=rest.eps, isotropic=iso, flags=flags)

    def _fetch_frame(self, st: DeviceTVD, plan, on_device: bool = False):
        """The model stage and the way back of the frame whose restoration ``_start_frame`` started."""
        ops = plan[2]
        post = ops if self.first_restoration_then_model else []
        out = DeviceArray(st.shape, st.dtype, st.device) if on_device else host_empty(st.shape, st.dtype)
        _lib.check(st.lib.dxh_tvd_download_model(st._h, _lib.ptr(out), C.cast(_ops_array(post), C.c_void_p), len(post)))
        return out

    def _device_call(self, probe: np.ndarray, plan) -> np.ndarray:
        dt = self._signal_dtype(probe, plan[0])
        st = self._ctx_for(0, (probe.shape[0], probe.shape[1], dt.str, "numpy2" if plan[5] == "hb2" else "native"), plan[1])
        self._upload_frame(st, probe, plan)
        return self._finish_frame(st, plan, isinstance(probe, DeviceArray))   # device frame in, device concentration out

    def _wrap(self, img, concentration: np.ndarray):
        probe = img.img
        metadata = img.metadata() if hasattr(img, "metadata") else {}
        if concentration.ndim == probe.ndim - 1 or concentration.ndim == 2:
            return ScalarImage(concentration, **{k: v for k, v in metadata.items() if k != "scalar"})
        if concentration.shape[-1] == 1:
            return ScalarImage(concentration[..., 0], **{k: v for k, v in metadata.items() if k != "scalar"})
        return type(img)(concentration, **metadata)

    def map(self, images):
        """Concentrations of a SERIES of frames (the reference analyses time series frame by frame,
        concentrationanalysis.py:230-310), yielded in order.  Same results as calling the object on every frame; the
        frames alternate between two device contexts on two streams, so that the upload of frame i + 1 (8 ms of PCIe
        for a 4000 x 8000 colour frame) runs while frame i is reduced, denoised and downloaded.  Frames the fused device
        path cannot take are processed one after the other."""
        from concurrent.futures import ThreadPoolExecutor

        it = iter(images)
        try:
            cur = next(it)
        except StopIteration:
            return

        def fusable(img):
            if img.img.dtype not in _FRAME_DTYPES:
                return None
            return self._fusable(img.img)

        def submit(slot: int, img):
            if img.img.dtype.kind != "f":   # concentrationanalysis.py:241-246, once per frame as in a frame-by-frame loop
                warn("The input for concentration analysis needed to be converted to float.")
            return pool.submit(upload, slot, img)

        plan = fusable(cur)
        if plan is None:
            yield self(cur)
            for img in it:
                yield self(img)
            return
        dt = self._signal_dtype(cur.img, plan[0])
        shape, channels = cur.img.shape, plan[1]
        key = (shape[0], shape[1], dt.str, "numpy2" if plan[5] == "hb2" else "native")
        if self._uploader is None:
            self._uploader = ThreadPoolExecutor(max_workers=1, thread_name_prefix="dxh-upload")
        pool = self._uploader
        self._ctx_for(0, key, channels)   # on this thread; slot 1's is created by the uploader at its first frame

        def upload(slot: int, img):
            self._upload_frame(self._ctx_for(slot, key, channels), img.img, plan)

        fut = submit(0, cur)
        fut_next = None
        try:
            # Two slots, two schedules (measured on 4000 x 8000 frames, 100 TV iterations; three slots with everything side
            # by side were slower than either):
            #  * upload-bound (float32 / float64 colour frames in the native flow: 7 - 15 ms of PCIe per frame): frame k + 1
            #    goes up while frame k iterates, runs its model stage and comes back;
            #  * compute-bound (the reference's float32 flow, whose TV stage is 2.3 x longer; frames of few bytes per
            #    pixel): frame k iterates (enqueued) while frame k - 1 comes back AND frame k + 1 goes up into a THIRD slot
            #    (three contexts on three streams: the copy up, the iterations and the copy down all overlap) -- results are
            #    yielded one frame behind the iterations.  DXH_MAP_SLOTS=2: frame k + 1 goes up only after frame k - 1 has
            #    come back, into its slot (round 3's schedule, A/B).
            bytes_per_pixel = cur.img.dtype.itemsize * (1 if cur.img.ndim == 2 else cur.img.shape[2])
            late_upload = plan[5] == "hb2" or bytes_per_pixel <= 6
            ns = 3 if (late_upload and os.environ.get("DXH_MAP_SLOTS", "3") != "2") else 2
            if ns == 3:
                if self._slot2_thread is None:
                    self._slot2_thread = ThreadPoolExecutor(max_workers=1, thread_name_prefix="dxh-slot2")
                self._slot2_thread.submit(self._ctx_for, 2, key, channels).result()
            k = 0
            pending = None   # (slot, image) of the frame whose iterations have been started but whose result is still on the device
            while cur is not None:
                nxt = next(it, None)
                same = nxt is not None and nxt.img.shape == shape and fusable(nxt) == plan and \
                    self._signal_dtype(nxt.img, plan[0]) == dt
                if same and ns == 3:
                    # into the slot of frame k - 2, fetched one turn ago; queued behind the upload of frame k on the uploader
                    # thread, so that the copies up follow each other without a gap
                    fut_next = submit((k + 1) % ns, nxt)
                fut.result()                                      # frame k is on the device (slot k % ns)
                fut, fut_next = fut_next, None
                if same and not late_upload:
                    fut = submit((k + 1) % ns, nxt)   # overlaps the work on frame k below
                self._start_frame(self._ctxs[k % ns], plan)
                if pending is not None:
                    out = self._fetch_frame(self._ctxs[pending[0]], plan, isinstance(pending[1].img, DeviceArray))
                    yield self._wrap(pending[1], out)
                    pending = None
                if late_upload:
                    if same and ns == 2:
                        fut = submit((k + 1) % ns, nxt)   # into the slot just fetched; overlaps the iterations of frame k
                    pending = (k % ns, cur)
                else:
                    out = self._fetch_frame(self._ctxs[k % ns], plan, isinstance(cur.img, DeviceArray))
                    yield self._wrap(cur, out)
                if nxt is not None and not same:
                    # a frame of another kind ends the pipelined stretch
                    if pending is not None:
                        out = self._fetch_frame(self._ctxs[pending[0]], plan, isinstance(pending[1].img, DeviceArray))
                        yield self._wrap(pending[1], out)
                        pending = None
                    yield self(nxt)
                    for img in it:
                        yield self(img)
                    return
                cur = nxt
                k += 1
            if pending is not None:
                out = self._fetch_frame(self._ctxs[pending[0]], plan, isinstance(pending[1].img, DeviceArray))
                yield self._wrap(pending[1], out)
        finally:
            for f in (fut, fut_next):   # generator abandoned or failed: join the uploads in flight
                if f is not None:
                    try:
                        f.result()
                    except Exception:
                        pass

    def __call__(self, img):
        """Extract the concentration from a probing image (concentrationanalysis.py:230-310)."""
        if img.img.dtype not in (np.float32, np.float64):
            # concentrationanalysis.py:241-246: converted the scikit-image way (uint8 / 255 ... as float64), with a warning
            if img.img.dtype not in (np.uint8, np.uint16) or not hasattr(img, "img_as"):
                raise TypeError("the probing image must be float32 / float64 (or uint8 / uint16, which are converted)")
            warn("The input for concentration analysis needed to be converted to float.")
            if self._fusable(img.img) is None:
                img = img.img_as(float)   # on the host for the generic chain; the fused path converts on the device
        probe = img.img
        plan = self._fusable(probe)
        # which path served the frame: "fused" (one device pipeline from frame to concentration) or "generic" (the
        # reference's stage-by-stage chain: NumPy on the host for the elementwise stages, the restoration operator on the
        # device) -- observable, because the two differ by an order of magnitude in time
        self.last_path, self.last_path_reason = ("fused", None) if plan is not None else ("generic", self._why_generic)
        if plan is not None:
            concentration = self._device_call(probe, plan)
        elif isinstance(probe, DeviceArray):
            raise NotImplementedError("a device-resident frame needs the fused pipeline (monochromatic reduction, model of "
                                      "linear / scaling / clip stages, TVD or no restoration, no balancing)")
        else:
            logger.info("ConcentrationAnalysis: stage-by-stage chain on the host (%s)", self._why_generic)
            probe_img = copy.deepcopy(img)
            diff = self._subtract_background(probe_img)
            signal = self._balance_signal(self._clean_signal(self._reduce_signal(diff)))
            if self.first_restoration_then_model:
                concentration = self._convert_signal(self._restore_signal(signal), diff)
            else:
                concentration = self._restore_signal(self._convert_signal(signal, diff))
        return self._wrap(img, concentration)
