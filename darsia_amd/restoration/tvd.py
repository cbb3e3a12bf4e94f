"""``TVD`` / ``tvd``: DarSIA's TV-denoising operator interface (reference:
restoration/tvd.py:15-163) in front of the MI355X split-Bregman implementation.

The constructor reproduces the reference's kwarg handling: ``key``-prefixed
``method`` / ``weight`` / ``max_num_iter`` / ``eps``, un-prefixed ``omega`` and
``regularization`` for the heterogeneous method, leftovers forwarded.

Divergence: the reference only *reads* ``regularization``
(tvd.py:37) and then forwards it a second time through ``**kwargs``, which
raises ``TypeError`` inside ``split_bregman_tvd``.  Here it is popped, so
``TVD(method="heterogeneous bregman", regularization=...)`` works.
"""
from __future__ import annotations

import threading

import numpy as np

from ..device import DeviceArray
from ..image import with_array
from .bregman_gs import denoise_tv_bregman
from .chambolle import denoise_tv_chambolle
from .split_bregman_tvd import split_bregman_tvd

_SKIMAGE_BREGMAN = ("anisotropic bregman", "isotropic bregman")


def _is_image(obj) -> bool:
    """darsia.Image duck type: carries ``.img`` and can ``copy()`` itself."""
    return hasattr(obj, "img") and hasattr(obj, "copy") and not isinstance(obj, np.ndarray)


class TVD:
    """Total variation denoising interface (tvd.py:15-129)."""

    def __init__(self, key: str = "", **kwargs) -> None:
        self.method = kwargs.pop(key + "method", "chambolle").lower()
        if self.method == "heterogeneous bregman":
            self.omega = kwargs.pop("omega", 1)
            self.regularization = kwargs.pop("regularization", 1.0)
        self.weight = kwargs.pop(key + "weight", 0.1)
        self.max_num_iter = kwargs.pop(key + "max_num_iter", 200)
        self.eps = kwargs.pop(key + "eps", 2e-4)
        self.kwargs = kwargs
        # Device contexts kept between calls (numpy2 flow: the coefficient planes -- a 256 MB float64 weight map at
        # 4000 x 8000 -- stay in HBM).  CONTRACT: ``weight`` / ``omega`` / ``regularization`` arrays are not edited in place
        # between calls; the planes are refreshed when the array OBJECT or a strided sample of its values changes
        # (concentration._fingerprint), which a localized in-place edit can miss -- after such an edit call ``refresh()``
        # (or assign a new array).  One call at a time per operator: the lock serialises concurrent callers.
        self._contexts = {}
        self._lock = threading.RLock()

    def __call__(self, img):
        with self._lock:
            if isinstance(img, (np.ndarray, DeviceArray)):  # a DeviceArray stays in HBM (darsia_amd.device)
                return self._tvd_array(img)
            elif _is_image(img):
                return self._tvd_image(img)
            else:
                raise TypeError(f"Input type {type(img)} not supported.")

    def _tvd_array(self, img: np.ndarray) -> np.ndarray:
        if self.method == "heterogeneous bregman":
            return split_bregman_tvd(
                img,
                mu=self.weight,
                omega=self.omega,
                ell=self.regularization,
                max_num_iter=self.max_num_iter,
                eps=self.eps,
                _context_cache=self._contexts,
                **self.kwargs,
            )
        elif self.method == "chambolle":
            # tvd.py:77-83: no channel axis -> colour images are denoised as 3-D volumes
            return denoise_tv_chambolle(img, weight=self.weight, max_num_iter=self.max_num_iter, eps=self.eps)
        elif self.method in _SKIMAGE_BREGMAN:
            # tvd.py:85-101: scikit-image's Gauss-Seidel split Bregman, no channel axis
            return denoise_tv_bregman(img, weight=self.weight, max_num_iter=self.max_num_iter, eps=self.eps,
                                      isotropic=self.method == "isotropic bregman")
        else:
            raise ValueError(f"Method {self.method} not supported.")

    def release(self) -> None:
        """Free the device contexts the operator keeps between calls (they come back on demand)."""
        with self._lock:
            for held in self._contexts.values():
                held[2].close()
            self._contexts.clear()

    def refresh(self) -> None:
        """Forget the coefficient planes held on the device: the next call uploads ``weight`` / ``omega`` /
        ``regularization`` afresh.  Call it after editing one of those arrays IN PLACE (see ``_contexts`` above)."""
        self.release()

    def __deepcopy__(self, memo):
        import copy

        new = copy.copy(self)
        # arrays are duplicated as the default deepcopy would (a copy must not see later edits of the original's weight map)
        for name in ("weight", "omega", "regularization"):
            setattr(new, name, copy.deepcopy(getattr(self, name), memo))
        new.kwargs = copy.deepcopy(self.kwargs, memo)
        new._contexts = {}          # device contexts are not shared between copies
        new._lock = threading.RLock()
        return new

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_contexts"] = {}     # device handles and the lock do not travel (pickle, multiprocessing)
        state.pop("_lock", None)
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._lock = threading.RLock()

    def _tvd_image(self, img):
        # tvd.py:127-129: a copy of the image with the denoised array
        return with_array(img, self._tvd_array(img.img))


def tvd(img, method: str = "chambolle", weight=1.0, max_num_iter: int = 200, eps: float = 2e-4, **kwargs):
    """Inline application of TVD (tvd.py:132-163)."""
    return TVD(method=method, weight=weight, max_num_iter=max_num_iter, eps=eps, **kwargs)(img)
