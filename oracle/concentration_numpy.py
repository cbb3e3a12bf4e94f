"""CPU restatement (NumPy) of the elementwise chain of DarSIA's ConcentrationAnalysis
(multi_image_analysis/concentrationanalysis.py:230-421) around the TV restoration.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  PINNED: bit for bit on outputs of the reference's own class
(tests/golden/concentration_golden.npz, made by tests/golden/make_golden_concentration.py, which executes the reference's
concentrationanalysis.py, monochromatic.py, the model classes and tvd.py under scikit-image 0.18.3 / NumPy 1.26;
tests/test_concentration_golden_cpu.py): the four difference options, the red / green / blue / red+green / negative-key
reductions, cleaning filters, Linear / Clip / Scaling / Combined models, TVD("heterogeneous bregman") before and after
the model, float32 and float64.  The one exception is ``cv2.cvtColor(.., COLOR_RGB2GRAY)`` for the "gray" reduction
(float32 weights 0.299 / 0.587 / 0.114), which stays UNPINNED (cv2 absent; that reduction is not in the vectors).
Version label of that restatement: ``RGB2Gray<float>`` of OpenCV 2.4 .. 4.x (``modules/imgproc/src/color_rgb.simd.hpp``, before
3.4.2 ``color.cpp``): ``src[0] * 0.299f + src[1] * 0.587f + src[2] * 0.114f`` in float32, left to right, for float input -- the
only input type ``MonochromaticReduction`` hands it (signals/reduction/monochromatic.py:65-68: ``img.astype(np.float32)``); the
8-bit integer path (15-bit fixed point since 3.4.2, 14-bit before) is not used.
"""
from __future__ import annotations

import numpy as np

from . import tvd_numpy


def diff(img, base, option):
    # concentrationanalysis.py:328-364 (skimage.util.compare_images(method="diff") is |a - b| for floats)
    if base is None:
        return {"positive": np.clip(img, 0, None), "negative": np.clip(-img, 0, None), "absolute": np.absolute(img),
                "plain": img}[option]
    return {"positive": np.clip(img - base, 0, None), "negative": np.clip(base - img, 0, None),
            "absolute": np.abs(img - base), "plain": img - base}[option]


def reduce(img, color):
    # signals/reduction/monochromatic.py:66-95
    if color is None:
        return img
    if color == "gray":
        f = img.astype(np.float32)
        return f[:, :, 0] * np.float32(0.299) + f[:, :, 1] * np.float32(0.587) + f[:, :, 2] * np.float32(0.114)
    if color in ("red", "green", "blue"):
        return img[:, :, ("red", "green", "blue").index(color)]
    if color == "red+green":
        return img[:, :, 0] + img[:, :, 1]
    if color == "negative-key":
        return 1 - np.min(1 - img, axis=2)
    raise ValueError(color)


def concentration(img, base, option, color, cleaning_filter, model, tvd_kwargs, restoration_first, promotion="strict"):
    """model: list of callables applied in order; tvd_kwargs: None or kwargs of tvd_numpy.split_bregman_tvd.
    ``promotion``: the dtype flow of the TV stage on a float32 signal -- "strict" is the reference under NumPy < 2 (what the
    class's golden vectors were made with), "numpy2" the reference under NumPy >= 2 (tvd_numpy's docstring)."""
    signal = reduce(diff(img, base, option), color)
    if cleaning_filter is not None:
        signal = np.clip(signal - cleaning_filter, 0, None)

    def apply_model(x):
        for m in model:
            x = m(x)
        return x

    def restore(x):
        return x if tvd_kwargs is None else tvd_numpy.split_bregman_tvd(np.ascontiguousarray(x), promotion=promotion, **tvd_kwargs)

    return apply_model(restore(signal)) if restoration_first else restore(apply_model(signal))
