"""Golden vectors for the scikit-image-backed TVD methods (reference call site
restoration/tvd.py:77-83: ``denoise_tv_chambolle(img, weight, max_num_iter, eps)`` WITHOUT
a channel axis, so colour images are treated as 3-D volumes).

TEST INFRASTRUCTURE.  scikit-image is a third-party, UNPINNED dependency of the
reference (pyproject.toml:44); the only copy available here is 0.18.3 under
/opt/conda (python 3.9), whose keyword is ``n_iter_max`` instead of
``max_num_iter``.  Run with that interpreter:

    /opt/conda/bin/python3.9 tests/golden/make_golden_skimage.py

The vectors are labelled with the library version; parity for these methods is
"unpinned by the reference" (its tests hold no numeric result for them).
"""
import os

import numpy as np
import skimage
from skimage.restoration import denoise_tv_bregman, denoise_tv_chambolle

HERE = os.path.dirname(os.path.abspath(__file__))


def noisy_blocks(shape, seed, dtype):
    rng = np.random.default_rng(seed)
    h, w = shape[:2]
    y, x = np.mgrid[0:h, 0:w]
    base = 0.5 + 0.25 * np.sign(np.sin(2 * np.pi * x / 32.0 + 0.3)) * np.sign(np.sin(2 * np.pi * y / 24.0 + 0.2))
    if len(shape) == 3:
        base = base[..., None] * np.linspace(1.0, 0.6, shape[2])
    img = base + rng.normal(0.0, 0.05, size=shape)
    return np.clip(img, 0.0, 1.0).astype(dtype)


def main():
    out = {"meta/skimage": np.array(skimage.__version__), "meta/numpy": np.array(np.__version__)}
    cases = [
        ("c2d_f64", noisy_blocks((40, 56), 21, np.float64), dict(weight=0.1, eps=2e-4, n_iter_max=200)),
        ("c2d_f32", noisy_blocks((40, 56), 21, np.float32), dict(weight=0.1, eps=2e-4, n_iter_max=200)),
        ("c2d_fixed_f64", noisy_blocks((33, 47), 22, np.float64), dict(weight=0.2, eps=0.0, n_iter_max=15)),
        ("c2d_fixed_f32", noisy_blocks((33, 47), 22, np.float32), dict(weight=0.2, eps=0.0, n_iter_max=15)),
        ("crgb_f32", noisy_blocks((30, 44, 3), 23, np.float32), dict(weight=0.1, eps=2e-4, n_iter_max=200)),
        ("crgb_fixed_f64", noisy_blocks((30, 44, 3), 23, np.float64), dict(weight=0.15, eps=0.0, n_iter_max=12)),
        ("c2d_one_iter", noisy_blocks((9, 11), 24, np.float64), dict(weight=0.1, eps=2e-4, n_iter_max=1)),
        ("c2d_tiny", noisy_blocks((1, 7), 25, np.float64), dict(weight=0.1, eps=0.0, n_iter_max=5)),
    ]
    for name, img, kw in cases:
        out[name + "/img"] = img
        out[name + "/out"] = denoise_tv_chambolle(img.copy(), **kw)
        out[name + "/kw"] = np.array([kw["weight"], kw["eps"], kw["n_iter_max"]], dtype=np.float64)
    # Gauss-Seidel split Bregman (tvd.py:85-101 calls it without a channel axis = multichannel=False in 0.18.3)
    bcases = [
        ("b_aniso_f64", noisy_blocks((40, 56), 31, np.float64), dict(weight=0.3, eps=1e-3, max_iter=100, isotropic=False)),
        ("b_iso_f64", noisy_blocks((40, 56), 31, np.float64), dict(weight=0.3, eps=1e-3, max_iter=100, isotropic=True)),
        ("b_aniso_f32", noisy_blocks((40, 56), 31, np.float32), dict(weight=4.0, eps=1e-4, max_iter=60, isotropic=False)),
        ("b_iso_f32", noisy_blocks((40, 56), 31, np.float32), dict(weight=4.0, eps=1e-4, max_iter=60, isotropic=True)),
        ("b_rgb_aniso_f32", noisy_blocks((33, 47, 3), 32, np.float32), dict(weight=2.0, eps=2e-4, max_iter=200, isotropic=False)),
        ("b_rgb_iso_f64", noisy_blocks((33, 47, 3), 32, np.float64), dict(weight=1.0, eps=0.0, max_iter=12, isotropic=True)),
        ("b_wide_aniso_f64", noisy_blocks((7, 130), 33, np.float64), dict(weight=8.0, eps=0.0, max_iter=9, isotropic=False)),
        ("b_tall_iso_f32", noisy_blocks((97, 5), 34, np.float32), dict(weight=0.8, eps=0.0, max_iter=7, isotropic=True)),
        ("b_uint8", (noisy_blocks((24, 31), 35, np.float64) * 255).astype(np.uint8), dict(weight=0.1, eps=1e-4, max_iter=100, isotropic=False)),
    ]
    for name, img, kw in bcases:
        out[name + "/img"] = img
        out[name + "/out"] = denoise_tv_bregman(img.copy(), multichannel=False, **kw)
        out[name + "/kw"] = np.array([kw["weight"], kw["eps"], kw["max_iter"], float(kw["isotropic"])], dtype=np.float64)
    path = os.path.join(HERE, "skimage_0_18_3_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", skimage.__version__)


if __name__ == "__main__":
    main()
