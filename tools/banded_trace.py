import sys, os
sys.path.insert(0, ".")
import numpy as np
from darsia_amd.restoration.split_bregman_tvd import solve_banded
rgb = np.random.default_rng(0).random((4000, 8000, 3)).astype(np.float32)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 3
solve_banded(rgb, 0.1, 1.0, None, 100, nbands=nb); solve_banded(rgb, 0.1, 1.0, None, 100, nbands=nb)
os.environ["DXH_BAND_TRACE"] = "1"
solve_banded(rgb, 0.1, 1.0, None, 100, nbands=nb)
