"""Developer scan: for which (mu, omega, ell) does the exhaustive check reject the three-operation division?
Runs tiny float32 solves with DXH_FUSED_TUNE=512 (debug line on stderr) in one child per batch."""
import os, subprocess, sys
CHILD = r'''
import sys; sys.path.insert(0, ".")
import numpy as np, darsia_amd as da
rng = np.random.default_rng(int(sys.argv[1]))
img = rng.random((40, 300, 3)).astype(np.float32)
for i in range(int(sys.argv[2])):
    mu, om, ell = float(rng.uniform(0.01, 2)), float(rng.choice([1.0, rng.uniform(0.2, 3)])), float(rng.uniform(0.02, 3))
    print(f"case {mu!r} {om!r} {ell!r}", file=sys.stderr, flush=True)
    da.split_bregman_tvd(img, mu=mu, omega=om, ell=ell, max_num_iter=4)
'''
env = dict(os.environ, DXH_FUSED_TUNE="512")
out = subprocess.run([sys.executable, "-c", CHILD, sys.argv[1], sys.argv[2]], env=env, capture_output=True, text=True)
case = None; n = bad = 0
for line in out.stderr.splitlines():
    if line.startswith("case"): case = line
    elif line.startswith("[dxh]") and case:
        n += 1
        if "div3=0" in line: bad += 1; print("rejected:", case)
        case = None
print(n, "coefficient sets,", bad, "rejected")
