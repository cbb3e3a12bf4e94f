"""Developer fuzz: the RCCL exchange path on ONE GPU (single rank exchanging with itself = periodic problem along axis 0)
against the thrice-tiled image run in one piece: random shapes, halo widths, iteration counts, call splitting, overlap on
and off, wait-value and polling-kernel variants."""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import darsia_amd as da
from darsia_amd import _lib
from darsia_amd.parallel import Comm, SlabTVD, make_plan, periodic_slab

def main():
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    os.environ.setdefault("MASTER_PORT", "29733")
    comm = Comm.rccl(0, 1, device=0)
    bad = 0
    for i in range(n_cases):
        halo = int(rng.integers(1, 25)); n = int(rng.integers(1, 90))
        tall = rng.random() < 0.5
        h = int(rng.integers(max(n + 3, 2 * halo + 8), 2600 if tall else 400)); w = int(rng.integers(16, 600)); c = int(rng.choice([0, 3]))
        shape = (h, w) if c == 0 else (h, w, c)
        img = rng.random(shape).astype(np.float32)
        flags = _lib.FLAG_NO_OVERLAP if rng.random() < 0.25 else 0
        if rng.random() < 0.3: os.environ["DXH_SLAB_WAIT"] = "2"
        else: os.environ.pop("DXH_SLAB_WAIT", None)
        parts, left = [], n
        while left > 0:
            k = int(rng.integers(1, left + 1)) if rng.random() < 0.5 else left
            parts.append(k); left -= k
        plan = make_plan(h, 1, 0, halo, periodic=True)
        flow = "numpy2" if rng.random() < 0.5 else "native"      # both dtype flows of float32 images
        s = SlabTVD(plan, periodic_slab(img, plan), mu=0.1, halo_iters=halo, comm=comm, flags=flags, promotion=flow)
        done = 0
        for p in parts:
            done += s.run(p, more_to_come=done + p < n, asynchronous=bool(rng.random() < 0.7))
        got = s.owned_result(); s.close()
        tiled = np.ascontiguousarray(np.concatenate([img, img, img], axis=0))
        with da.DeviceTVD(tiled.shape, tiled.dtype, dim=2, promotion=flow) as st:
            st.set_coeffs(0.1, 1.0, None, tiled); st.upload(tiled); st.run(n); ref = st.download()[h:2 * h]
        ok = np.array_equal(got, ref)
        print(("ok      " if ok else "MISMATCH"), shape, f"{flow} halo={halo} n={n} parts={parts} no_overlap={bool(flags)} wait={os.environ.get('DXH_SLAB_WAIT', '-')}", flush=True)
        bad += int(not ok)
    comm.close()
    print(f"{n_cases} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)

if __name__ == "__main__":
    main()
