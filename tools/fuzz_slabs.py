"""Developer fuzz: dxh_tvd_run_slab with 2-4 ranks on one GPU (host-staged transport over local sockets) against the
single-context run: shapes, halo widths, iteration counts, several calls per sequence, eps, isotropic, array-valued mu,
with and without the overlapped schedule."""
import os, sys, tempfile, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np

def main():
    import darsia_amd as da
    from test_parallel_gpu import _spawn
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    bad = 0
    for i in range(n_cases):
        world = int(rng.integers(2, 5))
        halo = int(rng.integers(1, 7))
        tall = rng.random() < 0.4  # tall slabs get the three-run (interface-first) chunk layout
        h = int(rng.integers(world * (2 * halo + 6), 2200 if tall else 260)); w = int(rng.integers(8, 400)); c = int(rng.choice([0, 3]))
        shape = (h, w) if c == 0 else (h, w, c)
        case = dict(shape=shape, world=world, halo_iters=halo, n_iter=int(rng.integers(1, 25)))
        if rng.random() < 0.3: case["eps"] = float(rng.choice([3e-3, 1e-2]))
        if rng.random() < 0.3: case["iso"] = True
        if rng.random() < 0.3: case["het"] = True
        if rng.random() < 0.3: case["no_overlap"] = True
        if "eps" not in case and rng.random() < 0.4:
            n, parts = case["n_iter"], []
            while n > 0:
                k = int(rng.integers(1, n + 1)); parts.append(k); n -= k
            case["parts"] = parts
            case["async"] = bool(rng.random() < 0.5)
        img = rng.random(shape).astype(np.float32)
        with tempfile.TemporaryDirectory() as tmp:
            np.save(os.path.join(tmp, "img.npy"), img)
            mu = (0.03 + 0.1 * rng.random(shape)).astype(np.float32) if case.get("het") else 0.1
            if case.get("het"): np.save(os.path.join(tmp, "mu.npy"), mu)
            _spawn(world, case, tmp)
            got = np.concatenate([np.load(os.path.join(tmp, f"out_{r}.npy")) for r in range(world)], axis=0)
            dones = {int(np.load(os.path.join(tmp, f"done_{r}.npy"))) for r in range(world)}
        with da.DeviceTVD(img.shape, img.dtype, dim=2) as st:
            st.set_coeffs(mu, 1.0, None, img); st.upload(img)
            res = st.run(case["n_iter"], eps=case.get("eps"), isotropic=case.get("iso", False)); ref = st.download()
        ok = np.array_equal(got, ref) and dones == {res.iters_done}
        print(("ok      " if ok else "MISMATCH"), case, dones, res.iters_done, flush=True)
        bad += int(not ok)
    print(f"{n_cases} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)

if __name__ == "__main__":
    main()
