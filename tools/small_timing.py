"""Developer timing: small images (config 0: 512 x 512 single channel, 20 iterations) -- whole call and device loop."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
for shape, dt in (((512, 512), np.float64), ((512, 512), np.float32), ((1024, 1024, 3), np.float32), ((64, 64), np.float32)):
    img = np.random.default_rng(0).random(shape).astype(dt)
    da.split_bregman_tvd(img, mu=0.1, max_num_iter=20)
    t0 = time.perf_counter()
    for _ in range(20): da.split_bregman_tvd(img, mu=0.1, max_num_iter=20)
    whole = (time.perf_counter() - t0) / 20 * 1e3
    st = da.DeviceTVD(img.shape, img.dtype); st.set_coeffs(0.1, 1.0, None, img); st.upload(img); st.run(20)
    r = st.run(200); st.close()
    print(f"{shape} {np.dtype(dt).name}: whole call 20 it {whole:.3f} ms; device loop {r.loop_ms/200*1e3:.1f} us/iteration ({r.launches} launches)")
    op = da.TVD()
    op(img); t0 = time.perf_counter()
    for _ in range(20): op(img)
    print(f"    TVD() default: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
