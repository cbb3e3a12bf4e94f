"""Developer timing: a raw 4000 x 8000 x 3 uint8 frame through colour correction -> curvature correction -> registration onto the
baseline (10 x 20 patches) -> concentration analysis (100 TV iterations), with the pixels in HBM from the first stage on against
the same operators called one by one on host arrays."""
import sys, time, warnings
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from scipy.ndimage import gaussian_filter

warnings.simplefilter("ignore")
H, W = 4000, 8000
rng = np.random.default_rng(5)
tex = gaussian_filter(rng.random((H, W), dtype=np.float32), 3.0)
tex = (tex - tex.min()) / (tex.max() - tex.min())
raw_base = (255 * np.clip(0.6 * tex[..., None] * np.array([1.0, 0.9, 0.8], np.float32) + 0.1, 0, 1)).astype(np.uint8)
raw_frame = np.roll(raw_base, (3, -4), axis=(0, 1))
raw_frame[1500:2500, 3000:5000, 0] = np.minimum(255, raw_frame[1500:2500, 3000:5000, 0].astype(np.int32) + 60).astype(np.uint8)

balance = da.AffineBalance()
balance.balance_scaling, balance.balance_translation = np.eye(3) + 0.03 * rng.random((3, 3)), 0.005 * rng.random(3)
curvature = da.CurvatureCorrection(config={"bulge": dict(horizontal_bulge=2e-9, vertical_bulge=-1e-9)})
tvd = da.TVD(method="heterogeneous bregman", weight=0.05, omega=1.0, regularization=0.1, max_num_iter=100, eps=None)
model = da.CombinedModel([da.LinearModel(scaling=2.0, offset=0.0), da.ClipModel(min_value=0.0, max_value=1.0)])


def prepare(raw):
    return curvature(balance.apply_balance(raw))


base_h = da.OpticalImage(prepare(raw_base))
reg_h = da.ImageRegistration(base_h, N_patches=[10, 20], rel_overlap=0.1)
ca = da.ConcentrationAnalysis(base_h, da.MonochromaticReduction(color="red"), None, tvd, model, **{"diff option": "positive"})
base_d = da.OpticalImage(prepare(da.DeviceArray.from_host(raw_base)))
reg_d = da.ImageRegistration(base_d, N_patches=[10, 20], rel_overlap=0.1)


def through_host(raw):
    return ca(da.OpticalImage(reg_h(da.OpticalImage(prepare(raw))).img))


def in_hbm(raw):
    d = da.DeviceArray.from_host(raw)
    out = ca(da.OpticalImage(reg_d(da.OpticalImage(prepare(d))).img))
    return out.img.get()


keep = []
for name, f in (("operator by operator through host arrays", through_host), ("pixels in HBM from the first stage on", in_hbm)):
    for rep in range(4):
        t0 = time.perf_counter(); keep.append(f(raw_frame)); dt = (time.perf_counter() - t0) * 1e3
        print(f"{name:45s} {dt:8.1f} ms per frame", flush=True)
print("same result:", np.array_equal(keep[3].img, keep[7]))

# stage by stage (every call returns with its stream synchronised)
def T(): return time.perf_counter()
for rep in range(3):
    t0 = T(); d = da.DeviceArray.from_host(raw_frame); t1 = T()
    c = balance.apply_balance(d); t2 = T()
    u = curvature(c); t3 = T()
    ta = reg_d.image_registration.translation_analysis
    img = da.OpticalImage(u); ta.load_image(img); t4 = T()
    ta.find_translation(); t5 = T()
    al = ta.translate_image(reverse=False); t6 = T()
    con = ca(da.OpticalImage(al.img)); t7 = T()
    res = con.img.get(); t8 = T()
    ms = [1e3 * (b - a) for a, b in zip((t0, t1, t2, t3, t4, t5, t6, t7), (t1, t2, t3, t4, t5, t6, t7, t8))]
    print("upload %.1f  colour %.1f  curvature %.1f  load %.1f  find_translation %.1f  translate_image %.1f  concentration %.1f  download %.1f  = %.1f ms"
          % (*ms, sum(ms)), flush=True)
