"""Developer probe: engine clock / power sampled with rocm-smi while the headline loop runs."""
import subprocess, sys, threading, time
sys.path.insert(0, ".")
import darsia_amd as da
from darsia_amd.synthetic import synthetic_image

def smi():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True).stdout
    keep = [l.strip() for l in out.splitlines() if any(k in l for k in ("sclk", "mclk", "fclk", "Power", "junction", "hotspot"))]
    return " | ".join(k.split("GPU[0]")[-1].strip(" :\t") for k in keep)

print("idle:", smi(), flush=True)
img = synthetic_image(4000, 8000, 3)
st = da.DeviceTVD(img.shape, img.dtype); st.set_coeffs(0.1, 1.0, None, img); st.upload(img); st.run(8)
stop = False
def loop():
    while not stop:
        r = st.run(2000)
        print(f"   loop {r.loop_ms/2000:.4f} ms/iter", flush=True)
th = threading.Thread(target=loop); th.start()
for i in range(12):
    time.sleep(0.7)
    print("busy:", smi(), flush=True)
stop = True; th.join()
