#!/usr/bin/env python3
"""Which rows of a periodic slab run differ between trimmed and untrimmed halo launches?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import subprocess, json

def run(tune, rows, n, k, W=2000, parts=None, flags=0):
    code = f"""
import os, sys, numpy as np
sys.path.insert(0, '.')
os.environ['MASTER_PORT']='29655'
if {tune}: os.environ['DXH_FUSED_TUNE']=str({tune})
from darsia_amd.parallel import Comm, SlabTVD, make_plan, periodic_slab
from darsia_amd.synthetic import synthetic_image
img = synthetic_image({rows}, {W}, 3)
comm = Comm.rccl(0,1,device=0)
plan = make_plan({rows},1,0,{k},periodic=True)
s = SlabTVD(plan, periodic_slab(img, plan), mu=0.1, halo_iters={k}, comm=comm, flags={flags})
done=0
for part in {parts or [n]}:
    done += s.run(part, more_to_come=done+part<{n}, asynchronous=True)
np.save('/tmp/trim_{tune}.npy', s.engine.download())
s.close(); comm.close()
"""
    subprocess.run([sys.executable, "-c", code], check=True, stdout=subprocess.DEVNULL)
    return np.load(f"/tmp/trim_{tune}.npy")

for (rows, n, k, parts, flags) in ((300, 8, 8, None, 0), (300, 16, 8, None, 0), (300, 20, 20, None, 0), (300, 40, 20, None, 0), (300, 40, 20, None, 32), (1000, 100, 20, [60, 40], 0)):
    a = run(0, rows, n, k, parts=parts, flags=flags)
    b = run(131072, rows, n, k, parts=parts, flags=flags)
    top, bot = k, k + 1
    own = slice(top, a.shape[0] - bot)
    bad = np.flatnonzero((a[own] != b[own]).any(axis=(1, 2)))
    print(dict(rows=rows, n=n, k=k, parts=parts, flags=flags), "owned rows differing:", len(bad), bad[:10], bad[-10:] if len(bad) else "")
