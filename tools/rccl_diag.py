import os, sys
sys.path.insert(0, ".")
if os.environ.get("WITH_TORCH"): import torch
os.environ["MASTER_PORT"] = "29777"
os.environ["NCCL_DEBUG"] = "WARN"
from darsia_amd.parallel import Comm
from darsia_amd import _lib
_lib.require_gpu()
def libs():
    maps = open("/proc/self/maps").read()
    return sorted({l.split()[-1] for l in maps.splitlines() if any(k in l for k in ("rccl", "amdhip", "hsa-runtime", "darsia"))})
print("before rccl:", libs())
try:
    c = Comm.rccl(0, 1, device=0)
    print("ok", c.allreduce([1.0]))
    c.close()
except Exception as e:
    print("FAILED", e)
print("after:", libs())
