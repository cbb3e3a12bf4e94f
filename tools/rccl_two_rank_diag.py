"""Diagnostic: does RCCL's bootstrap work between two processes of this box?  Both ranks use GPU 0, so the init is
expected to fail with RCCL's duplicate-GPU error AFTER the ranks have found each other (which is what is being tested)."""
import multiprocessing as mp, os, sys, time
sys.path.insert(0, ".")

def worker(rank, port, q):
    os.environ["MASTER_PORT"] = str(port)
    os.environ["NCCL_DEBUG"] = os.environ.get("NCCL_DEBUG", "WARN")
    from darsia_amd.parallel import Comm
    t0 = time.time()
    try:
        c = Comm.rccl(rank, 2, device=0)
        q.put((rank, "ok", time.time() - t0))
        c.close()
    except Exception as e:  # noqa: BLE001
        q.put((rank, f"error: {e}"[:300], time.time() - t0))

if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, 29881, q)) for r in range(2)]
    [p.start() for p in ps]
    for p in ps:
        p.join(120)
    while not q.empty():
        print(q.get())
    for p in ps:
        if p.is_alive():
            print("still alive (hung): terminating"); p.terminate()
