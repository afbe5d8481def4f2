"""debug: the dp2 test's two ranks with a SIGSEGV/SIGABRT backtrace handler in each"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.multiprocessing as mp


def run(rank, world, port, out):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    h = ctypes.CDLL(os.path.join(ROOT, "scripts", "dbg", "libabrt_trace.so"))
    import test_gpu_dp2 as D
    import madiff_amd.trainer as TR
    import faulthandler; faulthandler.enable(all_threads=True)
    real = TR.train_step
    def traced(*a, **k):
        print(f"[{rank}] train_step begin", flush=True)
        r = real(*a, **k)
        torch.cuda.synchronize()
        print(f"[{rank}] train_step end", flush=True)
        return r
    D_steps_globals = D._steps.__globals__
    TR.train_step = traced
    import madiff_amd.ops as OPS
    real_gc = OPS.graph_capture
    class Wrap:
        def __init__(self, g): self.cm = real_gc(g)
        def __enter__(self):
            print(f"[{rank}] capture enter...", flush=True); r = self.cm.__enter__(); print(f"[{rank}] capture entered", flush=True); return r
        def __exit__(self, *a):
            print(f"[{rank}] capture exit... order={getattr(OPT[0], '_graph_order', None)}", flush=True); r = self.cm.__exit__(*a); print(f"[{rank}] capture exited", flush=True); return r
    OPS.graph_capture = Wrap
    OPT = [None]
    real_bgc = TR.FlatAdamEMA.begin_graph_capture
    def bgc(self):
        OPT[0] = self
        return real_bgc(self)
    TR.FlatAdamEMA.begin_graph_capture = bgc
    D._run(rank, world, port, out)
    print(f"[{rank}] done", flush=True)


if __name__ == "__main__":
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(run, args=(2, 29711, out), nprocs=2, join=True)
    print("both ranks OK")
