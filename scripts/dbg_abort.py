"""debug: run the 1-rank RCCL graphed train step with a SIGABRT backtrace handler installed"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
h = ctypes.CDLL(os.path.join(ROOT, "scripts", "dbg", "libabrt_trace.so"))
import torch
import test_gpu_kernels as K
h.install_abrt_trace()
mode = sys.argv[1] if len(sys.argv) > 1 else "global"
if mode != "global":
    import madiff_amd.trainer as TR
    real = torch.cuda.graph
    TR.torch.cuda.graph = lambda g, **kw: real(g, capture_error_mode=mode, **kw)
if "seq" in sys.argv:
    K.test_graphed_train_step_matches_eager_steps(False)
    print("single done", flush=True)
K.test_graphed_train_step_matches_eager_steps(True)
print("dp graphed step OK in mode", mode, flush=True)
