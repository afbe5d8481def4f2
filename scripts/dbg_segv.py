"""debug: one rank of the dp2 test's step sequence (single process) with a SIGSEGV/SIGABRT backtrace handler"""
import ctypes, os, sys, faulthandler
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
h = ctypes.CDLL(os.path.join(ROOT, "scripts", "dbg", "libabrt_trace.so"))
import torch
import test_gpu_dp2 as D
h.install_abrt_trace()
dif, batch, draws = D._setup()
print("setup done", flush=True)
opt, gs = D._steps(dif, batch, draws)
print("single-process steps OK", flush=True)
