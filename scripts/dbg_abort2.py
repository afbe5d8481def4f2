"""debug: the graphed-train-step tests under pytest (-s, no faulthandler) with a SIGABRT backtrace handler"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
h = ctypes.CDLL(os.path.join(ROOT, "scripts", "dbg", "libabrt_trace.so"))
import pytest
h.install_abrt_trace()
sys.exit(pytest.main(["tests/test_gpu_kernels.py", "-x", "-q", "-s", "-p", "no:faulthandler", "-k", "graphed_train_step"] + sys.argv[1:]))
