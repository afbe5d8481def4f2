"""Diagnostic: the data-parallel capture with the branches rejoined into the main stream at every bucket.  Dumps the
captured graph (hipGraphDebugDotPrint on the graph under capture) right before hipStreamEndCapture, and installs a native
backtrace handler for the crash."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
h = ctypes.CDLL(os.path.join(ROOT, "scripts", "dbg", "libabrt_trace.so"))
import torch
import test_gpu_kernels as K
h.install_abrt_trace()
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
topo = sys.argv[1] if len(sys.argv) > 1 else "rejoin_per_bucket"
out = os.path.join(ROOT, "gpurun_out", f"r2_graph_{topo}.dot")
real_end = torch.cuda.CUDAGraph.capture_end


def capture_end(self):
    st, gid, graph = ctypes.c_int(), ctypes.c_ulonglong(), ctypes.c_void_p()
    deps, nd = ctypes.c_void_p(), ctypes.c_size_t()
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = hip.hipStreamGetCaptureInfo_v2(s, ctypes.byref(st), ctypes.byref(gid), ctypes.byref(graph), ctypes.byref(deps),
                                        ctypes.byref(nd))
    print("capture info rc", rc, "status", st.value, "ndeps", nd.value, flush=True)
    rc = hip.hipGraphDebugDotPrint(graph, out.encode(), ctypes.c_uint(0))
    print("dot print rc", rc, out, flush=True)
    return real_end(self)


torch.cuda.CUDAGraph.capture_end = capture_end
K.test_graphed_train_step_matches_eager_steps(True, topo)
print("OK", flush=True)
