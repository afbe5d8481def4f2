"""Launch ONE k5 layer (T2 1024->1024, N=1024) a few times with a chosen GEMM tile, for rocprofv3 --pmc passes.
usage: python scripts/ct_one.py <tile> <fwd|dgrad> [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madiff_amd import ops
ops.set_precision("bf16")
dev = torch.device("cuda:0")
tile, mode = int(sys.argv[1]), sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
N, T, ci, co = [int(v) for v in os.environ.get("SHAPE", "1024,6,1024,1024").split(",")]
M = N * T
x = torch.randn(M, ci, device=dev).to(torch.bfloat16)
w = torch.randn(co, ci, 5, device=dev) / (ci * 5) ** 0.5
Wf = ops._packed([w], ops.CONV5, False)
y = torch.empty(M, co, dtype=torch.bfloat16, device=dev)
gy = torch.randn(M, co, device=dev).to(torch.bfloat16)
dx = torch.empty(M, ci, dtype=torch.bfloat16, device=dev)
torch.cuda.synchronize()
for _ in range(reps):
    if mode == "fwd":
        ops.lib.call_rc("madiff_gemm_fwd", 1, 0, x.data_ptr(), ci, None, 0, ci, M, ci, 5, T, T, 1, 2, 0, Wf.data_ptr(),
                        Wf.stride(0), co, co, None, None, 0, y.data_ptr(), co, 0, 0, 0, tile, ops._stream())
    else:
        ops.lib.call_rc("madiff_gemm_dgrad_wt", gy.data_ptr(), co, M, co, 5, T, T, 1, 2, 1, Wf.data_ptr(), Wf.stride(0), ci, 0,
                        ci, ci, dx.data_ptr(), ci, tile, 0, ops._stream())
torch.cuda.synchronize()
