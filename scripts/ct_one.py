"""One all-taps conv variant on the T2 1024->1024 layer, launched a few times (for scripts/pmc_ct.sh).
usage: python scripts/ct_one.py <tile 14|15|16> <fwd|dgrad> [reps=3] [N=1024]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madiff_amd import ops
ops.set_precision("bf16")
tile, mode = int(sys.argv[1]), sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
N = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
dev = torch.device("cuda:0")
T, ci, co = 6, 1024, 1024
M = N * T
x = torch.randn(M, ci, device=dev).to(torch.bfloat16)
w = torch.randn(co, ci, 5, device=dev) / (ci * 5) ** 0.5
Wf = ops._packed([w], ops.CONV5, False)
y = torch.empty(M, co, dtype=torch.bfloat16, device=dev)
st = ops._stream()
if mode == "fwd":
    f = lambda: ops.lib.call_rc("madiff_gemm_fwd", 1, 0, x.data_ptr(), ci, None, 0, ci, M, ci, 5, T, T, 1, 2, 0, Wf.data_ptr(),
                                Wf.stride(0), co, co, None, None, 0, y.data_ptr(), co, 0, 0, 0, tile, st)
else:
    f = lambda: ops.lib.call_rc("madiff_gemm_dgrad_wt", x.data_ptr(), ci, M, co, 5, T, T, 1, 2, 1, Wf.data_ptr(), Wf.stride(0), ci,
                                0, ci, ci, y.data_ptr(), co, tile, 0, st)
for _ in range(reps):
    assert f() == 0
torch.cuda.synchronize()
