"""Time one all-taps conv variant on the T2 1024->1024 layer.  usage: ct_time.py <tile> <fwd|dgrad> [N=1024]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madiff_amd import ops
ops.set_precision("bf16")
tile, mode = int(sys.argv[1]), sys.argv[2]
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
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
assert f() == 0
t0 = time.time()
while time.time() - t0 < 1.0:
    for _ in range(50):
        f()
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    f()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
print(f"tile {tile} {mode} N={N}: {us:.1f} us  {2.0 * M * ci * co * 5 / us / 1e6:.0f} TF (nominal)", flush=True)
