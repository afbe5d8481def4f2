"""One weight-gradient variant on one layer shape, launched a few times (for rocprofv3 --pmc / --kernel-trace runs).
usage: python scripts/wgrad_probe.py <tile> <splits> [T=6] [Cin=1024] [Cout=1024] [taps=5] [reps=10]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madiff_amd import ops
ops.set_precision("bf16")
a = [int(v) for v in sys.argv[1:]]
tile, splits = a[0], a[1]
T, ci, co, taps, reps = (a[2:] + [6, 1024, 1024, 5, 10][len(a) - 2:])
dev = torch.device("cuda:0")
N = 1024
M = N * T
x = torch.randn(M, ci, device=dev).to(torch.bfloat16)
gy = torch.randn(M, co, device=dev).to(torch.bfloat16)
dW = torch.zeros(co, taps * ci, dtype=torch.float32, device=dev)
st = ops._stream()
pad = 2 if taps == 5 else 0
f = lambda: ops.lib.call_rc("madiff_gemm_wgrad", 1, gy.data_ptr(), co, co, x.data_ptr(), ci, None, 0, ci, M, ci, taps, T, T, 1,
                            pad, 0, dW.data_ptr(), dW.stride(0), co, 1, tile, splits, st)
assert f() == 0
import time
t0 = time.time()
while time.time() - t0 < 1.5:      # clocks settle only after a second or two of back-to-back launches
    for _ in range(50):
        f()
    torch.cuda.synchronize()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    f()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"tile {tile} splits {splits} T {T} {ci}->{co} taps {taps}: {ms * 1e3:.1f} us, {2.0 * M * ci * co * taps / ms / 1e9:.0f} TF")
