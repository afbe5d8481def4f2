"""Micro-benchmark of the implicit-GEMM kernels on the layer shapes of the bench workload (B=256, A=4 -> N=1024)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madiff_amd import ops
ops.set_precision("bf16")
dev = torch.device("cuda:0")
N = 1024
# (name, T, Cin, Cout, spec, count per forward)
shapes = [("T2 1024->1024 k5", 6, 1024, 1024, ops.CONV5, 7), ("T2 2048->512 k5", 6, 2048, 512, ops.CONV5, 1),
          ("T2 512->1024 k5", 6, 512, 1024, ops.CONV5, 1), ("T2 512->512 k5", 6, 512, 512, ops.CONV5, 3),
          ("T1 512->512 k5", 12, 512, 512, ops.CONV5, 3), ("T1 128->512 k5", 12, 128, 512, ops.CONV5, 1),
          ("T1 1024->128 k5", 12, 1024, 128, ops.CONV5, 1), ("T1 128->128 k5", 12, 128, 128, ops.CONV5, 3),
          ("T0 128->128 k5", 24, 128, 128, ops.CONV5, 4),
          ("T2 1024->1536 1x1", 6, 1024, 1536, ops.CONV1, 2), ("T1 512->768 1x1", 12, 512, 768, ops.CONV1, 1),
          ("T2 512->1024 1x1", 6, 512, 1024, ops.CONV1, 1), ("T2 2048->512 1x1", 6, 2048, 512, ops.CONV1, 1),
          ("T1 128->512 1x1", 12, 128, 512, ops.CONV1, 1), ("T1 1024->128 1x1", 12, 1024, 128, ops.CONV1, 1)]
tot = [0.0, 0.0, 0.0]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, T, ci, co, spec, cnt in shapes:
    x = torch.randn(N * T, ci, device=dev).to(torch.bfloat16).requires_grad_(True)
    w = (torch.randn(co, ci, spec.taps, device=dev) / (ci * spec.taps) ** 0.5).requires_grad_(True)
    b = torch.zeros(co, device=dev, requires_grad=True)
    y = ops.conv_gemm(x, [w], [b], spec, T)
    gy = torch.randn_like(y)
    flops = 2.0 * N * T * ci * co * spec.taps
    tf = timeit(lambda: ops.conv_gemm(x.detach(), [w], [b], spec, T))
    # backward pieces, timed through the raw launches
    Wb = ops._packed([w], spec, True)
    dx = torch.empty(N * T, ci, dtype=torch.bfloat16, device=dev)
    def dgrad():
        ops.lib.call("madiff_gemm_fwd", 1, 0, gy.data_ptr(), gy.stride(0), None, 0, gy.shape[1], N * T, gy.shape[1], spec.taps,
                     T, T, 1, spec.pad, 1, Wb.data_ptr(), Wb.stride(0), ci, ci, None, None, 0, dx.data_ptr(), dx.stride(0), 0, 0, 0, ops._stream())
    td = timeit(dgrad)
    dW = torch.empty(co, spec.taps * ci, dtype=torch.float32, device=dev)
    tw = timeit(lambda: ops._wgrad(gy, x.detach(), None, ci, N * T, ci, spec, T, T, dW, co))
    tot[0] += cnt * tf; tot[1] += cnt * td; tot[2] += cnt * tw
    print(f"{name:22s} x{cnt} fwd {tf*1e3:7.1f} us {flops/tf/1e9:7.1f} TF | dgrad {td*1e3:7.1f} us {flops/td/1e9:7.1f} TF | wgrad {tw*1e3:7.1f} us {flops/tw/1e9:7.1f} TF", flush=True)
print(f"weighted totals per step (ms): fwd {tot[0]:.3f} dgrad {tot[1]:.3f} wgrad {tot[2]:.3f}")
