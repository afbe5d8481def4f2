"""Micro-benchmark of the GroupNorm+Mish kernels (forward / backward) on the bench workload's layer shapes:
achieved HBM GB/s = algorithmic bytes (fwd: read y [+res], write out; bwd: read y, dout, write dy) / time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madiff_amd import ops
ops.set_precision("bf16")
dev = torch.device("cuda:0")
N = 1024
shapes = [(6, 1024, 14), (6, 512, 4), (12, 512, 6), (12, 128, 4), (24, 128, 5)]   # (T, C, blocks per forward ~)


def timeit(fn, n=30):
    """n launches captured into one hipGraph and replayed (the Python wrapper costs more than these kernels)"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


tot = [0.0, 0.0]
for T, C, cnt in shapes:
    y = torch.randn(N * T, C, device=dev).to(torch.bfloat16)
    res = torch.randn(N * T, C, device=dev).to(torch.bfloat16)
    rb = torch.randn(N, C, device=dev)
    gamma = torch.rand(C, device=dev) + 0.5
    beta = torch.randn(C, device=dev) * 0.1
    dout = torch.randn(N * T, C, device=dev).to(torch.bfloat16)
    nbytes = y.numel() * 2
    for mode, kw in (("plain", {}), ("temb", {"rowbias": rb}), ("res", {"res": res})):
        out, stats, geom = ops._gn_forward(y, gamma, beta, kw.get("rowbias"), kw.get("res"), N, T, C, 8)
        tf = timeit(lambda: ops._gn_forward(y, gamma, beta, kw.get("rowbias"), kw.get("res"), N, T, C, 8))
        tb = timeit(lambda: ops._gn_backward(y, gamma, beta, stats, geom, dout, conv_bias=beta))
        fb = nbytes * (3 if mode == "res" else 2)
        bb = nbytes * 3
        print(f"T={T:2d} C={C:4d} {mode:5s} fwd {tf * 1e3:6.1f} us {fb / tf / 1e6:7.0f} GB/s | bwd {tb * 1e3:6.1f} us {bb / tb / 1e6:7.0f} GB/s",
              flush=True)
    tot[0] += cnt * tf
    tot[1] += cnt * tb
print(f"approx per-step totals (ms): fwd {tot[0]:.3f} bwd {tot[1]:.3f}")
