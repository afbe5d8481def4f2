"""GroupNorm+Mish backward on operands that come from HBM, not from the 256 MB memory-side cache: R rotating operand sets
(R x 3 tensors > 1 GB), one launch per set, all captured in one hipGraph.  (scripts/gn_bench.py reuses one set, which stays
cache-resident: its figures are the cached ones.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madiff_amd import ops
ops.set_precision("bf16")
dev = torch.device("cuda:0")
N = 1024
shapes = [(6, 1024, 12), (6, 512, 0), (12, 512, 9), (12, 128, 0), (24, 128, 4)]   # (T, C, launches per step)
x = torch.randn(4096, 4096, device=dev)
for _ in range(200):                       # clocks up
    x @ x
tot = 0.0
for T, C, cnt in shapes:
    nb = N * T * C * 2
    R = max(4, int(1.2e9 / (3 * nb)))
    ys = [torch.randn(N * T, C, device=dev).to(torch.bfloat16) for _ in range(R)]
    ds = [torch.randn(N * T, C, device=dev).to(torch.bfloat16) for _ in range(R)]
    gamma = torch.rand(C, device=dev) + 0.5
    beta = torch.randn(C, device=dev) * 0.1
    rb = torch.randn(N, C, device=dev)
    for mode, kw in (("plain", {}), ("temb", {"rowbias": rb})):
        st = [ops._gn_forward(y, gamma, beta, kw.get("rowbias"), None, N, T, C, 8) for y in ys]
        sink = torch.zeros(3, C, device=dev)

        def run():
            for y, d, (_, stats, geom) in zip(ys, ds, st):
                ops._gn_backward(y, gamma, beta, stats, geom, d, conv_bias=beta)
        run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            run()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / R
        print(f"T={T:2d} C={C:4d} {mode:5s} R={R:3d} bwd {t * 1e3:6.1f} us/launch (incl. a zero-fill of 3xC floats) {3 * nb / t / 1e6:7.0f} GB/s",
              flush=True)
    tot += cnt * t
    del ys, ds, st
print(f"approx per-step total (ms): bwd {tot:.3f}")
