"""1x1-convolution GEMMs at sampler sizes: this repository's per-tap kernel (ops.conv_gemm, CONV1) against torch.matmul
(hipBLASLt), both as 20 launches inside one graph replay."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madiff_amd import ops
ops.set_precision("bf16")
dev = torch.device("cuda:0")
def graph_us(fn, iters=20):
    for _ in range(3): fn()
    g = torch.cuda.CUDAGraph(); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * iters)
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
with torch.no_grad():
    for name, T, ci, co in [("qkv T=6", 6, 1024, 1536), ("qkv T=12", 12, 512, 768), ("res 2048->512 T=6", 6, 2048, 512),
                            ("res 512->1024 T=6", 6, 512, 1024), ("res 1024->128 T=12", 12, 1024, 128), ("res 128->512 T=12", 12, 128, 512)]:
        M = nseq * T
        x = torch.randn(M, ci, device=dev).to(torch.bfloat16)
        w = torch.randn(co, ci, 1, device=dev) / ci ** 0.5
        b = torch.zeros(co, device=dev)
        wt = w[:, :, 0].t().contiguous().to(torch.bfloat16)
        bb = b.to(torch.bfloat16)
        own = graph_us(lambda: ops.conv_gemm(x, [w], [b], ops.CONV1, T))
        libt = graph_us(lambda: torch.addmm(bb, x, wt))
        fl = 2.0 * M * ci * co
        print(f"{name:22s} M {M:6d} {ci:5d}->{co:5d}: own {own:7.1f} us {fl/own/1e6:6.0f} TF | addmm {libt:7.1f} us {fl/libt/1e6:6.0f} TF", flush=True)
