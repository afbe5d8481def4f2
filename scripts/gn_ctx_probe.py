"""Why does gn_mish_bwd take ~40 us inside the step and ~18 us alone?  Times the SAME launch (event-record nodes inside one
hipGraph, as bench.py does) alone, behind an MFMA-heavy kernel, and behind an HBM-heavy copy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madiff_amd import ops
ops.set_precision("bf16")
dev = torch.device("cuda:0")
N, T, C = 1024, 6, 1024
R = 24
ys = [torch.randn(N * T, C, device=dev).to(torch.bfloat16) for _ in range(R)]
ds = [torch.randn(N * T, C, device=dev).to(torch.bfloat16) for _ in range(R)]
gamma = torch.rand(C, device=dev) + 0.5
beta = torch.randn(C, device=dev) * 0.1
st = [ops._gn_forward(y, gamma, beta, None, None, N, T, C, 8) for y in ys]
a = torch.randn(4096, 4096, device=dev).to(torch.bfloat16)
big = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
big2 = torch.empty_like(big)
for _ in range(300):
    a @ a


def filler(kind):
    if kind == "mfma":
        a @ a
    elif kind == "mfma3":
        for _ in range(3):
            a @ a
    elif kind == "copy":
        big2.copy_(big)


for kind in ("none", "mfma", "mfma3", "copy", "none"):
    def run():
        for y, d, (_, stats, geom) in zip(ys, ds, st):
            filler(kind)
            ops._gn_backward(y, gamma, beta, stats, geom, d, conv_bias=beta)
    run()
    torch.cuda.synchronize()
    gt = ops.GraphKernelTimer()
    ops.GRAPH_TIMER = gt
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    ops.GRAPH_TIMER = None
    for _ in range(3):
        g.replay()
        gt.collect()
    s = gt.summary()["gn_mish_bwd_rb_kernel"]
    print(f"filler {kind:6s}: gn_mish_bwd {1e3 * s['ms'] / s['launches']:6.1f} us/launch over {s['launches']} launches", flush=True)
    gt.close()
