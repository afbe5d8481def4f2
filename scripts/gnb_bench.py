"""Standalone timing of the k5 data gradient at training size: plain (madiff_gemm_dgrad_wt, all-taps tile), with the
GroupNorm + Mish backward of the preceding block as its epilogue (madiff_conv_dgrad_gn_bwd), and the chained form
(madiff_conv_dgrad_gn_bwd_chain); 20 launches per graph replay.  usage: gnb_bench.py [n_seq=1024] [T=6] [C=1024]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madiff_amd import ops
ops.set_precision("bf16")
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 6
C = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
Co, G = C, 8
M = N * T
def graph_us(fn, iters=20):
    assert fn() == 0
    g = torch.cuda.CUDAGraph(); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * iters)
dz = torch.randn(M, Co, device=dev).to(torch.bfloat16)
y0 = torch.randn(M, C, device=dev).to(torch.bfloat16)
add = torch.randn(M, C, device=dev).to(torch.bfloat16)
w = torch.randn(Co, C, 5, device=dev) / (5 * C) ** 0.5
Wf = ops._packed([w], ops.CONV5, False)
gamma = torch.rand(C, device=dev) + 0.5; beta = torch.randn(C, device=dev) * 0.1
mean = torch.zeros(N * G, device=dev); rstd = torch.ones(N * G, device=dev)
dX = torch.empty(M, C, dtype=torch.bfloat16, device=dev); dH = torch.empty_like(dX)
dgamma, dbeta, dbias = (torch.zeros(C, device=dev) for _ in range(3))
drb = torch.empty(N, C, device=dev)
ops._stream()
L = ops.lib.call_rc
for tile in (14, 16):
    t = graph_us(lambda: L("madiff_gemm_dgrad_wt", dz.data_ptr(), Co, M, Co, 5, T, T, 1, 2, 1, Wf.data_ptr(), Wf.stride(0), C, 0, C, C,
                           dX.data_ptr(), C, tile, 0, ops._stream()))
    print(f"plain data gradient, tile {tile}: {t:.1f} us")
for v in (2,):
    t = graph_us(lambda: L("madiff_conv_dgrad_gn_bwd", dz.data_ptr(), Co, M, Co, T, Wf.data_ptr(), Wf.stride(0), C, C, y0.data_ptr(), C,
                           gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(), G, dX.data_ptr(), C, dgamma.data_ptr(),
                           dbeta.data_ptr(), dbias.data_ptr(), drb.data_ptr(), C, v, ops._stream()))
    print(f"+ GroupNorm backward epilogue (variant {v}): {t:.1f} us")
    for wa in (False, True):
        t = graph_us(lambda: L("madiff_conv_dgrad_gn_bwd_chain", dz.data_ptr(), Co, M, Co, T, Wf.data_ptr(), Wf.stride(0), C, C,
                               y0.data_ptr(), C, gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(), G, dX.data_ptr(),
                               C, dgamma.data_ptr(), dbeta.data_ptr(), dbias.data_ptr(), add.data_ptr() if wa else None, C,
                               dH.data_ptr(), C, v, ops._stream()))
        print(f"+ chained form (variant {v}, addend {wa}): {t:.1f} us")
