"""Attention core forward / backward on the bench shapes, with and without the gamma residual (dgamma atomic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madiff_amd import ops
ops.set_precision("bf16")
dev = torch.device("cuda:0")
def timeit(fn, n=30):
    """n launches captured in one hipGraph (the host cannot keep up with 15 us kernels otherwise)"""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
A = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B, H = 256, 4
print(f"A={A} MADIFF_ATTN_MFMA={os.environ.get('MADIFF_ATTN_MFMA', '1')}", flush=True)
for T, C in ((6, 1024), (12, 512), (24, 128)):
    Cq = C // 4
    M = B * A * T
    qkv = torch.randn(M, 2 * Cq + C, device=dev).to(torch.bfloat16)
    x = torch.randn(M, C, device=dev).to(torch.bfloat16)
    dout = torch.randn(M, C, device=dev).to(torch.bfloat16)
    out = torch.empty_like(x); dqkv = torch.empty_like(qkv)
    gamma = torch.full((1,), 0.5, device=dev); dgamma = torch.zeros(1, device=dev)
    for residual in (1, 0):
        tf = timeit(lambda: ops.lib.call("madiff_attn_fwd", 1, qkv.data_ptr(), qkv.stride(0), x.data_ptr(), x.stride(0), out.data_ptr(),
                                         out.stride(0), gamma.data_ptr(), B, A, T, H, Cq, C, residual, ops._stream()))
        tb = timeit(lambda: ops.lib.call("madiff_attn_bwd", 1, qkv.data_ptr(), qkv.stride(0), dout.data_ptr(), dout.stride(0),
                                         dqkv.data_ptr(), dqkv.stride(0), gamma.data_ptr(), dgamma.data_ptr(), B, A, T, H, Cq, C,
                                         residual, ops._stream()))
        drb = torch.empty(B * A, 2 * Cq + C, device=dev)
        tr = timeit(lambda: ops.lib.call("madiff_attn_bwd_rb", 1, qkv.data_ptr(), qkv.stride(0), dout.data_ptr(), dout.stride(0),
                                         dqkv.data_ptr(), dqkv.stride(0), gamma.data_ptr(), dgamma.data_ptr(), drb.data_ptr(),
                                         drb.stride(0), B, A, T, H, Cq, C, residual, ops._stream()))
        mb = M * (2 * (2 * Cq + C) + C) * 2 / 1e6
        print(f"T={T} C={C} residual={residual}: fwd {tf:.1f} us  bwd {tb:.1f} us ({mb / tb * 1e3:.0f} GB/s)  bwd + time-bias sums {tr:.1f} us",
              flush=True)
