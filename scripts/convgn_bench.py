"""Fused conv+GN+Mish launch vs the plain all-taps convolution on the T2 1024->1024 layer (N sequences)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madiff_amd import ops
ops.set_precision("bf16")
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
SH = [(1024, 6, 1024, 1024), (8192, 6, 1024, 1024), (8192, 12, 512, 512), (8192, 24, 128, 128)]
if len(sys.argv) > 1:
    SH = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for N, T, ci, co in SH:
    M = N * T
    x = torch.randn(M, ci, device=dev).to(torch.bfloat16)
    w = torch.randn(co, ci, 5, device=dev) / (ci * 5) ** 0.5
    b = torch.randn(co, device=dev); gamma = torch.rand(co, device=dev) + 0.5; beta = torch.randn(co, device=dev) * 0.1
    rb = torch.randn(N, co, device=dev)
    Wf = ops._packed([w], ops.CONV5, False)
    y = torch.empty(M, co, dtype=torch.bfloat16, device=dev); out = torch.empty_like(y)
    stats = torch.empty(2, N * 8, device=dev)
    flops = 2.0 * M * ci * co * 5
    res = []
    for tile in (14, 15, 16):
        t = timeit(lambda: ops.lib.call_rc("madiff_gemm_fwd", 1, 0, x.data_ptr(), ci, None, 0, ci, M, ci, 5, T, T, 1, 2, 0, Wf.data_ptr(),
                                          Wf.stride(0), co, co, b.data_ptr(), None, 0, y.data_ptr(), co, 0, 0, 0, tile, ops._stream()))
        res.append("plain t%d %.1f us %.0f TF" % (tile, t, flops / t / 1e6))
    for variant in (0, 1, 2):
        for wy in (1, 0):
            t = timeit(lambda: ops.lib.call_rc("madiff_conv_gn_mish_fwd", x.data_ptr(), ci, None, 0, ci, M, ci, T, Wf.data_ptr(), Wf.stride(0),
                                               co, b.data_ptr(), y.data_ptr() if wy else None, co, out.data_ptr(), co, gamma.data_ptr(),
                                               beta.data_ptr(), stats[0].data_ptr() if wy else None, stats[1].data_ptr() if wy else None,
                                               8, 1e-5, 1, rb.data_ptr(), co, None, 0, variant, ops._stream()))
            res.append("fused v%d y=%d %.1f us %.0f TF" % (variant, wy, t, flops / t / 1e6))
    gnt = timeit(lambda: ops._gn_forward(y, gamma, beta, rb, None, N, T, co, 8))
    print(f"N={N} T={T} {ci}->{co}: " + " | ".join(res) + f" | standalone GN {gnt:.1f} us", flush=True)
