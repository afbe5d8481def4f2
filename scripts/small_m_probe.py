"""Small-batch regime: which kernel variant serves a k5 conv layer best when M is small (batch 64: 192 sequences)?
Times every tile of madiff_gemm_fwd (0..13 per-tap, 14..16 all-taps) and the fused conv+GN variants, launches captured in
a graph (Python launch overhead excluded)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madiff_amd import ops
ops.set_precision("bf16")
dev = torch.device("cuda:0")


def timeit(fn, n=20):
    if fn() != 0:
        return None
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
    return e0.elapsed_time(e1) / n * 1e3


a = torch.randn(4096, 4096, device=dev)
for _ in range(100):
    a @ a
SHAPES = [(192, 6, 1024, 1024), (192, 12, 512, 512), (192, 24, 128, 128), (96, 6, 1024, 1024)]
if len(sys.argv) > 1:
    n_ = int(sys.argv[1])
    SHAPES = [(n_, 6, 1024, 1024), (n_, 6, 512, 1024), (n_, 12, 512, 512), (n_, 12, 128, 512), (n_, 24, 128, 128)]
for N, T, ci, co in SHAPES:
    M = N * T
    x = torch.randn(M, ci, device=dev).to(torch.bfloat16)
    w = torch.randn(co, ci, 5, device=dev) / (ci * 5) ** 0.5
    b = torch.randn(co, device=dev); gamma = torch.rand(co, device=dev) + 0.5; beta = torch.randn(co, device=dev) * 0.1
    rb = torch.randn(N, co, device=dev)
    Wf = ops._packed([w], ops.CONV5, False)
    y = torch.empty(M, co, dtype=torch.bfloat16, device=dev); out = torch.empty_like(y)
    stats = torch.empty(2, N * 8, device=dev)
    res = []
    for tile in range(ops.lib.tile_count()):
        t = timeit(lambda: ops.lib.call_rc("madiff_gemm_fwd", 1, 0, x.data_ptr(), ci, None, 0, ci, M, ci, 5, T, T, 1, 2, 0, Wf.data_ptr(),
                                           Wf.stride(0), co, co, b.data_ptr(), None, 0, y.data_ptr(), co, 0, 0, 0, tile, ops._stream()))
        if t is not None:
            res.append((t, "tile %d (%s)" % (tile, "x".join(map(str, ops.lib.tile_info(tile)[:3])))))
    for variant in (1, 2):
        t = timeit(lambda: ops.lib.call_rc("madiff_conv_gn_mish_fwd", x.data_ptr(), ci, None, 0, ci, M, ci, T, Wf.data_ptr(), Wf.stride(0),
                                           co, b.data_ptr(), y.data_ptr(), co, out.data_ptr(), co, gamma.data_ptr(),
                                           beta.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(),
                                           8, 1e-5, 1, rb.data_ptr(), co, None, 0, variant, ops._stream()))
        if t is not None:
            res.append((t, "FUSED conv+GN v%d" % variant))
    gnt = timeit(lambda: (ops._gn_forward(y, gamma, beta, rb, None, N, T, co, 8), 0)[1])
    res.sort()
    print(f"N={N} T={T} {ci}->{co} (M={M}); standalone GN {gnt:.1f} us; fastest first:", flush=True)
    for t, name in res[:8]:
        print(f"    {t:7.1f} us  {name}")
