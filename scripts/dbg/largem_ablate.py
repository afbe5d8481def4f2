"""Where does a K-step of the per-tap kernel go at LARGE M?  (diagnosis builds: -DMADIFF_ABL bit 0 no MFMA, 1 no fragment reads
after the first step, 2 no DMA after the prologue, 3 no barrier; loaded through MADIFF_LIB_PATH.)  1x1 layers at sampler sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from madiff_amd import ops
ops.set_precision("bf16")
dev = torch.device("cuda:0")
T = 6
tag = os.environ.get("ABL", "0")
for N, ci, co in [(6144, 1024, 1536), (6144, 2048, 512)]:
    M = N * T
    x = torch.randn(M, ci, device=dev).to(torch.bfloat16)
    w = torch.randn(co, ci, 1, device=dev) / ci ** 0.5
    b = torch.randn(co, device=dev)
    Wf = ops._packed([w], ops.CONV1, False)
    y = torch.empty(M, co, dtype=torch.bfloat16, device=dev)
    out = []
    for tile in (0, 1, 2, 4, 6):
        fn = lambda: ops.lib.call_rc("madiff_gemm_fwd", 1, 0, x.data_ptr(), ci, None, 0, ci, M, ci, 1, T, T, 1, 0, 0, Wf.data_ptr(),
                                     Wf.stride(0), co, co, b.data_ptr(), None, 0, y.data_ptr(), co, 0, 0, 0, tile, ops._stream())
        assert fn() == 0
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(10):
                fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
        out.append(f"t{tile} {e0.elapsed_time(e1) / 20 * 1e3:6.1f}")
    print(f"abl {tag:>2s} {M}x{ci}->{co}: " + "  ".join(out) + " us", flush=True)
