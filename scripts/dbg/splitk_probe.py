"""Does the all-taps kernel's reduction split engage at planner sizes, and what does it buy? (60 sequences)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for T, ci, co in [(6, 1024, 1024), (6, 2048, 512), (12, 512, 512)]:
    M = N * T
    x = torch.randn(M, ci, device=dev).to(torch.bfloat16)
    w = torch.randn(co, ci, 5, device=dev) / (ci * 5) ** 0.5
    b = torch.randn(co, device=dev)
    Wf = ops._packed([w], ops.CONV5, False)
    y = torch.empty(M, co, dtype=torch.bfloat16, device=dev)
    for tile in (13, 14, 15, 16):
        t = timeit(lambda: ops.lib.call_rc("madiff_gemm_fwd", 1, 0, x.data_ptr(), ci, None, 0, ci, M, ci, 5, T, T, 1, 2, 0, Wf.data_ptr(),
                                           Wf.stride(0), co, co, b.data_ptr(), None, 0, y.data_ptr(), co, 0, 0, 0, tile, ops._stream()))
        print(f"N={N} T={T} {ci}->{co} tile {tile} {ops.lib.tile_info(tile)[:3]}: {t} us  (MADIFF_CONV_SPLITK={os.environ.get('MADIFF_CONV_SPLITK','1')}, slots used {ops.lib._lib.madiff_splitk_slots_used()})", flush=True)
