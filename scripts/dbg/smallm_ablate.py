"""Where does a K-step of the per-tap kernel go at small M?  (diagnosis build: MADIFF_DBG_ABLATE bit 0 no MFMA, 1 no fragment
reads, 2 no DMA after the prologue, 3 no barrier.)  1024->1024 k5, 60 sequences x T=6, tile 13 (64x64, 4 stages)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from madiff_amd import ops
ops.set_precision("bf16")
dev = torch.device("cuda:0")
N, T, ci, co = 60, 6, 1024, 1024
M = N * T
x = torch.randn(M, ci, device=dev).to(torch.bfloat16)
w = torch.randn(co, ci, 5, device=dev) / (ci * 5) ** 0.5
b = torch.randn(co, device=dev)
Wf = ops._packed([w], ops.CONV5, False)
y = torch.empty(M, co, dtype=torch.bfloat16, device=dev)
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 13
fn = lambda: ops.lib.call_rc("madiff_gemm_fwd", 1, 0, x.data_ptr(), ci, None, 0, ci, M, ci, 5, T, T, 1, 2, 0, Wf.data_ptr(),
                             Wf.stride(0), co, co, b.data_ptr(), None, 0, y.data_ptr(), co, 0, 0, 0, tile, ops._stream())
assert fn() == 0
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20):
        fn()
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
print(f"ablate {os.environ.get('MADIFF_DBG_ABLATE', '0'):>2s} tile {tile}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)
