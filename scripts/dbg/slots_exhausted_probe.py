"""Root cause check of a flaky "graph == eager" assertion: with the 16 reduction-split slots used up, a capture runs the
all-taps kernel unsplit while the eager stream (which owns a slot) splits.  usage: python scripts/dbg/slots_exhausted_probe.py [reset]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import torch
from madiff_amd import modules as M, ops
ops.set_precision("bf16")
dev = torch.device("cuda:0")
N, T, C = 6, 6, 1024
x = torch.randn(N * T, C, device=dev).to(torch.bfloat16)
w = (torch.randn(C, C, 5, device=dev) / (5 * C) ** 0.5)
b = torch.zeros(C, device=dev)
g, be = torch.ones(C, device=dev), torch.zeros(C, device=dev)
with torch.no_grad():
    ops.conv_block(x, w, b, g, be, ops.CONV5, T, 8)            # main stream: slot 0 (+ tuning)
    streams = [torch.cuda.Stream() for _ in range(20)]
    for s in streams:
        with torch.cuda.stream(s):
            ops.conv_block(x, w, b, g, be, ops.CONV5, T, 8)
torch.cuda.synchronize()
print("slots used:", ops.lib._lib.madiff_splitk_slots_used(), flush=True)
if "reset" in sys.argv:
    ops.reset_splitk_slots()
    print("after reset:", ops.lib._lib.madiff_splitk_slots_used(), flush=True)
import test_gpu_parity as P
if "notune" in sys.argv:
    ops._AUTOTUNE = False          # every k5 block takes the fused all-taps launch (the reduction split whenever a slot exists)
try:
    P.test_sampler_split_time_projection_matches_one_gemm()
    print("test passed", flush=True)
except AssertionError as e:
    print("test FAILED:", str(e)[:200].replace("\n", " "), flush=True)
