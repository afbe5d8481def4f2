"""Probe: elementwise kernels over the same 6144 x 1024 bf16 activation as the big GroupNorm layer (kernel trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madiff_amd import ops
ops.set_precision("bf16")
dev = torch.device("cuda:0")
y = torch.randn(6144, 1024, device=dev).to(torch.bfloat16)
out = torch.empty_like(y)
for _ in range(20):
    ops.lib.call("madiff_act_fwd", 1, 2, y.data_ptr(), out.data_ptr(), y.numel(), ops._stream())   # our Mish
for _ in range(20):
    out.copy_(y)                                                                                     # plain copy
for _ in range(20):
    torch.nn.functional.mish(y)
for _ in range(20):
    torch.relu(y)
torch.cuda.synchronize()
