import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from madiff_amd import ops, modules as M
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
ops.set_precision(prec)
dev = torch.device("cuda:0")
for (N, T, C) in [(5, 12, 64), (4, 6, 1024), (40, 12, 64)]:
    torch.manual_seed(0)
    x = torch.randn(N, C, T, device=dev)
    if prec == "bf16":
        x = x.bfloat16().float()
    x.requires_grad_(True)
    gamma = (torch.randn(C, device=dev) * 0.5 + 1).requires_grad_(True)
    beta = (torch.randn(C, device=dev) * 0.3).requires_grad_(True)
    gy = torch.randn(N, C, T, device=dev)
    if prec == "bf16":
        gy = gy.bfloat16().float()
    ref = F.mish(F.group_norm(x, 8, gamma, beta, eps=1e-5))
    gref = torch.autograd.grad(ref, [x, gamma, beta], gy)
    x2, g2, b2 = [t.detach().clone().requires_grad_(True) for t in (x, gamma, beta)]
    out = ops.GNMish.apply(M.to_cl(x2), g2, b2, None, None, N, T, C, 8)
    o = M.from_cl(out, N, T, C)
    g = torch.autograd.grad(o, [x2, g2, b2], gy)
    e = (g[0] - gref[0]).abs().amax(dim=(1, 2))
    print((N, T, C), "fwd err", float((o - ref).abs().max()), "dx err per n", [round(float(v), 4) for v in e][:12],
          "dgamma err", float((g[1] - gref[1]).abs().max()), "dbeta err", float((g[2] - gref[2]).abs().max()))
