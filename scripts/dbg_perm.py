"""debug: batch-permutation equivariance / run-to-run determinism of the full-size denoiser forward"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import diffuser.models as DM
from madiff_amd import ops
from conftest import rel_err
dev = torch.device("cuda:0")
torch.manual_seed(0)
A, Tt, Fd = 4, 24, 16
net = DM.SharedConvAttentionDeconv(horizon=Tt, transition_dim=Fd, dim=128, dim_mults=(1, 4, 8), n_agents=A,
                                   returns_condition=True, condition_dropout=0.25).to(dev)
for a in net.self_attn:
    a.gamma.data.fill_(0.5)
x = torch.rand(256, Tt, A, Fd, device=dev) * 2 - 1
ret = torch.rand(256, 1, A, device=dev)
t = torch.randint(0, 200, (256,), device=dev)
perm = torch.randperm(256, device=dev)
for fuse in (True, False):
    ops.FUSE_CONV_GN = fuse
    for grad in (False, True):
        with torch.set_grad_enabled(grad):
            y = net(x, t, returns=ret, use_dropout=False).detach()
            y2 = net(x, t, returns=ret, use_dropout=False).detach()
            yp = net(x[perm], t[perm], returns=ret[perm], use_dropout=False).detach()
        print(f"fuse={fuse} grad={grad}: rerun {rel_err(y2.cpu(), y.cpu()):.3e}  perm {rel_err(yp.cpu(), y[perm].cpu()):.3e}", flush=True)
