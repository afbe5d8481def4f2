"""debug: bf16 vs fp32 build sampler outputs at full width, per step count"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diffuser.models as DM
from madiff_amd import ops
from oracle import madiff_oracle as O
dev = torch.device("cuda:0")
A, Tt, Fd, hh, act, nact, B = 4, 24, 16, 0, 5, 0, 64
for K in (2, 4, 12):
    torch.manual_seed(0)
    net = DM.SharedConvAttentionDeconv(horizon=Tt, transition_dim=Fd, dim=128, history_horizon=hh, dim_mults=(1, 4, 8),
                                       n_agents=A, returns_condition=True, condition_dropout=0.25)
    dif = DM.GaussianDiffusion(net, n_agents=A, horizon=Tt - hh, history_horizon=hh, observation_dim=Fd, action_dim=act,
                               n_timesteps=K, returns_condition=True).to(dev)
    for a in net.self_attn:
        a.gamma.data.fill_(0.5)
    b = O.synthetic_batch(B, Tt, A, Fd, act, hh, seed=0)
    cond = {k: v.to(dev) for k, v in b["cond"].items()}
    ret = b["returns"].to(dev)
    g = torch.Generator().manual_seed(7)
    x_init = (0.5 * torch.randn(B, Tt, A, Fd, generator=g)).to(dev)
    z = torch.randn(max(K - 1, 1), B, Tt, A, Fd, generator=g).to(dev)
    xs = {}
    for prec in ("bf16", "fp32"):
        ops.set_precision(prec)
        for mode in (False, None):
            xs[(prec, mode)] = dif.conditional_sample(cond, returns=ret, verbose=False, x_init=x_init, step_noise=z, use_graph=mode)
    ops.set_precision("bf16")
    d = (xs[("bf16", None)] - xs[("fp32", None)]).abs()
    print(f"K={K}: bf16 vs fp32: max {float(d.max()):.4f} mean {float(d.mean()):.5f} p99.9 {float(d.flatten().kthvalue(int(d.numel()*0.999)).values):.4f}; "
          f"graph vs eager bf16 {float((xs[('bf16', None)] - xs[('bf16', False)]).abs().max()):.2e} fp32 {float((xs[('fp32', None)] - xs[('fp32', False)]).abs().max()):.2e}; "
          f"frac clipped {float((xs[('fp32', None)].abs() > 0.999).float().mean()):.3f}", flush=True)
    # eps of a single model call
    ts = torch.full((B,), K - 1, device=dev, dtype=torch.long)
    e = {}
    for prec in ("bf16", "fp32"):
        ops.set_precision(prec)
        with torch.no_grad():
            e[prec] = dif.get_model_output(x_init.clone(), ts, ret)
    ops.set_precision("bf16")
    print(f"      one guided eps call: rel max err {float((e['bf16'] - e['fp32']).abs().max() / e['fp32'].abs().max()):.4f}", flush=True)
