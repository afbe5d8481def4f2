"""Which GEMM-layout gradient slots of the flat optimizer get non-zero PAD columns after one backward?  (They must stay
exactly zero: pad weights would otherwise start to move and multiply whatever the pad columns of an activation hold.)"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import diffuser.models as DM
from madiff_amd import ops
from madiff_amd.host import batch_to_device
from madiff_amd.trainer import FlatAdamEMA, train_step
from oracle import madiff_oracle as O

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
ops.set_precision(prec)
dev = torch.device("cuda:0")
A, T, FD, ACT, B = 3, 12, 16, 5, 8
torch.manual_seed(0)
net = DM.SharedConvAttentionDeconv(horizon=T, transition_dim=FD, dim=32, dim_mults=(1, 4, 8), n_agents=A,
                                   returns_condition=True, condition_dropout=0.25)
dif = DM.GaussianDiffusion(net, n_agents=A, horizon=T, history_horizon=0, observation_dim=FD, action_dim=ACT,
                           n_timesteps=50, returns_condition=True, hidden_dim=64).to(dev)
for a in net.self_attn:
    a.gamma.data.fill_(0.3)
batch = batch_to_device(O.synthetic_batch(B, T, A, FD, ACT, 0, seed=1), dev)
opt = FlatAdamEMA(dif, copy.deepcopy(dif), lr=1e-3, ema_decay=0.9, bucket_mb=0.25)
train_step(dif, opt, [batch], do_ema=2)
opt.overlap = False
# every uninitialised buffer now holds NaN: a kernel that leaves pad columns (or anything else) unwritten shows up
_empty, _empty_like = torch.empty, torch.empty_like
def empty(*a, **k):
    t = _empty(*a, **k)
    return t.fill_(float("nan")) if t.is_floating_point() and t.is_cuda else t
def empty_like(x, **k):
    t = _empty_like(x, **k)
    return t.fill_(float("nan")) if t.is_floating_point() and t.is_cuda else t
torch.empty, torch.empty_like = empty, empty_like
import madiff_amd.modules as MM
seen = {}
_cg = ops._conv_forward
def conv_forward(x, x2, rowbias, spec, t_in, act, perm, weights, biases):
    out, g = _cg(x, x2, rowbias, spec, t_in, act, perm, weights, biases)
    bad = int(torch.isnan(out.float()).sum())
    if bad:
        print("conv_forward out NaNs", bad, tuple(out.shape), "osum", g.osum, "cin", g.cin, "taps", spec.taps, "perm", perm)
    for nm, t in (("x", x), ("x2", x2)):
        if t is not None and int(torch.isnan(t.float()).sum()):
            print("conv_forward INPUT", nm, "NaNs", int(torch.isnan(t.float()).sum()), tuple(t.shape), "osum", g.osum, "taps", spec.taps)
    return out, g
ops._conv_forward = conv_forward
opt.set_sync(True)
loss, _ = dif.loss(**batch)
print("loss", float(loss))
loss.backward()
ops.join_aux_streams()
torch.cuda.synchronize()
bad = 0
for n, p, o, geom in zip(opt.names, opt.params, opt.offsets, opt.geoms):
    if geom is None:
        continue
    kind, taps, Od, I, ipad = geom
    for what, flat in (("grad", opt.flat_g), ("param", opt.flat_p)):
        v = flat[o:o + Od * taps * ipad].view(Od, taps, ipad)
        if ipad > I:
            m = float(v[:, :, I:].abs().max())
            if m != 0.0:
                bad += 1
                print(f"{what:5s} {n:55s} {kind:6s} I={I:4d} ipad={ipad:4d} max|pad| = {m:.3e}  (max|real| = {float(v[:, :, :I].abs().max()):.3e})")
print("non-zero pad slots:", bad)
