import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diffuser.models as DM
from madiff_amd import ops, diffusion as D
from madiff_amd.host import batch_to_device
from madiff_amd.trainer import FlatAdamEMA, GraphedTrainStep, train_step
from oracle import madiff_oracle as O
dev = torch.device("cuda:0")
ops.set_precision("fp32")
torch.manual_seed(0)
A, T, Fd, B = 3, 24, 16, 4
net = DM.SharedConvAttentionDeconv(horizon=T, transition_dim=Fd, dim=32, dim_mults=(1, 4, 8), n_agents=A,
                                   returns_condition=True, condition_dropout=0.25)
dif = DM.GaussianDiffusion(net, n_agents=A, horizon=T, history_horizon=0, observation_dim=Fd, action_dim=5,
                           n_timesteps=50, returns_condition=True, hidden_dim=64).to(dev)
for a in net.self_attn:
    a.gamma.data.fill_(0.3)
batch = batch_to_device(O.synthetic_batch(B, T, A, Fd, 5, 0, seed=1), dev)
g = torch.Generator().manual_seed(5)
draws = dict(t=torch.randint(0, 50, (B,), generator=g).to(dev), noise=torch.randn(B, T, A, Fd, generator=g).to(dev),
             dropout_mask=(torch.rand(B, A, 1, generator=g) < 0.75).float().to(dev))
ref = None
for side in (False, True):
    for mode in ("eager", "graph"):
        D.INV_SIDE_STREAM = side
        m = copy.deepcopy(dif)
        ema = copy.deepcopy(m)
        opt = FlatAdamEMA(m, ema, lr=1e-3, ema_decay=0.9)
        losses = [float(train_step(m, opt, [{**batch, **draws}], do_ema=2)[0])]
        if mode == "eager":
            for i in range(4):
                losses.append(float(train_step(m, opt, [{**batch, **draws}], do_ema=i % 2)[0]))
        else:
            gs = GraphedTrainStep(m, opt, batch, draws=draws)
            for i in range(1, 4):
                losses.append(float(gs.step(batch, do_ema=i % 2)[0]))
        torch.cuda.synchronize()
        print(side, mode, ["%.10f" % l for l in losses], float(opt.flat_p.double().abs().sum()))
        fp = opt.flat_p.clone()
        if ref is None:
            ref = fp
        else:
            worst = []
            for n, o, p_ in zip(opt.names, opt.offsets, opt.params):
                k = p_.numel() if opt.geoms[opt.names.index(n)] is None else None
                hi = o + (k if k is not None else opt.geoms[opt.names.index(n)][2] * opt.geoms[opt.names.index(n)][1] * opt.geoms[opt.names.index(n)][4])
                d = (fp[o:hi] - ref[o:hi]).abs().max().item() / (ref[o:hi].abs().max().item() + 1e-12)
                worst.append((d, n))
            worst.sort(reverse=True)
            print("   worst params:", [(f"{d:.1e}", n) for d, n in worst[:6]])
        ops.SINK = None
