import copy, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffuser.models as DM
from madiff_amd import ops
from madiff_amd.host import batch_to_device
from madiff_amd.trainer import FlatAdamEMA, GraphedTrainStep, train_step
from oracle import madiff_oracle as O
dev = torch.device("cuda:0")
ops.set_precision(sys.argv[1] if len(sys.argv) > 1 else "fp32")
torch.manual_seed(0)
A, T, Fd, B = 3, 24, 16, 4
net = DM.SharedConvAttentionDeconv(horizon=T, transition_dim=Fd, dim=32, dim_mults=(1, 4, 8), n_agents=A, returns_condition=True, condition_dropout=0.25)
dif = DM.GaussianDiffusion(net, n_agents=A, horizon=T, history_horizon=0, observation_dim=Fd, action_dim=5, n_timesteps=50, returns_condition=True, hidden_dim=64).to(dev)
for a in net.self_attn: a.gamma.data.fill_(0.3)
batch = batch_to_device(O.synthetic_batch(B, T, A, Fd, 5, 0, seed=1), dev)
g = torch.Generator().manual_seed(5)
draws = dict(t=torch.randint(0, 50, (B,), generator=g).to(dev), noise=torch.randn(B, T, A, Fd, generator=g).to(dev),
             dropout_mask=(torch.rand(B, A, 1, generator=g) < 0.75).float().to(dev))
for mode in ("eager", "graph"):
    m = copy.deepcopy(dif); ema = copy.deepcopy(m)
    opt = FlatAdamEMA(m, ema, lr=1e-3, ema_decay=0.9)
    l, _ = train_step(m, opt, [{**batch, **draws}], do_ema=2)
    ls = [float(l)]; norms = []
    if mode == "eager":
        for i in range(4):
            l, _ = train_step(m, opt, [{**batch, **draws}], do_ema=i % 2); ls.append(float(l)); norms.append(float(opt.flat_p.norm()))
    else:
        gs = GraphedTrainStep(m, opt, batch, draws=draws); ls.append(float("nan")); norms.append(float(opt.flat_p.norm()))
        for i in range(1, 4):
            l, _ = gs.step(batch, do_ema=i % 2); ls.append(float(l)); norms.append(float(opt.flat_p.norm()))
            print("   state", gs.state.tolist(), "gnorm", float(opt.flat_g.norm()))
    print(mode, ls, norms)
    ops.SINK = None
