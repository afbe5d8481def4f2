"""diffuser.utils.Trainer (madiff_amd/training.py; reference diffuser/utils/training.py:42-227) on the GPU against a
plain loop over the CPU oracle: gradient accumulation, Adam, the EMA schedule (copy before step_start_ema, lerp after,
every update_ema_every steps), checkpoint round trip.  Every random draw travels inside the dataset items, so the
eager and the graph-replayed Trainer and the oracle loop see identical inputs (fp32 build)."""
import copy

import pytest
import torch

from conftest import rel_err
from oracle import madiff_oracle as O

pytestmark = pytest.mark.gpu

A, T, FD, ACT, K, B, ACC, STEPS = 3, 12, 10, 5, 20, 4, 2, 7
ZERO_GRAD = ("key_layer.bias", "key_time_mlp.1.bias")     # softmax shift invariance: analytically zero gradient


class _Items(torch.utils.data.Dataset):
    def __init__(self, n):
        b = O.synthetic_batch(n, T, A, FD, ACT, 0, seed=11)
        g = torch.Generator().manual_seed(12)
        b["t"] = torch.randint(0, K, (n,), generator=g)
        b["noise"] = torch.randn(n, T, A, FD, generator=g)
        b["dropout_mask"] = (torch.rand(n, A, 1, generator=g) < 0.75).float()
        self.b, self.n = b, n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return {k: ({kk: vv[i] for kk, vv in v.items()} if isinstance(v, dict) else v[i]) for k, v in self.b.items()}


def _build(dev):
    import diffuser.models as DM
    net = DM.SharedConvAttentionDeconv(horizon=T, transition_dim=FD, dim=16, dim_mults=(1, 4, 8), n_agents=A,
                                       returns_condition=True, condition_dropout=0.25)
    dif = DM.GaussianDiffusion(net, n_agents=A, horizon=T, history_horizon=0, observation_dim=FD, action_dim=ACT,
                               n_timesteps=K, returns_condition=True, hidden_dim=32).to(dev)
    sd = {k: (O.closed_form_param(k, tuple(v.shape)) if "loss_fn" not in k else v) for k, v in dif.state_dict().items()}
    dif.load_state_dict(sd)
    return dif, {k: v.clone() for k, v in sd.items() if "loss_fn" not in k}


def _oracle_loop(params, data, lr, beta, start, every):
    dcfg = O.DenoiserCfg(horizon=T, transition_dim=FD, dim=16, dim_mults=(1, 4, 8), n_agents=A, returns_condition=True,
                         condition_dropout=0.25)
    c = O.DiffusionCfg(A, T, 0, FD, ACT, K, hidden_dim=32, returns_condition=True)
    sched = O.DDPMSchedule(K)
    params = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ema = {k: v.detach().clone() for k, v in params.items()}
    opt = torch.optim.Adam(list(params.values()), lr=lr)
    pos = 0
    for step in range(STEPS):
        for _ in range(ACC):
            idx = [(pos + j) % len(data) for j in range(B)]
            pos += B
            mb = torch.utils.data.default_collate([data[j] for j in idx])
            loss, _, _ = O.diffusion_loss(params, dcfg, c, sched, mb["x"], mb["cond"], mb["loss_masks"], mb["t"], mb["noise"],
                                          mb["returns"], mb["dropout_mask"])
            (loss / ACC).backward()
        opt.step()
        opt.zero_grad()
        if step % every == 0:
            with torch.no_grad():
                for k in ema:
                    ema[k] = params[k].detach().clone() if step < start else beta * ema[k] + (1 - beta) * params[k].detach()
    return {k: v.detach() for k, v in params.items()}, ema


@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "graph"])
def test_trainer_matches_oracle_loop(use_graph, tmp_path):
    from diffuser.utils import Trainer
    from madiff_amd import ops
    dev = torch.device("cuda:0")
    ops.set_precision("fp32")
    try:
        data = _Items(B * ACC * 4)              # 4 optimizer steps per epoch: the loop wraps around (cycle)
        dif, params = _build(dev)
        tr = Trainer(dif, data, None, ema_decay=0.9, train_batch_size=B, train_lr=1e-3, gradient_accumulate_every=ACC,
                     step_start_ema=3, update_ema_every=2, log_freq=3, sample_freq=0, save_freq=100, eval_freq=0,
                     bucket=str(tmp_path), train_device="cuda:0", use_graph=use_graph, shuffle=False, bucket_mb=0.25)
        assert [tr.ema_mode(s) for s in range(7)] == [2, 0, 2, 0, 1, 0, 1]
        tr.train(STEPS)
        torch.cuda.synchronize()
        assert tr.step == STEPS and tr.optimizer.step_count == STEPS
        assert (tr._graphed is not None) == use_graph
        ref_p, ref_e = _oracle_loop(params, data, 1e-3, 0.9, 3, 2)
        # Tolerance in units of one Adam step (= lr per element while the moments are young): an element whose
        # gradient is at fp32 noise level may move by a fraction of a step differently, a systematic error (a missed
        # step, a wrong bias correction, a wrong EMA mode) moves EVERY element by about a whole step.
        lr = 1e-3
        for got, ref, what in ((tr.model.state_dict(), ref_p, "model"), (tr.ema_model.state_dict(), ref_e, "ema")):
            worst = worst_mean = 0.0
            for k, r in ref.items():
                if k.startswith("model.self_attn.3.") or k.endswith(ZERO_GRAD):
                    continue
                d = (got[k].detach().cpu() - r).abs()
                worst, worst_mean = max(worst, float(d.max())), max(worst_mean, float(d.mean()))
                assert float(d.max()) < 0.3 * lr and float(d.mean()) < 0.02 * lr, (what, k, float(d.max()), float(d.mean()))
            print(f"{what}: worst element error {worst / lr:.3f} lr, worst per-tensor mean error {worst_mean / lr:.4f} lr")
        # the unused attention layer never moves (torch.optim.Adam skips grad None)
        for k, v in params.items():
            if k.startswith("model.self_attn.3."):
                assert torch.equal(tr.model.state_dict()[k].cpu(), v)
        # checkpoint round trip (training.py:197-227)
        tr.save()
        before = {k: v.detach().clone() for k, v in tr.model.state_dict().items()}
        ema_before = {k: v.detach().clone() for k, v in tr.ema_model.state_dict().items()}
        with torch.no_grad():
            for p in tr.model.parameters():
                p.add_(1.0)
        tr.step = 0
        tr.load()
        assert tr.step == STEPS
        for k, v in tr.model.state_dict().items():
            assert torch.equal(v, before[k]), k
        for k, v in tr.ema_model.state_dict().items():
            assert torch.equal(v, ema_before[k]), k
        tr.train(1)                              # and training goes on from the restored weights
        torch.cuda.synchronize()
        assert tr.step == STEPS + 1
    finally:
        ops.SINK = None
        ops.set_precision("bf16")
