"""Sampling shards episodes over ranks with no collective inside the K-step loop and ONE final gather (SURVEY.md 8e
row 2; reference call site diffuser/utils/evaluator.py:101-106,127-134 samples all episodes in one process).

CPU part (world_size 2, gloo): madiff_amd.sharding.sample_sharded around the ORACLE's K-step guided sampler (the HIP
sampler needs a GPU) on a ragged episode count; the gathered plan must equal the unsharded run, and the loop must
contain no collective.  GPU part (two ranks on cuda:0, gloo gather on host copies as in test_gpu_dp2.py): the same
with the HIP sampler replaying its captured graph, against the oracle on all episodes."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
A, T, FD, ACT, K, B = 3, 8, 6, 2, 4, 5          # B = 5 over 2 ranks: ragged shards (3 + 2)


def _oracle_parts():
    sys.path.insert(0, ROOT)
    from oracle import madiff_oracle as O
    dcfg = O.DenoiserCfg(horizon=T, transition_dim=FD, dim=16, dim_mults=(1, 4, 8), n_agents=A, returns_condition=True,
                         condition_dropout=0.25)
    c = O.DiffusionCfg(A, T, 0, FD, ACT, K, hidden_dim=32, returns_condition=True)
    shapes = {**{"model." + k: v for k, v in O.denoiser_param_shapes(dcfg).items()}, **O.inv_param_shapes(c)}
    params = O.closed_form_state(shapes)
    for i in range(4):
        params[f"model.self_attn.{i}.gamma"] = torch.full((1,), 0.4)
    return O, dcfg, c, params


def _inputs():
    g = torch.Generator().manual_seed(11)
    cond_x = torch.rand(B, T, A, FD, generator=g) * 2 - 1
    masks = torch.zeros(B, T, A, FD, dtype=torch.bool)
    masks[:, :1] = True
    returns = torch.rand(B, 1, A, generator=g)
    x_init = 0.5 * torch.randn(B, T, A, FD, generator=g)
    step_noise = torch.randn(K, B, T, A, FD, generator=g)
    return {"x": cond_x, "masks": masks}, returns, x_init, step_noise


class _OracleSampler:
    """The oracle's guided K-step loop behind GaussianDiffusion.conditional_sample's signature; counts calls so the
    test can show the loop ran once per rank on the rank's own shard."""

    def __init__(self):
        self.O, self.dcfg, self.c, self.params = _oracle_parts()
        self.sched = self.O.DDPMSchedule(K)
        self.batches = []

    @torch.no_grad()
    def conditional_sample(self, cond, returns=None, env_ts=None, verbose=True, x_init=None, step_noise=None):
        self.batches.append(cond["x"].shape[0])
        return self.O.conditional_sample(self.params, self.dcfg, self.c, self.sched, cond, returns, x_init,
                                         list(step_noise))


def _run_cpu(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from madiff_amd.sharding import sample_sharded, shard_range
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    s = _OracleSampler()
    cond, returns, x_init, step_noise = _inputs()
    # count collectives issued while the sampler runs: there must be none inside the loop
    calls = {"n": 0}
    real = {n: getattr(dist, n) for n in ("all_reduce", "all_gather", "gather", "broadcast", "barrier")}

    def counted(name):
        def f(*a, **k):
            calls["n"] += 1
            return real[name](*a, **k)
        return f

    for n in real:
        setattr(dist, n, counted(n))
    x = sample_sharded(s, cond, returns, x_init=x_init, step_noise=step_noise, dst=0)
    xa = sample_sharded(s, cond, returns, x_init=x_init, step_noise=step_noise, dst=None)
    for n in real:
        setattr(dist, n, real[n])
    lo, hi = shard_range(B, rank, world)
    assert s.batches == [hi - lo, hi - lo], s.batches
    assert calls["n"] == 2, calls                       # one gather, one all_gather: nothing else
    assert (x is None) == (rank != 0)
    out[rank] = (None if x is None else x.clone(), xa.clone())
    dist.destroy_process_group()


def test_shard_range_partitions():
    sys.path.insert(0, ROOT)
    from madiff_amd.sharding import shard_range
    for n in (0, 1, 5, 8, 1024, 8192, 8191):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [h - l for l, h in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(8192, 3, 8) == (3072, 4096)


def test_two_rank_sharded_sampling_matches_unsharded():
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    port = 29800 + os.getpid() % 150
    mp.spawn(_run_cpu, args=(2, port, out), nprocs=2, join=True)
    x0, xa0 = out[0]
    _, xa1 = out[1]
    assert torch.equal(xa0, xa1) and torch.equal(x0, xa0)
    s = _OracleSampler()
    cond, returns, x_init, step_noise = _inputs()
    ref = s.conditional_sample(cond, returns, x_init=x_init, step_noise=step_noise)
    assert x0.shape == ref.shape == (B, T, A, FD)
    # same arithmetic per episode; CPU GEMM blocking may differ with the batch size, hence not bitwise
    assert (x0 - ref).abs().max().item() <= 2e-4 * ref.abs().max().item()
    assert torch.equal(x0[:, :1], cond["x"][:, :1])     # inpainting survives the gather


def test_single_process_is_the_plain_call():
    sys.path.insert(0, ROOT)
    from madiff_amd.sharding import sample_sharded
    s = _OracleSampler()
    cond, returns, x_init, step_noise = _inputs()
    x = sample_sharded(s, cond, returns, x_init=x_init, step_noise=step_noise)
    assert s.batches == [B] and x.shape == (B, T, A, FD)


# ------------------------------------------------------------------------------------------------ GPU
def _run_gpu(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    real_gather = dist.gather

    def gather(t, gather_list=None, dst=0, group=None):           # gloo's CUDA path hangs on this image: stage on host
        torch.cuda.synchronize()
        h = t.detach().cpu()
        hl = [torch.empty_like(h) for _ in gather_list] if gather_list is not None else None
        real_gather(h, hl, dst=dst, group=group)
        if gather_list is not None:
            for d, s_ in zip(gather_list, hl):
                d.copy_(s_)

    dist.gather = gather
    import diffuser.models as DM
    from madiff_amd import ops
    from madiff_amd.sharding import sample_sharded
    ops.set_precision("fp32")
    dev = torch.device("cuda:0")
    O, dcfg, c, params = _oracle_parts()
    net = DM.SharedConvAttentionDeconv(horizon=T, transition_dim=FD, dim=16, dim_mults=(1, 4, 8), n_agents=A,
                                       returns_condition=True, condition_dropout=0.25)
    dif = DM.GaussianDiffusion(net, n_agents=A, horizon=T, history_horizon=0, observation_dim=FD, action_dim=ACT,
                               n_timesteps=K, returns_condition=True, hidden_dim=32)
    missing = dif.load_state_dict(params, strict=False)
    assert set(missing.missing_keys) <= {"loss_fn.weights"}, missing
    dif = dif.to(dev)
    cond, returns, x_init, step_noise = _inputs()
    mv = lambda t: t.to(dev)                                     # noqa: E731
    x = sample_sharded(dif, {k: mv(v) for k, v in cond.items()}, mv(returns), x_init=mv(x_init),
                       step_noise=mv(step_noise), dst=0)
    out[rank] = None if x is None else x.cpu()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_sharded_sampling_vs_oracle():
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    port = 29860 + os.getpid() % 100
    mp.spawn(_run_gpu, args=(2, port, out), nprocs=2, join=True)
    assert out[1] is None
    s = _OracleSampler()
    cond, returns, x_init, step_noise = _inputs()
    ref = s.conditional_sample(cond, returns, x_init=x_init, step_noise=step_noise)
    err = (out[0] - ref).abs().max().item() / ref.abs().max().item()
    assert err <= 1e-3, err                                      # north-star bound for sampled trajectories, fp32 build
