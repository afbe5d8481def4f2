"""The data-parallel train step on the GPU with TWO ranks, both on cuda:0: eager step with bucket hooks + GradSink
notifications, then the graph-replayed step (graph-external bucket events, all-reduce behind them, eager Adam).  RCCL
refuses two ranks on one device and gloo's own CUDA path hangs on this image, so the collectives are routed through gloo
on host copies (stream-ordered: the wrapper drains the stream it is called on first, i.e. it really waits for the
bucket's event).  Both ranks must end with identical parameters, close to a single-process run on the concatenated
batch (fp32 build)."""
import copy
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
A, T, FD, ACT, B = 3, 12, 16, 5, 8


def _setup():
    sys.path.insert(0, ROOT)
    import diffuser.models as DM
    from madiff_amd import ops
    from madiff_amd.host import batch_to_device
    from oracle import madiff_oracle as O
    ops.set_precision("fp32")
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = DM.SharedConvAttentionDeconv(horizon=T, transition_dim=FD, dim=32, dim_mults=(1, 4, 8), n_agents=A,
                                       returns_condition=True, condition_dropout=0.25)
    dif = DM.GaussianDiffusion(net, n_agents=A, horizon=T, history_horizon=0, observation_dim=FD, action_dim=ACT,
                               n_timesteps=50, returns_condition=True, hidden_dim=64).to(dev)
    for a in net.self_attn:
        a.gamma.data.fill_(0.3)
    batch = batch_to_device(O.synthetic_batch(B, T, A, FD, ACT, 0, seed=1), dev)
    g = torch.Generator().manual_seed(5)
    draws = dict(t=torch.randint(0, 50, (B,), generator=g).to(dev), noise=torch.randn(B, T, A, FD, generator=g).to(dev),
                 dropout_mask=(torch.rand(B, A, 1, generator=g) < 0.75).float().to(dev))
    return dif, batch, draws


def _slice(v, sl):
    return {k: _slice(u, sl) for k, u in v.items()} if isinstance(v, dict) else v[sl].contiguous()


def _steps(dif, batch, draws):
    from madiff_amd import ops
    from madiff_amd.trainer import FlatAdamEMA, GraphedTrainStep, train_step
    ema = copy.deepcopy(dif)
    opt = FlatAdamEMA(dif, ema, lr=1e-3, ema_decay=0.9, bucket_mb=0.25)
    opt.broadcast_parameters(0)
    train_step(dif, opt, [{**batch, **draws}], do_ema=2)          # materialises the flat buffers (one flat all-reduce)
    train_step(dif, opt, [{**batch, **draws}], do_ema=1)          # eager: bucket hooks / sink notifications
    gs = GraphedTrainStep(dif, opt, batch, draws=draws)           # one more eager step, then capture
    for i in range(2):
        gs.step(batch, do_ema=i % 2)
    torch.cuda.synchronize()
    assert opt.step_count == 5
    ops.SINK = None
    return opt, gs


def _host_staged_collectives(dist):
    real_ar, real_bc = dist.all_reduce, dist.broadcast

    class _Done:
        def wait(self):
            return True

    def all_reduce(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
        if not t.is_cuda:
            return real_ar(t, op=op, group=group, async_op=async_op)
        torch.cuda.current_stream().synchronize()        # everything the caller ordered before this collective
        h = t.detach().cpu()
        real_ar(h, op=op, group=group)
        t.copy_(h)
        return _Done() if async_op else None

    def broadcast(t, src=0, group=None, async_op=False):
        if not t.is_cuda:
            return real_bc(t, src=src, group=group, async_op=async_op)
        torch.cuda.current_stream().synchronize()
        h = t.detach().cpu()
        real_bc(h, src=src, group=group)
        t.copy_(h)
        return _Done() if async_op else None

    dist.all_reduce, dist.broadcast = all_reduce, broadcast


def _run(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _host_staged_collectives(dist)
    dif, batch, draws = _setup()
    half = slice(0, B // 2) if rank == 0 else slice(B // 2, B)
    opt, gs = _steps(dif, _slice(batch, half), _slice(draws, half))
    assert opt.dp and gs.dp and len(opt.buckets) >= 3
    out[rank] = (opt.flat_p.cpu(), opt.flat_ema.cpu())
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_match_single_process():
    from conftest import rel_err
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    port = 29600 + os.getpid() % 300
    mp.spawn(_run, args=(2, port, out), nprocs=2, join=True)
    (p0, e0), (p1, e1) = out[0], out[1]
    assert torch.equal(p0, p1) and torch.equal(e0, e1), "ranks diverged"
    from madiff_amd import ops
    try:
        dif, batch, draws = _setup()
        opt, gs = _steps(dif, batch, draws)
        assert not opt.dp
        assert rel_err(p0, opt.flat_p.cpu()) < 1e-4
        assert rel_err(e0, opt.flat_ema.cpu()) < 1e-4
    finally:
        ops.SINK = None
        ops.set_precision("bf16")


def _run_trainer(rank, world, port, out, ckpt_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _host_staged_collectives(dist)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_trainer as TT
    from diffuser.utils import Trainer
    from madiff_amd import ops
    ops.set_precision("fp32")
    dif, _ = TT._build(torch.device("cuda:0"))
    data = TT._Items(TT.B * TT.ACC * 2 * 4)
    tr = Trainer(dif, data, None, ema_decay=0.9, train_batch_size=TT.B, train_lr=1e-3, gradient_accumulate_every=TT.ACC,
                 step_start_ema=3, update_ema_every=2, log_freq=10 ** 9, sample_freq=0, save_freq=4, eval_freq=0,
                 bucket=ckpt_dir, train_device="cuda:0", use_graph=True, shuffle=True, seed=3, bucket_mb=0.25)
    assert tr.world == 2 and tr.optimizer.dp
    tr.train(6)
    torch.cuda.synchronize()
    assert tr.step == 6 and tr._graphed is not None and tr._graphed.dp
    out[rank] = (tr.optimizer.flat_p.cpu(), tr.optimizer.flat_ema.cpu(), sorted(os.listdir(os.path.join(ckpt_dir, "checkpoint"))))
    ops.SINK = None
    dist.destroy_process_group()


@pytest.mark.gpu
def test_trainer_two_ranks_on_one_gpu(tmp_path):
    """diffuser.utils.Trainer under data parallelism (two ranks on cuda:0, collectives staged through gloo): every rank
    draws its own DistributedSampler shard, accumulates two micro-batches inside one captured graph, all-reduces per
    bucket behind the graph's event nodes; the ranks must stay bit-identical and only rank 0 writes checkpoints."""
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    port = 29650 + os.getpid() % 300
    mp.spawn(_run_trainer, args=(2, port, out, str(tmp_path)), nprocs=2, join=True)
    (p0, e0, f0), (p1, e1, f1) = out[0], out[1]
    assert torch.equal(p0, p1) and torch.equal(e0, e1), "ranks diverged"
    assert bool(torch.isfinite(p0).all()) and f0 == ["state.pt"]
