"""The N>1 path on CPU: world_size-2 gloo run of the data-parallel optimizer (flat buffers, reverse-order
buckets, hooks, accumulation without sync) on a small nn.Module.  Both ranks must end with identical parameters,
equal to a single-process run on the concatenated batch."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.body = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16),
                                        torch.nn.Tanh(), torch.nn.Linear(16, 3))
        self.unused = torch.nn.Linear(4, 4)   # never receives a gradient, like self_attn.3.* of the denoiser

    def forward(self, x):
        return self.body(x)


def _model():
    torch.manual_seed(0)
    return _Net()


def _data(rank_slice=None):
    g = torch.Generator().manual_seed(1)
    x, y = torch.randn(4, 8, 6, generator=g), torch.randn(4, 8, 3, generator=g)   # [step*2 micro][rows]
    return x, y


def _run(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from madiff_amd.trainer import FlatAdamEMA
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import copy
    m = _model()
    ema = copy.deepcopy(m)
    opt = FlatAdamEMA(m, ema, lr=1e-2, bucket_mb=0.0005, use_hip_kernel=False)
    opt.broadcast_parameters(0)
    x, y = _data()
    half = slice(0, 4) if rank == 0 else slice(4, 8)
    for step in range(2):
        for micro in range(2):
            opt.set_sync(micro == 1)
            loss = ((m(x[step * 2 + micro, half]) - y[step * 2 + micro, half]) ** 2).mean()
            (loss / 2).backward()
        opt.step(do_ema=1)
        opt.zero_grad()
    assert len(opt.buckets) > 2
    assert m.unused.weight.grad is None
    out[rank] = {k: v.detach().clone() for k, v in list(m.state_dict().items()) + [("ema." + k, v) for k, v in ema.state_dict().items()]}
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process():
    import copy
    from madiff_amd.trainer import FlatAdamEMA
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    port = 29500 + os.getpid() % 500
    mp.spawn(_run, args=(2, port, out), nprocs=2, join=True)
    a, b = out[0], out[1]
    for k in a:
        assert torch.equal(a[k], b[k]), k
    # single process on the full rows: mean over 8 rows == average of the two ranks' means over 4 rows
    m = _model()
    ema = copy.deepcopy(m)
    opt = FlatAdamEMA(m, ema, lr=1e-2, use_hip_kernel=False)
    x, y = _data()
    for step in range(2):
        for micro in range(2):
            opt.set_sync(micro == 1)
            loss = ((m(x[step * 2 + micro]) - y[step * 2 + micro]) ** 2).mean()
            (loss / 2).backward()
        opt.step(do_ema=1)
        opt.zero_grad()
    ref = {k: v for k, v in list(m.state_dict().items()) + [("ema." + k, v) for k, v in ema.state_dict().items()]}
    for k in ref:
        assert torch.allclose(a[k], ref[k], rtol=1e-5, atol=1e-6), k
    # and against torch.optim.Adam
    m2 = _model()
    o2 = torch.optim.Adam(m2.parameters(), lr=1e-2)
    for step in range(2):
        for micro in range(2):
            (((m2(x[step * 2 + micro]) - y[step * 2 + micro]) ** 2).mean() / 2).backward()
        o2.step()
        o2.zero_grad()
    for k, v in m2.state_dict().items():
        assert torch.allclose(ref[k], v, rtol=1e-5, atol=1e-6), k
