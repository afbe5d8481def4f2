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


def _run(rank, world, port, out, bucket_dtype=None):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from madiff_amd.trainer import FlatAdamEMA
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import copy
    m = _model()
    ema = copy.deepcopy(m)
    opt = FlatAdamEMA(m, ema, lr=1e-2, bucket_mb=0.0005, use_hip_kernel=False, bucket_dtype=bucket_dtype)
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
    if bucket_dtype == "bf16":
        assert opt._stage16 is not None and opt._stage16.dtype == torch.bfloat16 and opt.flat_g.dtype == torch.float32
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


def test_two_rank_gloo_bf16_gradient_transport():
    """bucket_dtype="bf16": buckets travel as bf16 and come back into the fp32 master gradient.  Ranks stay bit-identical
    (they apply the same reduced values); against the fp32 transport the FIRST-step gradient differs by bf16 rounding
    (2^-9 relative per addend), which Adam turns into parameter differences far below one step (lr = 1e-2)."""
    mgr = mp.get_context("spawn").Manager()
    out16, out32 = mgr.dict(), mgr.dict()
    port = 29000 + os.getpid() % 400
    mp.spawn(_run, args=(2, port, out16, "bf16"), nprocs=2, join=True)
    mp.spawn(_run, args=(2, port + 1, out32, None), nprocs=2, join=True)
    a, b, ref = out16[0], out16[1], out32[0]
    worst = 0.0
    for k in a:
        assert torch.equal(a[k], b[k]), k
        worst = max(worst, float((a[k] - ref[k]).abs().max()))
    assert 0.0 < worst < 2e-3, worst          # not identical to fp32 transport, and well inside 2 steps x lr


def test_bucket_dtype_is_validated():
    from madiff_amd.trainer import FlatAdamEMA
    with pytest.raises(ValueError):
        FlatAdamEMA(_model(), None, bucket_dtype="fp8")


def test_host_arithmetic_optimizer_must_be_asked_for():
    """The fused Adam/EMA step is a HIP kernel.  The torch host-arithmetic branch exists for the gloo tests above only: a
    module that is not on a GPU raises unless use_hip_kernel=False is passed explicitly (no silent second path)."""
    from madiff_amd.trainer import FlatAdamEMA
    m = _model()
    x, y = _data()
    ((m(x[0]) - y[0]) ** 2).mean().backward()
    opt = FlatAdamEMA(m, None, lr=1e-2)
    with pytest.raises(RuntimeError, match="use_hip_kernel=False"):
        opt.step()
    with pytest.raises(RuntimeError, match="needs the module on a GPU"):
        FlatAdamEMA(m, None, lr=1e-2, use_hip_kernel=True).step()
    FlatAdamEMA(m, None, lr=1e-2, use_hip_kernel=False).step()          # the explicit flag still works


def test_wgrad_branch_belongs_to_the_parameter():
    """Parameter-gradient launches are read-modify-writes of the flat gradient: both launches for one parameter (the
    micro-batches of an unfolded accumulation step) must go to the same auxiliary stream, whatever was issued between."""
    from madiff_amd import ops
    sink = ops.GradSink()
    ws = [torch.nn.Parameter(torch.zeros(2, 2)) for _ in range(2 * ops.WGRAD_STREAMS + 1)]
    first = [ops.wgrad_branch_name(sink, [w]) for w in ws]
    assert len(set(first)) == max(1, ops.WGRAD_STREAMS)                  # taken in turn over all branches
    again = [ops.wgrad_branch_name(sink, [w]) for w in reversed(ws)][::-1]
    assert again == first                                                # the same branch on every later launch
    assert ops.wgrad_branch_name(sink, [ws[3], ws[0]]) == first[3]       # a fused group follows its first member
