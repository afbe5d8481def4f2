"""madiff_loss_fwd_static (the entry point that allocates nothing: its scratch is a static device array inside the
library) against madiff_loss_fwd (caller-provided workspace) through the C ABI: same kernels, so the outputs must be
bit-identical -- on two calls in a row, the second with a smaller input (the array keeps the first call's contents), for
the continuous and the discrete head, and the capacity limit must be refused with -2."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(B, T, A, F, act, nout, discrete, seed, dev):
    g = torch.Generator().manual_seed(seed)
    eps = torch.randn(B, T, A, F, generator=g)
    noise = torch.randn(B, T, A, F, generator=g)
    w = torch.rand(T, A, F, generator=g) + 0.5
    masks = (torch.rand(B, T, A, generator=g) < 0.8).double()
    masks[:, -1] = 1.0                                     # every sample keeps a non-empty mask
    ld = 32
    pred = torch.zeros(B * (T - 1) * A, ld)
    pred[:, :nout] = torch.randn(B * (T - 1) * A, nout, generator=g)
    x = torch.rand(B, T, A, act + F, generator=g) * 2 - 1
    legal = None
    if discrete:
        a_id = torch.randint(0, nout, (B, T, A), generator=g)
        x[..., 0] = a_id.float()
        legal = (torch.rand(B, T, A, nout, generator=g) < 0.7).float()
        legal.scatter_(-1, a_id[..., None], 1.0)
    mv = lambda t: None if t is None else t.to(dev).contiguous()          # noqa: E731
    return [mv(t) for t in (eps, noise, w, masks, pred, x, legal)], (B, T, A, F, act, nout, int(discrete), A * ld, ld)


def _run(lib, tensors, geom, static):
    eps, noise, w, masks, pred, x, legal = tensors
    B, T, A, F, act, nout, discrete, rs, as_ = geom
    dev = eps.device
    out = torch.full((6,), float("nan"), dtype=torch.float64, device=dev)
    out_f = torch.full((6,), float("nan"), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: None if t is None else t.data_ptr()                      # noqa: E731
    common = (p(eps), p(noise), p(w), p(masks), 1, p(pred), rs, as_, p(x), p(legal), B, T, A, F, act, nout, discrete)
    if static:
        rc = lib._lib.madiff_loss_fwd_static(*common, out.data_ptr(), out_f.data_ptr(), st)
    else:
        sums = torch.empty(B, 3, dtype=torch.float32, device=dev)
        ws = torch.empty(lib._lib.madiff_loss_ws_doubles(), dtype=torch.float64, device=dev)
        rc = lib._lib.madiff_loss_fwd(*common, sums.data_ptr(), ws.data_ptr(), out.data_ptr(), out_f.data_ptr(), st)
    torch.cuda.synchronize()
    return rc, out.cpu(), out_f.cpu()


@pytest.mark.parametrize("discrete", [False, True], ids=["continuous", "discrete"])
def test_static_loss_matches_workspace_form(discrete):
    from madiff_amd import lib
    dev = torch.device("cuda:0")
    act, nout = (1, 14) if discrete else (5, 5)
    assert lib._lib.madiff_loss_static_init() == 0
    for (B, T, A, F, seed) in ((96, 24, 4, 16, 1), (7, 12, 3, 18, 2)):     # the second call is smaller than the first
        tensors, geom = _case(B, T, A, F, act, nout, discrete, seed, dev)
        rc0, ref, ref_f = _run(lib, tensors, geom, static=False)
        before = torch.cuda.memory_allocated()
        rc1, got, got_f = _run(lib, tensors, geom, static=True)
        assert rc0 == 0 and rc1 == 0
        assert torch.isfinite(ref).all() and float(ref[0]) > 0
        assert torch.equal(got, ref) and torch.equal(got_f, ref_f), (B, got, ref)
        assert torch.cuda.memory_allocated() == before                      # (outputs were allocated by the test before)
    # capacity: more samples than the static array holds -> refused, nothing launched, outputs untouched
    tensors, geom = _case(2, 4, 2, 4, act, nout, discrete, 3, dev)
    big = (16385,) + geom[1:]
    rc, out, out_f = _run(lib, tensors, big, static=True)
    assert rc == -2 and torch.isnan(out).all() and torch.isnan(out_f).all()
