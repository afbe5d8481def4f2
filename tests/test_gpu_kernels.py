"""Kernel-level GPU tests: every C-ABI op (through the autograd Functions) against plain PyTorch fp32 ops of the
same mathematics, forward and backward, in both builds.  fp32 build tolerance 1e-5, bf16 build 2e-2 (relative to
the tensor's max magnitude)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL = {"fp32": 2e-5, "bf16": 2e-2}


@pytest.fixture(params=["fp32", "bf16"])
def prec(request):
    from madiff_amd import ops
    ops.set_precision(request.param)
    yield request.param
    ops.set_precision("bf16")


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def _q(x, prec):
    """Round test inputs to the storage type so the comparison isolates the kernel's arithmetic."""
    return x.to(torch.bfloat16).float() if prec == "bf16" else x


@pytest.mark.parametrize("case", ["conv5", "conv5_wide", "down", "down_odd", "up", "conv1", "linear_groups", "first_layer"])
def test_conv_gemm_fwd_bwd(prec, case):
    from madiff_amd import modules as M, ops
    dev = _dev()
    cfgs = {  # N, T, Cin, Cout(s), spec
        "conv5": (5, 12, 64, [96], ops.CONV5), "conv5_wide": (9, 6, 256, [160], ops.CONV5),
        "down": (4, 14, 32, [32], ops.DOWN), "down_odd": (3, 7, 32, [32], ops.DOWN), "up": (4, 7, 64, [64], ops.UP),
        "conv1": (3, 24, 128, [40], ops.CONV1), "linear_groups": (37, 1, 64, [16, 48, 24], ops.LINEAR),
        "first_layer": (6, 24, 18, [128], ops.CONV5),
    }
    N, T, Cin, Couts, spec = cfgs[case]
    k = spec.taps
    x = _q(_rand(N, Cin, T, seed=1), prec).to(dev).requires_grad_(True)
    ws, bs = [], []
    for i, co in enumerate(Couts):
        shape = (co, Cin, k) if spec.kind == "conv" else ((Cin, co, k) if spec.kind == "convT" else (co, Cin))
        ws.append(_q(_rand(*shape, seed=10 + i, scale=1 / math.sqrt(Cin * k)), prec).to(dev).requires_grad_(True))
        bs.append(_rand(co, seed=20 + i, scale=0.1).to(dev).requires_grad_(True))
    # reference
    refs = []
    for w, b in zip(ws, bs):
        if spec.kind == "conv":
            refs.append(F.conv1d(x, w, b, stride=spec.stride, padding=spec.pad))
        elif spec.kind == "convT":
            refs.append(F.conv_transpose1d(x, w, b, stride=spec.stride, padding=spec.pad))
        else:
            refs.append(F.linear(x[:, :, 0], w, b)[:, :, None])
    ref = torch.cat(refs, dim=1)
    gy = _q(_rand(*ref.shape, seed=3), prec).to(dev)
    gref = torch.autograd.grad(ref, [x] + ws + bs, gy)
    # ours
    x2 = x.detach().clone().requires_grad_(True)
    ws2 = [w.detach().clone().requires_grad_(True) for w in ws]
    bs2 = [b.detach().clone().requires_grad_(True) for b in bs]
    y = ops.conv_gemm(M.to_cl(x2), ws2, bs2, spec, T)
    T2 = spec.out_len(T)
    out = M.from_cl(y, N, T2, sum(Couts))
    assert out.shape == ref.shape
    assert rel_err(out.detach().cpu(), ref.detach().cpu()) < TOL[prec]
    if y.shape[1] > sum(Couts):
        assert float(y[:, sum(Couts):].abs().max()) == 0.0      # zero-padding invariant
    gours = torch.autograd.grad(out, [x2] + ws2 + bs2, gy)
    for a, b, name in zip(gours, gref, ["dx"] + ["dw"] * len(ws) + ["db"] * len(bs)):
        assert a.shape == b.shape
        assert rel_err(a.cpu(), b.cpu()) < TOL[prec] * (3 if prec == "bf16" else 1), name


@pytest.mark.parametrize("case", ["conv5", "conv5_t6_tiles", "conv5_t12_tiles", "conv5_t24", "conv5_t8", "down", "up", "down_parity",
                                  "up_parity", "conv1_ragged", "conv1_xcd"])
def test_gemm_tile_variants_and_dgrad_from_forward_weights(case):
    """Every output tile of the LDS-DMA kernel gives bit-identical results (forward), and the data gradient computed
    straight from the forward weight layout (transposing LDS reads) matches autograd, for each of its tiles."""
    from madiff_amd import lib, modules as M, ops
    ops.set_precision("bf16")
    dev = _dev()
    # (conv5_t6_tiles / conv5_t12_tiles: several output tiles and K-steps, the last one partly filled, at the sequence
    # lengths where the all-taps kernels order their rows time-major and skip the taps that leave the sequence)
    cfgs = {"conv5": (9, 6, 192, 128, ops.CONV5), "conv5_t6_tiles": (75, 6, 128, 128, ops.CONV5),
            "conv5_t12_tiles": (37, 12, 64, 128, ops.CONV5), "conv5_t24": (7, 24, 128, 192, ops.CONV5),
            "conv5_t8": (13, 8, 64, 64, ops.CONV5), "down": (5, 12, 64, 256, ops.DOWN), "up": (7, 6, 128, 128, ops.UP),
            "conv1_ragged": (11, 7, 320, 192, ops.CONV1),
            # transposed stride-2 maps with several tiles per output parity (parity-major rows: tiles of one parity skip
            # every second tap, the tile that straddles the parity boundary runs them all)
            "down_parity": (41, 12, 64, 128, ops.DOWN), "up_parity": (41, 6, 64, 128, ops.UP),
            # rows >= 4 x columns: the per-tap kernel hands every XCD a contiguous run of the tile order (xcd_chunk_remap);
            # 1026 rows = 17 / 11 / 9 row tiles x 2..3 column tiles, none a multiple of 8
            "conv1_xcd": (171, 6, 64, 192, ops.CONV1)}
    N, T, Cin, Co, spec = cfgs[case]
    k = spec.taps
    x = _q(_rand(N, Cin, T, seed=1), "bf16").to(dev).requires_grad_(True)
    shape = (Co, Cin, k) if spec.kind == "conv" else (Cin, Co, k)
    w = _q(_rand(*shape, seed=2, scale=1 / math.sqrt(Cin * k)), "bf16").to(dev).requires_grad_(True)
    ref = F.conv1d(x, w, None, stride=spec.stride, padding=spec.pad) if spec.kind == "conv" else \
        F.conv_transpose1d(x, w, None, stride=spec.stride, padding=spec.pad)
    T2 = spec.out_len(T)
    gy = _q(_rand(*ref.shape, seed=3), "bf16").to(dev)
    gx_ref, gw_ref = torch.autograd.grad(ref, [x, w], gy)
    xcl = M.to_cl(x.detach())
    Wf = ops._packed([w.detach()], spec, False)
    st = torch.cuda.current_stream().cuda_stream
    # variants 0..13 (one K-step per tap and channel chunk) share one reduction order: bit-identical.  Variants 14..16
    # (all-taps k5 kernel, conv_taps_bf16.hip: channel-chunk-major order) cover stride-1 k5 layers only, agree with each
    # other bit for bit and with the others to bf16 rounding.
    FIRST_TAPS = 14
    taps_case = case.startswith("conv5")
    outs = {}
    for tile in range(lib.tile_count()):
        y = torch.full((N * T2, Co), float("nan"), dtype=torch.bfloat16, device=dev)
        rc = lib.call_rc("madiff_gemm_fwd", lib.BF16, 0, xcl.data_ptr(), xcl.stride(0), None, 0, Cin, N * T2, Cin, k, T2,
                         T, spec.stride, spec.pad, spec.tr, Wf.data_ptr(), Wf.stride(0), Co, Co, None, None, 0,
                         y.data_ptr(), Co, 0, 0, 0, tile, st)
        covered = tile < FIRST_TAPS or (taps_case and (tile != 16 or Cin % 128 == 0))   # 16: chunk pairs only
        assert rc == (0 if covered else 1), (tile, rc)
        if rc:
            continue
        outs[tile] = y
        assert rel_err(M.from_cl(y, N, T2, Co).cpu(), ref.detach().cpu()) < TOL["bf16"], tile
        base = outs[0] if tile < FIRST_TAPS else outs[FIRST_TAPS]
        assert torch.equal(y, base), f"tile {tile} differs from tile {0 if tile < FIRST_TAPS else FIRST_TAPS}"
    gycl = M.to_cl(gy)
    gxs = {}
    for tile in lib.tiles(wt=True):
        gx = torch.full((N * T, Cin), float("nan"), dtype=torch.bfloat16, device=dev)
        rc = lib.call_rc("madiff_gemm_dgrad_wt", gycl.data_ptr(), gycl.stride(0), N * T, Co, k, T, T2, spec.stride,
                         spec.pad, 1 - spec.tr, Wf.data_ptr(), Wf.stride(0), Cin, 0, Cin, Cin, gx.data_ptr(), Cin, tile, 0,
                         st)
        covered = tile < FIRST_TAPS or (taps_case and (tile != 16 or Co % 128 == 0))
        assert rc == (0 if covered else 1), (tile, rc)
        if rc:
            continue
        gxs[tile] = gx
        assert rel_err(M.from_cl(gx, N, T, Cin).cpu(), gx_ref.cpu()) < 3 * TOL["bf16"], tile
        assert torch.equal(gx, gxs[0] if tile < FIRST_TAPS else gxs[FIRST_TAPS]), tile
    # sliced reduction (fp32 atomics, fp32 output)
    gx32 = torch.full((N * T, Cin), float("nan"), dtype=torch.float32, device=dev)
    rc = lib.call_rc("madiff_gemm_dgrad_wt", gycl.data_ptr(), gycl.stride(0), N * T, Co, k, T, T2, spec.stride, spec.pad,
                     1 - spec.tr, Wf.data_ptr(), Wf.stride(0), Cin, 0, Cin, Cin, gx32.data_ptr(), Cin, -1, 3, st)
    assert rc == 0
    assert rel_err(M.from_cl(gx32, N, T, Cin).cpu(), gx_ref.cpu()) < 3 * TOL["bf16"]
    # weight gradient: every (O x C) tile and several reduction splits, overwrite and accumulate modes
    gw_ref = gw_ref.permute(0, 2, 1) if spec.kind == "conv" else gw_ref.permute(1, 2, 0)        # -> [O, taps, I]
    for tile in range(len(ops._WGRAD_TILES)):
        for splits in (1, 3):
            for accumulate in (0, 1):
                dW = torch.full((Co, k * Cin), 1.0 if accumulate else float("nan"), dtype=torch.float32, device=dev)
                rc = lib.call_rc("madiff_gemm_wgrad", lib.BF16, gycl.data_ptr(), gycl.stride(0), gycl.shape[1],
                                 xcl.data_ptr(), xcl.stride(0), None, 0, Cin, N * T2, Cin, k, T2, T, spec.stride, spec.pad,
                                 spec.tr, dW.data_ptr(), dW.stride(0), Co, accumulate, tile, splits, st)
                assert rc == (0 if (tile < 4 or taps_case) else 1), (tile, rc)
                if rc:
                    continue
                got = dW.view(Co, k, Cin) - (1.0 if accumulate else 0.0)
                assert rel_err(got.cpu(), gw_ref.cpu()) < 3 * TOL["bf16"], (tile, splits, accumulate)
                # the same launch with the bias gradient riding along (column sums of dY), both destination forms: a
                # contiguous vector and one address per output column (scattered over two buffers here); accumulating
                db_ref = gy.sum(dim=(0, 2)).cpu()
                for form in ("vector", "table"):
                    dW2 = torch.full_like(dW, 1.0 if accumulate else float("nan"))
                    db = torch.full((2, Co), 0.5, dtype=torch.float32, device=dev)
                    tab = None
                    if form == "table":
                        tab = torch.tensor([db[o % 2].data_ptr() + 4 * o for o in range(Co)], dtype=torch.int64).to(dev)
                    rc = lib.call_rc("madiff_gemm_wgrad_bias", lib.BF16, gycl.data_ptr(), gycl.stride(0), gycl.shape[1],
                                     xcl.data_ptr(), xcl.stride(0), None, 0, Cin, N * T2, Cin, k, T2, T, spec.stride, spec.pad,
                                     spec.tr, dW2.data_ptr(), dW2.stride(0), Co, accumulate, tile, splits,
                                     None if tab is not None else db[0].data_ptr(), None if tab is None else tab.data_ptr(), st)
                    assert rc == 0
                    assert rel_err((dW2.view(Co, k, Cin) - (1.0 if accumulate else 0.0)).cpu(), gw_ref.cpu()) < 3 * TOL["bf16"]
                    if form == "vector":
                        got_b, untouched = db[0] - 0.5, db[1]
                    else:
                        sel = (torch.arange(Co, device=dev) % 2).bool()
                        got_b = torch.where(sel, db[1], db[0]) - 0.5
                        untouched = torch.where(sel, db[0], db[1])
                    assert rel_err(got_b.cpu(), db_ref) < 2e-3, (tile, splits, form)
                    assert torch.all(untouched == 0.5)


def test_conv_gemm_split_source(prec):
    """conv over cat((x, skip), channels) without materialising the concat (ma_temporal.py:405)."""
    from madiff_amd import modules as M, ops
    dev = _dev()
    N, T, C1, C2, Co = 7, 6, 128, 128, 64
    a = _q(_rand(N, C1, T, seed=1), prec).to(dev).requires_grad_(True)
    b = _q(_rand(N, C2, T, seed=2), prec).to(dev).requires_grad_(True)
    w = _q(_rand(Co, C1 + C2, 5, seed=3, scale=0.03), prec).to(dev).requires_grad_(True)
    bias = _rand(Co, seed=4, scale=0.1).to(dev).requires_grad_(True)
    ref = F.conv1d(torch.cat((a, b), dim=1), w, bias, padding=2)
    gy = _q(_rand(*ref.shape, seed=5), prec).to(dev)
    gref = torch.autograd.grad(ref, [a, b, w, bias], gy)
    a2, b2, w2, bias2 = [t.detach().clone().requires_grad_(True) for t in (a, b, w, bias)]
    y = ops.conv_gemm(M.to_cl(a2), [w2], [bias2], ops.CONV5, T, x2=M.to_cl(b2))
    out = M.from_cl(y, N, T, Co)
    assert rel_err(out.detach().cpu(), ref.detach().cpu()) < TOL[prec]
    for g1, g0 in zip(torch.autograd.grad(out, [a2, b2, w2, bias2], gy), gref):
        assert rel_err(g1.cpu(), g0.cpu()) < TOL[prec] * (3 if prec == "bf16" else 1)


@pytest.mark.parametrize("mode", ["plain", "temb", "res"])
@pytest.mark.parametrize("shape", [(6, 24, 16), (5, 12, 64), (4, 6, 1024), (3, 7, 128)])
def test_gn_mish(prec, mode, shape):
    from madiff_amd import modules as M, ops
    dev = _dev()
    N, T, C = shape
    x = _q(_rand(N, C, T, seed=1, scale=2.0), prec).to(dev).requires_grad_(True)
    gamma = (_rand(C, seed=2) * 0.5 + 1).to(dev).requires_grad_(True)
    beta = _rand(C, seed=3, scale=0.3).to(dev).requires_grad_(True)
    temb = _rand(N, C, seed=4).to(dev).requires_grad_(True)
    res = _q(_rand(N, C, T, seed=5), prec).to(dev).requires_grad_(True)
    ref = F.mish(F.group_norm(x, 8, gamma, beta, eps=1e-5))
    ins = [x, gamma, beta]
    if mode == "temb":
        ref = ref + temb[:, :, None]; ins.append(temb)
    elif mode == "res":
        ref = ref + res; ins.append(res)
    gy = _q(_rand(*ref.shape, seed=6), prec).to(dev)
    gref = torch.autograd.grad(ref, ins, gy)
    x2, g2, b2, t2, r2 = [t.detach().clone().requires_grad_(True) for t in (x, gamma, beta, temb, res)]
    out = ops.GNMish.apply(M.to_cl(x2), g2, b2, t2 if mode == "temb" else None, M.to_cl(r2) if mode == "res" else None,
                           N, T, C, 8)
    o = M.from_cl(out, N, T, C)
    assert rel_err(o.detach().cpu(), ref.detach().cpu()) < TOL[prec]
    ins2 = [x2, g2, b2] + ([t2] if mode == "temb" else []) + ([r2] if mode == "res" else [])
    for a, b in zip(torch.autograd.grad(o, ins2, gy), gref):
        assert rel_err(a.cpu(), b.cpu()) < TOL[prec] * (3 if prec == "bf16" else 2)


@pytest.mark.parametrize("geom", [(3, 3, 6, 64), (2, 8, 6, 128), (2, 4, 12, 512), (5, 3, 6, 1024), (3, 8, 6, 1024),
                                  (2, 6, 7, 512), (9, 4, 14, 512)])
def test_attention_core(prec, geom):
    """(.., 512) and (.., 1024): head slices of 32 / 128 and 64 / 256 features -> the MFMA form in the bf16 build (packed
    rows; B not a multiple of the 4 or 2 batch elements per tile, A = 3 and 6 padded to 4 and 8); narrower ones: v2 / v1."""
    from madiff_amd import modules as M, ops
    dev = _dev()
    B, A, T, C = geom
    H, Cq = 4, (C // 16) * 4
    W = 2 * Cq + C
    qkv = _q(_rand(B * A * T, W, seed=1), prec).to(dev).requires_grad_(True)
    x = _q(_rand(B * A * T, C, seed=2), prec).to(dev).requires_grad_(True)
    gamma = torch.tensor([0.7], device=dev, requires_grad=True)

    def ref_fn(qkv, x, gamma):
        def heads(m, c):     # [(b a) t, c] -> h b a (d t)
            d = c // H
            return m.reshape(B, A, T, H, d).permute(3, 0, 1, 4, 2).reshape(H, B, A, d * T)
        q, k, v = heads(qkv[:, :Cq], Cq), heads(qkv[:, Cq:2 * Cq], Cq), heads(qkv[:, 2 * Cq:], C)
        att = torch.softmax(torch.einsum("hbif,hbjf->hbij", q, k) / math.sqrt(q.shape[-1]), dim=-1)
        o = torch.einsum("hbij,hbjf->hbif", att, v)                       # h b a (d t)
        o = o.reshape(H, B, A, C // H, T).permute(1, 2, 4, 0, 3).reshape(B * A * T, C)
        return x + gamma * o

    ref = ref_fn(qkv, x, gamma)
    gy = _q(_rand(*ref.shape, seed=3), prec).to(dev)
    gref = torch.autograd.grad(ref, [qkv, x, gamma], gy)
    q2, x2, g2 = [t.detach().clone().requires_grad_(True) for t in (qkv, x, gamma)]
    out = ops.AttnCore.apply(q2.to(ops.act_dtype()), x2.to(ops.act_dtype()), g2, B, A, T, H, Cq, C, True)
    assert rel_err(out.detach().float().cpu(), ref.detach().cpu()) < TOL[prec]
    for a, b in zip(torch.autograd.grad(out, [q2, x2, g2], gy.to(ops.act_dtype())), gref):
        assert rel_err(a.float().cpu(), b.cpu()) < TOL[prec] * (3 if prec == "bf16" else 2)
    if prec == "bf16":
        assert ops.lib._lib.madiff_attn_mfma_covers(ops.lib.BF16, A, T, H, Cq, C) == int(C >= 512)
    # the row-bias gradient of the fused projection through the C ABI: drb[(b,a), col] = sum_t dqkv[(b,a,t), col]
    qk = q2.detach().to(ops.act_dtype()).contiguous()
    g = gy.to(ops.act_dtype()).contiguous()
    dqkv = torch.empty_like(qk)
    dgam = torch.zeros(1, device=dev)
    drb = torch.full((B * A, W), float("nan"), device=dev)
    ops.lib.call("madiff_attn_bwd_rb", ops._dt(), qk.data_ptr(), qk.stride(0), g.data_ptr(), g.stride(0), dqkv.data_ptr(),
                 dqkv.stride(0), g2.data_ptr(), dgam.data_ptr(), drb.data_ptr(), drb.stride(0), B, A, T, H, Cq, C, 1,
                 ops._stream())
    assert rel_err(dqkv.float().cpu(), gref[0].cpu()) < TOL[prec] * (3 if prec == "bf16" else 2)
    assert abs(float(dgam) - float(gref[2])) < TOL[prec] * 3 * max(abs(float(gref[2])), 1.0)
    want = gref[0].reshape(B * A, T, W).sum(dim=1)
    assert rel_err(drb.cpu(), want.cpu()) < TOL[prec] * (3 if prec == "bf16" else 2)


def test_pack_unpack_noise_inpaint(prec):
    from madiff_amd import ops
    from madiff_amd.schedulers import DDPMScheduler
    dev = _dev()
    B, T, A, Fd = 3, 6, 4, 18
    x = _rand(B, T, A, Fd, seed=1).to(dev)
    cl = ops.pack_traj(x, 32)
    assert cl.shape == (B * A * T, 32) and float(cl[:, Fd:].abs().max()) == 0
    back = ops.unpack_traj(cl, B, T, A, Fd)
    assert rel_err(back.cpu(), x.cpu()) < (1e-7 if prec == "fp32" else 8e-3)
    s = DDPMScheduler(num_train_timesteps=200)
    noise = _rand(B, T, A, Fd, seed=2).to(dev)
    t = torch.tensor([0, 57, 199], device=dev)
    cond = _rand(B, T, A, Fd, seed=3).to(dev)
    mask = (torch.rand(B, T, A, Fd, generator=torch.Generator().manual_seed(4)) < 0.3).to(dev)
    xn = torch.empty_like(x)
    ops.pack_traj(x, 32, noise=noise, t=t, sqrt_ac=(s.alphas_cumprod ** 0.5).to(dev),
                  sqrt_1mac=((1 - s.alphas_cumprod) ** 0.5).to(dev), cond=cond, mask=mask.to(torch.uint8), x_noisy=xn)
    ref = torch.where(mask, cond, s.add_noise(x, noise, t))
    assert rel_err(xn.cpu(), ref.cpu()) < 1e-6


def test_posterior_and_mse():
    from madiff_amd import ops
    from madiff_amd.schedulers import DDIMScheduler, DDPMScheduler
    from oracle import madiff_oracle as O
    dev = _dev()
    B, T, A, Fd = 2, 6, 3, 18
    x, ec, eu, z, cond = [_rand(B, T, A, Fd, seed=i, scale=1.5).to(dev) for i in range(5)]
    mask = (torch.rand(B, T, A, Fd, generator=torch.Generator().manual_seed(9)) < 0.2).to(dev)
    s, o = DDPMScheduler(num_train_timesteps=200), O.DDPMSchedule(200)
    coef = s.coef_table().to(dev)
    for i, t in enumerate(s.timesteps.tolist()):
        if t not in (199, 100, 1, 0):
            continue
        out = torch.empty_like(x)
        ops.posterior_step(x, ec, eu, 1.2, z, cond, mask.to(torch.uint8), coef, out, step_val=i)
        eps = eu + 1.2 * (ec - eu)
        ref = torch.where(mask.cpu(), cond.cpu(), o.step(eps.cpu(), t, x.cpu(), z.cpu()))
        assert rel_err(out.cpu(), ref) < 2e-6, t
    d = DDIMScheduler(num_train_timesteps=200); d.set_timesteps(15)
    od = O.DDIMSchedule(200, 15)
    cd = d.coef_table().to(dev)
    state = torch.tensor([3], dtype=torch.int32, device=dev)
    out = torch.empty_like(x)
    ops.posterior_step(x, ec, None, 0.0, None, None, None, cd, out, step_ptr=state)
    assert rel_err(out.cpu(), od.step(ec.cpu(), d.timesteps[3].item(), x.cpu())) < 2e-6
    # masked weighted MSE, forward + backward, against the reference formula (float64 masks)
    pred = _rand(B, T, A, Fd, seed=11).to(dev).requires_grad_(True)
    targ = _rand(B, T, A, Fd, seed=12).to(dev)
    w = (torch.rand(T, A, Fd, generator=torch.Generator().manual_seed(13)) + 0.5).to(dev)
    m = (torch.rand(B, T, A, 1, generator=torch.Generator().manual_seed(14)) < 0.7).double().to(dev)
    l = (pred - targ) ** 2 * w
    ref = ((l * m).mean(dim=[1, 2]) / m.mean(dim=[1, 2])).mean()
    gref, = torch.autograd.grad(ref, pred)
    p2 = pred.detach().clone().requires_grad_(True)
    loss, a0 = ops.MaskedStateLoss.apply(p2, targ, w, m.float().reshape(B, T, A).contiguous())
    assert loss.dtype == torch.float64 and abs(loss.item() - ref.item()) < 1e-6 * abs(ref.item())
    assert abs(a0.item() - l.mean().item()) < 2e-6 * abs(l.mean().item())
    g2, = torch.autograd.grad(loss, p2)
    assert rel_err(g2.cpu(), gref.cpu()) < 1e-5


def test_posterior_step_device_noise_matches_philox_oracle():
    """madiff_posterior_step_rng: (a) with an explicit per-step table it equals madiff_posterior_step on that row;
    (b) with rng = {seed, stream} its noise is Philox4x32-10 + Box-Muller word for word (oracle.philox_normal, pinned by the
    published known-answer vectors in tests/test_scheduler.py): coefficient row {0, 1, 0, 0, 0, 1} makes out = z."""
    from madiff_amd import ops
    from madiff_amd.schedulers import DDPMScheduler
    from oracle import madiff_oracle as O
    dev = _dev()
    B, T, A, Fd = 3, 6, 3, 17                                   # 918 elements: not a multiple of 4 (ragged last quad)
    x, ec, eu, cond = [_rand(B, T, A, Fd, seed=i, scale=1.5).to(dev) for i in range(4)]
    mask = (torch.rand(B, T, A, Fd, generator=torch.Generator().manual_seed(9)) < 0.2).to(dev).to(torch.uint8)
    coef = DDPMScheduler(num_train_timesteps=200).coef_table().to(dev)
    z_all = torch.stack([_rand(B, T, A, Fd, seed=20 + i) for i in range(4)]).to(dev)
    state = torch.tensor([2], dtype=torch.int32, device=dev)
    a, b = torch.empty_like(x), torch.empty_like(x)
    ops.posterior_step(x, ec, eu, 1.2, z_all[2].contiguous(), cond, mask, coef, a, step_val=2)
    ops.posterior_step_rng(x, ec, eu, 1.2, z_all, None, cond, mask, coef, b, step_ptr=state)
    assert torch.equal(a, b)
    unit = torch.tensor([[0, 1, 0, 0, 0, 1]] * 8, dtype=torch.float32, device=dev)
    zero = torch.zeros_like(x)
    seed, stream = 0x0123456789ABCDEF, 0x1_0000_0007
    rng = torch.tensor([seed, stream], dtype=torch.int64, device=dev)
    for step in (0, 5):
        out = torch.empty_like(x)
        ops.posterior_step_rng(zero, zero, None, 0.0, None, rng, None, None, unit, out, step_val=step)
        ref = O.philox_normal(seed, stream, step, x.numel()).reshape(x.shape)
        assert np.abs(out.cpu().numpy().astype(np.float64) - ref).max() < 2e-5        # fp32 log / sincos against float64
    big = torch.zeros(1 << 20, device=dev)
    out = torch.empty_like(big)
    ops.posterior_step_rng(big, big, None, 0.0, None, rng, None, None, unit, out, step_val=1)
    assert abs(float(out.mean())) < 4e-3 and abs(float(out.std()) - 1) < 4e-3
    assert np.abs(out[:4096].cpu().numpy() - O.philox_normal(seed, stream, 1, 4096)).max() < 2e-5


@pytest.mark.parametrize("dim,A,B", [(128, 3, 37), (32, 8, 5), (128, 1, 16)])
def test_fused_embedding_head_matches_separate_launches(dim, A, B):
    """madiff_embed_mlp_fwd (sinusoid -> time MLP, returns MLP, dropout mask / force_dropout, concatenation: one launch,
    both wide layers on MFMA) against the same head issued as separate launches (which the reference goldens pin), bf16
    build, autograd off: plain, masked, forced, and the activated classifier-free-guidance pair."""
    import torch.nn.functional as F
    from madiff_amd import modules as M, ops
    ops.set_precision("bf16")
    dev = _dev()
    net = M.TemporalUnet(horizon=8, transition_dim=6, dim=dim, dim_mults=(1, 2), returns_condition=True,
                         condition_dropout=0.25).to(dev)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for p_ in net.parameters():
            p_.copy_(torch.randn(p_.shape, generator=g).to(dev) * (1.0 / math.sqrt(max(p_.shape[-1], 1))))
    ops.params_changed()
    t = torch.randint(0, 200, (B,), generator=g).to(dev)
    r = torch.rand(B, 1, A, generator=g).to(dev)
    mask = (torch.rand(B, A, 1, generator=g) < 0.75).float().to(dev)
    with torch.no_grad():
        for kw in (dict(use_dropout=False, force_dropout=False), dict(use_dropout=True, force_dropout=False, dropout_mask=mask),
                   dict(use_dropout=False, force_dropout=True)):
            M.EMBED_FUSED = False
            ref = net.embed(t, r, A, **kw).float()
            M.EMBED_FUSED = True
            got = net.embed_fused(t, r, A, kw.get("dropout_mask"), kw["force_dropout"])
            assert got is not None and got.shape == ref.shape
            assert rel_err(got.float().cpu(), ref.cpu()) < 2e-2, kw
            if kw["force_dropout"]:
                assert float(got[:, dim:2 * dim].abs().max()) == 0.0
        M.EMBED_FUSED = False
        tc, tu = net.embed(t, r, A, False, False), net.embed(t, r, A, False, True)
        M.EMBED_FUSED = True
        both = net.embed_fused(t, r, A, None, False, act=True, twin=True)
        want = F.mish(torch.cat((tc, tu), dim=0).float())
        assert both.shape == want.shape and rel_err(both.float().cpu(), want.cpu()) < 2e-2
        assert float(both[B * A:, dim:2 * dim].abs().max()) == 0.0
    # with autograd on the head stays on the differentiable launches
    assert net.embed_fused(t, r, A) is None


def test_sinusoid_and_activations(prec):
    from madiff_amd import modules as M, ops
    from oracle import madiff_oracle as O
    dev = _dev()
    emb = M.SinusoidalPosEmb(128)
    t = torch.tensor([[0, 1, 7], [100, 199, 42]], device=dev)
    y = emb(t)
    assert rel_err(y.cpu(), O.sinusoidal_embedding(t.cpu(), 128)) < (2e-6 if prec == "fp32" else 8e-3)
    x = _q(_rand(33, 64, seed=1, scale=4), prec).to(dev).requires_grad_(True)
    ref = F.mish(x)
    gy = _q(_rand(33, 64, seed=2), prec).to(dev)
    gref, = torch.autograd.grad(ref, x, gy)
    x2 = x.detach().clone().requires_grad_(True)
    out = ops.mish(x2.to(ops.act_dtype()))
    assert rel_err(out.detach().float().cpu(), ref.detach().cpu()) < TOL[prec]
    g2, = torch.autograd.grad(out, x2, gy.to(ops.act_dtype()))
    assert rel_err(g2.cpu(), gref.cpu()) < TOL[prec]


def test_adam_ema_kernel():
    from madiff_amd import lib
    dev = _dev()
    n = 100003
    p = _rand(n, seed=1).to(dev); g = _rand(n, seed=2).to(dev)
    ref_p = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref_p], lr=2e-4)
    m = torch.zeros_like(p); v = torch.zeros_like(p); ema = p.clone()
    ema_ref = p.clone()
    for step in range(1, 4):
        ref_p.grad = g.clone() * step
        opt.step()
        ema_ref = 0.995 * ema_ref + 0.005 * ref_p.detach()
        gs = g * step
        lib.call("madiff_adam_ema", p.data_ptr(), gs.data_ptr(), m.data_ptr(), v.data_ptr(), ema.data_ptr(), n, 2e-4,
                 0.9, 0.999, 1e-8, 1 - 0.9 ** step, math.sqrt(1 - 0.999 ** step), 1.0, 1, 0.995, None, None, 0,
                 torch.cuda.current_stream().cuda_stream)
    assert rel_err(p.cpu(), ref_p.detach().cpu()) < 1e-6
    assert rel_err(ema.cpu(), ema_ref.cpu()) < 1e-6


def test_flat_optimizer_with_gradient_sink_matches_torch_adam():
    """FlatAdamEMA (flat buffers + fused Adam/EMA kernel + backward kernels accumulating straight into the flat
    gradient through ops.GradSink) against torch.optim.Adam + the reference's EMA rule on a twin model."""
    import copy
    import diffuser.models as DM
    from madiff_amd import ops
    from madiff_amd.host import batch_to_device
    from madiff_amd.trainer import FlatAdamEMA, train_step
    from oracle import madiff_oracle as O
    dev = _dev()
    ops.set_precision("fp32")
    try:
        torch.manual_seed(0)
        A, T, Fd = 3, 24, 18
        net = DM.SharedConvAttentionDeconv(horizon=T, transition_dim=Fd, dim=32, dim_mults=(1, 4, 8), n_agents=A,
                                           returns_condition=True, condition_dropout=0.25)
        dif = DM.GaussianDiffusion(net, n_agents=A, horizon=T, history_horizon=0, observation_dim=Fd, action_dim=5,
                                   n_timesteps=50, returns_condition=True, hidden_dim=64).to(dev)
        for a in net.self_attn:
            a.gamma.data.fill_(0.3)
        twin = copy.deepcopy(dif)
        ema, ema_twin = copy.deepcopy(dif), copy.deepcopy(dif)
        opt = FlatAdamEMA(dif, ema, lr=1e-3, ema_decay=0.9)
        topt = torch.optim.Adam(twin.parameters(), lr=1e-3)
        batches = [batch_to_device(O.synthetic_batch(4, T, A, Fd, 5, 0, seed=s), dev) for s in range(4)]
        g = torch.Generator().manual_seed(5)
        for step in range(2):
            draws = []
            for i in range(2):
                draws.append(dict(t=torch.randint(0, 50, (4,), generator=g).to(dev),
                                  noise=torch.randn(4, T, A, Fd, generator=g).to(dev),
                                  dropout_mask=torch.ones(4, A, 1, device=dev)))
            mb = [{**batches[2 * step + i], **draws[i]} for i in range(2)]
            train_step(dif, opt, mb, do_ema=1)
            for b in mb:
                loss, _ = twin.loss(**b)
                (loss / 2).backward()
            topt.step()
            topt.zero_grad()
            with torch.no_grad():
                for pe, pt in zip(ema_twin.parameters(), twin.parameters()):
                    pe.mul_(0.9).add_(pt, alpha=0.1)
        assert ops.SINK is not None
        # two Adam steps turn the fp32 summation-order noise of a small gradient element into a visible fraction of lr
        # (update = lr * m / sqrt(v)): measured 1.5e-4 .. 2.8e-4 depending on the channel padding of the build
        for (k, p), (_, q) in zip(dif.named_parameters(), twin.named_parameters()):
            assert rel_err(p.detach().cpu(), q.detach().cpu()) < 5e-4, k
        for (k, p), (_, q) in zip(ema.named_parameters(), ema_twin.named_parameters()):
            assert rel_err(p.detach().cpu(), q.detach().cpu()) < 5e-4, k
    finally:
        ops.SINK = None
        ops.set_precision("bf16")


@pytest.mark.parametrize("dp,topology", [(False, "default"), (True, "default"), (True, "no_wgrad_branch"),
                                         (True, "rejoin_per_bucket"), (False, "no_wgrad_branch"), (True, "bf16_transport")],
                         ids=["single", "dp_1rank_rccl", "dp_no_wgrad_branch", "dp_rejoin_per_bucket", "single_no_wgrad_branch",
                              "dp_bf16_transport"])
def test_graphed_train_step_matches_eager_steps(dp, topology):
    """GraphedTrainStep (loss fwd + bwd + fused Adam/EMA replayed from one hipGraph) against the eager train_step on a
    twin model with the same fixed draws: identical launches, so parameters and EMA must agree to rounding.
    dp: the data-parallel form of the step (graph = forward + backward with one graph-external event node per gradient
    bucket; RCCL all-reduce of each bucket issued eagerly on the communication stream after the replay; eager Adam),
    exercised in a 1-rank RCCL group (MADIFF_FORCE_DP keeps the bucket machinery on at world_size 1).
    topology: other shapes of the captured graph around the hand-made event-record nodes -- without the parameter-
    gradient branch (MADIFF_WGRAD_SIDE_STREAM=0: event nodes on the main stream, the inverse-dynamics branch joined at
    each), and with the branches rejoined into the MAIN stream at every bucket (the topology whose first version crashed
    hipStreamEndCapture in round 1 because it also waited on streams outside the capture); bf16_transport: the buckets
    travel as bf16 through a staging buffer (FlatAdamEMA(bucket_dtype="bf16")) on the RCCL path."""
    import copy
    import os
    import torch.distributed as dist
    import diffuser.models as DM
    from madiff_amd import ops
    from madiff_amd.host import batch_to_device
    from madiff_amd.trainer import FlatAdamEMA, GraphedTrainStep, train_step
    from oracle import madiff_oracle as O
    dev = _dev()
    ops.set_precision("fp32")
    side0 = ops.WGRAD_SIDE_STREAM
    ops.WGRAD_SIDE_STREAM = side0 and topology != "no_wgrad_branch"
    if dp:
        os.environ["MADIFF_FORCE_DP"] = "1"
        if not dist.is_initialized():
            dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29517", rank=0, world_size=1,
                                    device_id=dev)
    try:
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
        results = []
        for mode in ("eager", "graph"):
            m = copy.deepcopy(dif)
            ema = copy.deepcopy(m)
            opt = FlatAdamEMA(m, ema, lr=1e-3, ema_decay=0.9, bucket_mb=0.25,
                              bucket_dtype="bf16" if topology == "bf16_transport" else None)
            opt.bucket_events_on_main = topology == "rejoin_per_bucket"
            train_step(m, opt, [{**batch, **draws}], do_ema=2)          # materialises the flat buffers
            if mode == "eager":
                for i in range(4):
                    loss, _ = train_step(m, opt, [{**batch, **draws}], do_ema=i % 2)
            else:
                gs = GraphedTrainStep(m, opt, batch, draws=draws)        # its constructor performs one eager step
                assert gs.dp == dp
                for i in range(1, 4):
                    loss, _ = gs.step(batch, do_ema=i % 2)
            torch.cuda.synchronize()
            assert opt.dp == dp and (not dp or len(opt.buckets) >= 2)
            assert opt.step_count == 5
            results.append((float(loss), opt.flat_p.clone(), opt.flat_ema.clone()))
            ops.SINK = None
        (l0, p0, e0), (l1, p1, e1) = results
        # not bit-identical: several gradient reductions use fp32 atomics, whose order differs from run to run
        assert abs(l0 - l1) <= 2e-6 * abs(l0), (l0, l1)
        # (bf16 transport: an atomics-order difference can flip the bf16 rounding of a gradient element, 2^-8 relative)
        tol = 1e-3 if topology == "bf16_transport" else 3e-5
        assert rel_err(p1.cpu(), p0.cpu()) < tol
        assert rel_err(e1.cpu(), e0.cpu()) < tol
        if topology == "bf16_transport":
            assert opt._stage16 is not None and opt._stage16.dtype == torch.bfloat16
    finally:
        ops.SINK = None
        ops.WGRAD_SIDE_STREAM = side0
        ops.set_precision("bf16")
        os.environ.pop("MADIFF_FORCE_DP", None)
        if dp and dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("case", ["t6_c128", "t12_c512_x2", "t24_c128", "t7_c256_ragged"])
@pytest.mark.parametrize("mode", ["plain", "temb", "res"])
def test_conv_block_fused_matches_unfused(case, mode):
    """madiff_conv_gn_mish_fwd (all-taps k5 convolution with the GroupNorm + Mish epilogue, one launch) against the
    two-launch path (madiff_gemm_fwd + madiff_gn_mish_fwd) and against torch fp32, with and without the pre-norm output."""
    from madiff_amd import modules as M, ops
    ops.set_precision("bf16")
    dev = _dev()
    N, T, C1, C2, Co = {"t6_c128": (37, 6, 128, 0, 128), "t12_c512_x2": (19, 12, 128, 128, 512),
                        "t24_c128": (9, 24, 64, 0, 128), "t7_c256_ragged": (41, 7, 192, 0, 256)}[case]
    G = 8
    Cin = C1 + C2
    x = _q(_rand(N, Cin, T, seed=1), "bf16").to(dev)
    w = _q(_rand(Co, Cin, 5, seed=2, scale=1 / math.sqrt(Cin * 5)), "bf16").to(dev)
    b = _rand(Co, seed=3).to(dev)
    gamma = (1 + 0.3 * _rand(Co, seed=4)).to(dev)
    beta = (0.2 * _rand(Co, seed=5)).to(dev)
    rowbias = _rand(N, Co, seed=6).to(dev) if mode == "temb" else None
    res = M.to_cl(_q(_rand(N, Co, T, seed=7), "bf16").to(dev)) if mode == "res" else None
    ref = F.mish(F.group_norm(F.conv1d(x, w, b, padding=2), G, gamma, beta, eps=1e-5))
    if rowbias is not None:
        ref = ref + rowbias[:, :, None]
    if res is not None:
        ref = ref + M.from_cl(res, N, T, Co)
    xcl = M.to_cl(x)
    xa, xb = (xcl[:, :C1].contiguous(), xcl[:, C1:].contiguous()) if C2 else (xcl, None)
    old, old_tune = ops.FUSE_CONV_GN, ops._AUTOTUNE
    ops._AUTOTUNE = False           # no timing decisions here (these toy shapes would pick the separate kernels): the
    try:                            # fused launch itself is under test, with the library's default variant
        outs = {}
        for fuse in (False, True):
            ops.FUSE_CONV_GN = fuse
            for need_y in (True, False):
                r = ops._conv_block_fused(xa, xb, rowbias, res, ops.CONV5, T, G, w, b, gamma, beta, need_y) if fuse else None
                if fuse:
                    assert r is not None, "layer should be covered by the fused kernel"
                    out, y, stats = r[0], r[1], r[2]
                    assert (y is None) == (not need_y)
                else:
                    y, g = ops._conv_forward(xa, xb, None, ops.CONV5, T, 0, None, [w], [b])
                    out, stats, _ = ops._gn_forward(y, gamma, beta, rowbias, res, N, T, Co, G)
                outs[(fuse, need_y)] = (out, y, stats)
                assert rel_err(M.from_cl(out, N, T, Co).cpu(), ref.cpu()) < TOL["bf16"], (fuse, need_y)
        o0, y0, s0 = outs[(False, True)]
        o1, y1, s1 = outs[(True, True)]
        assert rel_err(y1.float().cpu(), y0.float().cpu()) < 1e-2
        assert rel_err(s1.cpu(), s0.cpu()) < 1e-2                 # statistics: fp32 accumulators vs the bf16-rounded y
        assert torch.equal(outs[(True, False)][0], o1)             # dropping the y store changes nothing else
    finally:
        ops.FUSE_CONV_GN, ops._AUTOTUNE = old, old_tune


@pytest.mark.parametrize("case", ["t6_c128", "t6_c256_tiles", "t12_c512", "t24_c128", "t7_c256_ragged"])
@pytest.mark.parametrize("variant", [1, 2])
def test_dgrad_gn_bwd_fused_matches_autograd(case, variant):
    """madiff_conv_dgrad_gn_bwd through the C ABI: data gradient of a k5 convolution B with the GroupNorm + Mish backward
    of the Conv1dBlock A that produced its input as the kernel's epilogue, against torch fp32 autograd of
    z = conv5(mish(GN(y0)) + temb): d y0, d gamma, d beta, the conv-bias gradient (column sums of d y0) and d temb.
    Both tile variants, several tiles with a partly filled last one, the time-major tile order (T = 6, T = 12 on the
    192-row tile) and the sequence-major one (T = 24, T = 7)."""
    from madiff_amd import lib, modules as M, ops
    ops.set_precision("bf16")
    dev = _dev()
    N, T, C, Co = {"t6_c128": (37, 6, 128, 128), "t6_c256_tiles": (75, 6, 256, 128), "t12_c512": (19, 12, 512, 256),
                   "t24_c128": (9, 24, 128, 128), "t7_c256_ragged": (41, 7, 256, 128)}[case]
    G = 8
    y0 = _q(_rand(N, C, T, seed=1), "bf16").to(dev).requires_grad_(True)
    wB = _q(_rand(Co, C, 5, seed=2, scale=1 / math.sqrt(C * 5)), "bf16").to(dev)
    gamma = (1 + 0.3 * _rand(C, seed=4)).to(dev).requires_grad_(True)
    beta = (0.2 * _rand(C, seed=5)).to(dev).requires_grad_(True)
    temb = _rand(N, C, seed=6).to(dev).requires_grad_(True)
    dz = _q(_rand(N, Co, T, seed=7), "bf16").to(dev)
    z = F.conv1d(F.mish(F.group_norm(y0, G, gamma, beta, eps=1e-5)) + temb[:, :, None], wB, None, padding=2)
    z.backward(dz)
    yg = y0.detach().view(N, G, (C // G) * T)
    mean = yg.mean(dim=2).reshape(-1).contiguous()
    rstd = torch.rsqrt(yg.var(dim=2, unbiased=False) + 1e-5).reshape(-1).contiguous()
    Wf = ops._packed([wB], ops.CONV5, False)
    dzcl, y0cl = M.to_cl(dz), M.to_cl(y0.detach())
    dX = torch.full((N * T, C), float("nan"), dtype=torch.bfloat16, device=dev)
    dgamma, dbeta, dbias = (torch.full((C,), 0.5, device=dev) for _ in range(3))      # accumulated into
    drb = torch.full((N, C), float("nan"), device=dev)
    rc = lib.call_rc("madiff_conv_dgrad_gn_bwd", dzcl.data_ptr(), dzcl.stride(0), N * T, Co, T, Wf.data_ptr(), Wf.stride(0), C,
                     C, y0cl.data_ptr(), y0cl.stride(0), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                     G, dX.data_ptr(), C, dgamma.data_ptr(), dbeta.data_ptr(), dbias.data_ptr(), drb.data_ptr(), C, variant,
                     torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    tol = 3 * TOL["bf16"]
    assert rel_err(M.from_cl(dX, N, T, C).float().cpu(), y0.grad.cpu()) < tol
    assert rel_err((dgamma - 0.5).cpu(), gamma.grad.cpu()) < tol
    assert rel_err((dbeta - 0.5).cpu(), beta.grad.cpu()) < tol
    assert rel_err((dbias - 0.5).cpu(), y0.grad.sum(dim=(0, 2)).cpu()) < tol
    assert rel_err(drb.cpu(), temb.grad.cpu()) < tol


@pytest.mark.parametrize("case", ["t6_c128", "t6_c256_tiles", "t12_c512", "t24_c128", "t7_c256_ragged"])
@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("with_addend", [False, True])
def test_dgrad_gn_bwd_chain_matches_autograd(case, variant, with_addend):
    """madiff_conv_dgrad_gn_bwd_chain: the form for two consecutive ResidualTemporalBlocks, where dh = dgrad + addend is
    itself a gradient that has to reach memory (dH) besides going through the preceding block's GroupNorm + Mish
    backward.  Reference: torch autograd of  <dz, conv5(o)> + <add, o>,  o = mish(GN(y0))."""
    from madiff_amd import lib, modules as M, ops
    ops.set_precision("bf16")
    dev = _dev()
    N, T, C, Co = {"t6_c128": (37, 6, 128, 128), "t6_c256_tiles": (75, 6, 256, 128), "t12_c512": (19, 12, 512, 256),
                   "t24_c128": (9, 24, 128, 128), "t7_c256_ragged": (41, 7, 256, 128)}[case]
    G = 8
    y0 = _q(_rand(N, C, T, seed=1), "bf16").to(dev).requires_grad_(True)
    wB = _q(_rand(Co, C, 5, seed=2, scale=1 / math.sqrt(C * 5)), "bf16").to(dev)
    gamma = (1 + 0.3 * _rand(C, seed=4)).to(dev).requires_grad_(True)
    beta = (0.2 * _rand(C, seed=5)).to(dev).requires_grad_(True)
    dz = _q(_rand(N, Co, T, seed=7), "bf16").to(dev)
    add = _q(_rand(N, C, T, seed=8), "bf16").to(dev)
    o = F.mish(F.group_norm(y0, G, gamma, beta, eps=1e-5))
    o.retain_grad()
    loss = (F.conv1d(o, wB, None, padding=2) * dz).sum()
    if with_addend:
        loss = loss + (o * add).sum()
    loss.backward()
    yg = y0.detach().view(N, G, (C // G) * T)
    mean = yg.mean(dim=2).reshape(-1).contiguous()
    rstd = torch.rsqrt(yg.var(dim=2, unbiased=False) + 1e-5).reshape(-1).contiguous()
    Wf = ops._packed([wB], ops.CONV5, False)
    dzcl, y0cl, addcl = M.to_cl(dz), M.to_cl(y0.detach()), M.to_cl(add)
    dX = torch.full((N * T, C), float("nan"), dtype=torch.bfloat16, device=dev)
    dH = torch.full((N * T, C), float("nan"), dtype=torch.bfloat16, device=dev)
    dgamma, dbeta, dbias = (torch.full((C,), 0.5, device=dev) for _ in range(3))
    rc = lib.call_rc("madiff_conv_dgrad_gn_bwd_chain", dzcl.data_ptr(), dzcl.stride(0), N * T, Co, T, Wf.data_ptr(),
                     Wf.stride(0), C, C, y0cl.data_ptr(), y0cl.stride(0), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(),
                     rstd.data_ptr(), G, dX.data_ptr(), C, dgamma.data_ptr(), dbeta.data_ptr(), dbias.data_ptr(),
                     addcl.data_ptr() if with_addend else None, C, dH.data_ptr(), C, variant,
                     torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    tol = 3 * TOL["bf16"]
    assert rel_err(M.from_cl(dH, N, T, C).float().cpu(), o.grad.cpu()) < tol
    assert rel_err(M.from_cl(dX, N, T, C).float().cpu(), y0.grad.cpu()) < tol
    assert rel_err((dgamma - 0.5).cpu(), gamma.grad.cpu()) < tol
    assert rel_err((dbeta - 0.5).cpu(), beta.grad.cpu()) < tol
    assert rel_err((dbias - 0.5).cpu(), y0.grad.sum(dim=(0, 2)).cpu()) < tol


@pytest.mark.parametrize("case", ["n6_t6_c1024", "n13_t12_c512", "n5_t24_c256"])
def test_all_taps_reduction_split_matches_unsplit(case):
    """Small-batch layers: the all-taps kernel cuts the input channels into slices, partial tiles meet in the registered
    workspace and the last workgroup finishes the tile (plain, fused GroupNorm forward, data gradient, fused GroupNorm
    backward).  Against the same launches with the split switched off (MADIFF_CONV_SPLITK is read once per process, so the
    reference here is torch): outputs to bf16 rounding, and the counters must be back at zero."""
    import torch.nn.functional as F
    from madiff_amd import modules as M, ops
    ops.set_precision("bf16")
    dev = _dev()
    N, T, C = {"n6_t6_c1024": (6, 6, 1024), "n13_t12_c512": (13, 12, 512), "n5_t24_c256": (5, 24, 256)}[case]
    x = _q(_rand(N, C, T, seed=1), "bf16").to(dev).requires_grad_(True)
    w = _q(_rand(C, C, 5, seed=2, scale=1 / math.sqrt(C * 5)), "bf16").to(dev).requires_grad_(True)
    b = _rand(C, seed=3, scale=0.1).to(dev).requires_grad_(True)
    gamma = (_rand(C, seed=4) * 0.5 + 1).to(dev).requires_grad_(True)
    beta = _rand(C, seed=5, scale=0.3).to(dev).requires_grad_(True)
    temb = _rand(N, C, seed=6).to(dev).requires_grad_(True)
    ref = F.mish(F.group_norm(F.conv1d(x, w, b, padding=2), 8, gamma, beta, eps=1e-5)) + temb[:, :, None]
    gy = _q(_rand(*ref.shape, seed=7), "bf16").to(dev)
    gref = torch.autograd.grad(ref, [x, w, b, gamma, beta, temb], gy)
    ops._stream()                                              # registers the workspace
    ws, cnt = ops._SPLITK_WS[torch.cuda.current_device()]
    x2, w2, b2, g2, be2, t2 = [t.detach().clone().requires_grad_(True) for t in (x, w, b, gamma, beta, temb)]
    out = ops.conv_block(M.to_cl(x2), w2, b2, g2, be2, ops.CONV5, T, 8, rowbias=t2)
    o = M.from_cl(out, N, T, C)
    assert rel_err(o.detach().cpu(), ref.detach().cpu()) < TOL["bf16"]
    for a, r in zip(torch.autograd.grad(o, [x2, w2, b2, g2, be2, t2], gy), gref):
        assert rel_err(a.cpu(), r.cpu()) < 3 * TOL["bf16"]
    torch.cuda.synchronize()
    assert int(cnt.abs().sum()) == 0
    # the split was really taken for this shape: few tiles, many chunks
    import ctypes
    assert ops.lib.exported("madiff_set_splitk_workspace")
