"""SURVEY.md 8f row F4, against goldens captured from the reference's own files (oracle/gen_golden.py gen_alternates):
non-temporal SelfAttention (helpers.py:77-124), TemporalUnet.forward (temporal.py:348-406), the wrapper denoisers
(ma_temporal_wrappers.py) and the SMAC data encoder (utils/data_encoder.py:11-20) inside GaussianDiffusion.
Tolerances as in test_gpu_parity.py."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import madiff_oracle as O
from test_gpu_parity import OUT_TOL, T, digest_check, fill, l2_err, prec  # noqa: F401

pytestmark = pytest.mark.gpu
KW = dict(dim=16, dim_mults=(1, 4, 8), returns_condition=True, condition_dropout=0.25)


def test_self_attention_vs_reference_golden(golden, prec):
    import diffuser.models as DM
    g = golden("g9_alternates")
    dev = torch.device("cuda:0")
    att = DM.SelfAttention(64, 4, 16, residual=True).to(dev)
    fill(att, "selfattn.")
    x = T(g["selfattn.x"], dev).requires_grad_(True)
    y = att(x)
    tol = OUT_TOL[prec] * (2 if prec == "fp32" else 1)
    assert rel_err(y.detach().cpu(), g["selfattn.y"]) < tol
    assert float(np.abs(g["selfattn.y"] - g["selfattn.x"]).max()) > 1e-2          # not an identity (gamma != 0)
    y.backward(T(g["selfattn.gy"], dev))
    gerr = (lambda a, b: rel_err(a, b) / 4) if prec == "fp32" else l2_err
    gtol = tol if prec == "fp32" else 0.1
    assert gerr(x.grad.cpu(), g["selfattn.gx"]) < gtol
    for k, p in att.named_parameters():
        ref = g["selfattn.gp." + k]
        if np.abs(ref).max() < 1e-4:          # key bias: analytically zero gradient
            assert float(p.grad.abs().max()) < (1e-4 if prec == "fp32" else 5e-2), k
        else:
            assert gerr(p.grad.cpu(), ref) < gtol, k


def _build(tag):
    import diffuser.models as DM
    if tag == "unet":
        return DM.TemporalUnet(horizon=24, transition_dim=18, **KW)
    if tag == "concat":
        return DM.ConcatenatedTemporalUnet(n_agents=3, horizon=24, transition_dim=6, **KW)
    if tag == "indep":
        return DM.IndependentTemporalUnet(n_agents=2, horizon=24, transition_dim=10, **KW)
    return DM.SharedIndependentTemporalUnet(n_agents=3, horizon=24, history_horizon=0, transition_dim=10, **KW)


@pytest.mark.parametrize("tag", ["unet", "concat", "indep", "shared_indep"])
def test_attention_free_denoisers_vs_reference_golden(golden, prec, tag):
    g = golden("g9_alternates")
    dev = torch.device("cuda:0")
    net = _build(tag).to(dev)
    fill(net, "model.")
    x, t, r = T(g[tag + ".x"], dev), T(g[tag + ".t"], dev), T(g[tag + ".returns"], dev)
    tol = OUT_TOL[prec]
    with torch.no_grad():
        assert rel_err(net(x, t, returns=r, use_dropout=False).cpu(), g[tag + ".y_cond"]) < tol
        assert rel_err(net(x, t, returns=r, use_dropout=False, force_dropout=True).cpu(), g[tag + ".y_uncond"]) < tol
    masks = T(g[tag + ".dropout_masks"], dev)
    dm = [m for m in masks] if tag == "indep" else masks[0]
    xg = x.clone().requires_grad_(True)
    y = net(xg, t, returns=r, use_dropout=True, dropout_mask=dm)
    assert rel_err(y.detach().cpu(), g[tag + ".y_drop"]) < tol
    y.backward(T(g[tag + ".gy"], dev))
    assert (rel_err if prec == "fp32" else l2_err)(xg.grad.cpu(), g[tag + ".gx"]) < (1e-4 if prec == "fp32" else 6e-2)
    worst = digest_check(dict(net.named_parameters()), g[tag + ".gp_names"], g[tag + ".gp_digest"], prec)
    assert worst < (1e-3 if prec == "fp32" else 0.1), worst


def test_smac_data_encoder_vs_reference_golden(golden, prec):
    """GaussianDiffusion(data_encoder=SMAC3mEncoder()): the loss (encoder after the conditioning, diffusion.py:294) and a
    K = 8 guided sample (encoder inside every loop iteration and after the last one, :252,:269)."""
    import diffuser.models as DM
    from diffuser.utils import SMAC3mEncoder
    from madiff_amd.host import batch_to_device
    g = golden("g9_alternates")
    dev = torch.device("cuda:0")
    A, Tt, Fd, act, K = 3, 24, 18, 5, 8
    net = DM.SharedConvAttentionDeconv(horizon=Tt, transition_dim=Fd, dim=16, dim_mults=(1, 4, 8), n_agents=A,
                                       returns_condition=True, condition_dropout=0.25)
    dif = DM.GaussianDiffusion(net, n_agents=A, horizon=Tt, history_horizon=0, observation_dim=Fd, action_dim=act,
                               n_timesteps=K, returns_condition=True, condition_guidance_w=1.2, hidden_dim=64,
                               data_encoder=SMAC3mEncoder()).to(dev)
    fill(dif)
    batch = batch_to_device(O.synthetic_batch(3, Tt, A, Fd, act, 0, False, 0, seed=21), dev)
    loss, info = dif.loss(**batch, t=T(g["enc.t"], dev), noise=T(g["enc.noise"], dev),
                          dropout_mask=T(g["enc.dropout_mask"], dev))
    tol = OUT_TOL[prec]
    assert abs(loss.item() - float(g["enc.loss"])) < tol * abs(float(g["enc.loss"]))
    assert abs(float(info["a0_loss"]) - float(g["enc.a0_loss"])) < tol * abs(float(g["enc.a0_loss"]))
    loss.backward()
    worst = digest_check(dict(dif.named_parameters()), g["enc.gp_names"], g["enc.gp_digest"], prec)
    assert worst < (1e-3 if prec == "fp32" else 0.1), worst
    if prec != "fp32":
        return
    shape = (2, Tt, A, Fd)
    x_init = 0.5 * T(O.closed_form_normal("g9enc.0", shape), dev)
    z = torch.stack([T(O.closed_form_normal(f"g9enc.{i}", shape)) for i in range(1, K)]).to(dev)
    cond = {"x": T(g["enc.cond_x"], dev), "masks": T(g["enc.cond_masks"], dev)}
    for use_graph in (False, True):
        x = dif.conditional_sample(cond, returns=T(g["enc.returns"], dev), verbose=False, x_init=x_init, step_noise=z,
                                   use_graph=use_graph)
        assert rel_err(x.cpu(), g["enc.sample"]) < 1e-3, use_graph
        assert torch.equal(x[:, 1:, :, :3], x[:, :1, :, :3].expand(-1, Tt - 1, -1, -1))     # the encoder's invariant


def test_wrapper_inside_gaussian_diffusion_runs():
    """A wrapper denoiser behind GaussianDiffusion's generic path (loss + backward + a short guided sample)."""
    import diffuser.models as DM
    from madiff_amd.host import batch_to_device
    dev = torch.device("cuda:0")
    A, Tt, Fd, act, K = 3, 24, 10, 5, 4
    net = DM.SharedIndependentTemporalUnet(n_agents=A, horizon=Tt, history_horizon=0, transition_dim=Fd, **KW)
    dif = DM.GaussianDiffusion(net, n_agents=A, horizon=Tt, history_horizon=0, observation_dim=Fd, action_dim=act,
                               n_timesteps=K, returns_condition=True, hidden_dim=64).to(dev)
    batch = batch_to_device(O.synthetic_batch(2, Tt, A, Fd, act, 0, False, 0, seed=5), dev)
    loss, info = dif.loss(**batch)
    loss.backward()
    assert torch.isfinite(loss) and all(p.grad is not None and torch.isfinite(p.grad).all() for p in dif.parameters())
    x = dif.conditional_sample(batch["cond"], returns=batch["returns"], verbose=False)
    m = batch["cond"]["masks"]
    assert torch.isfinite(x).all() and torch.equal(x[m], batch["cond"]["x"][m])
