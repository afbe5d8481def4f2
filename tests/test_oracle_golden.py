"""Pin the CPU oracle against golden vectors captured from the reference's own modules.

The fixtures were produced by oracle/gen_golden.py (it imports /root/reference's model files under
stubs); this test needs only tests/golden and runs without a GPU.
"""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import madiff_oracle as O

SHAPES = {"spread": (3, 24, 18, 0), "tag": (3, 24, 16, 0), "ant": (4, 28, 54, 18), "smac": (8, 24, 88, 20)}
TOL = 2e-6  # same fp32 arithmetic on the same CPU, different op grouping only


def T(a):
    return torch.from_numpy(np.asarray(a))


def den_cfg(name, width):
    A, Tt, Fd, hh = SHAPES[name]
    return O.DenoiserCfg(horizon=Tt, transition_dim=Fd, dim=width, history_horizon=hh, dim_mults=(1, 4, 8),
                         n_agents=A, returns_condition=True, condition_dropout=0.25)


def dif_cfg(name, K, discrete=False, share_inv=True, hidden=64):
    A, Tt, Fd, hh = SHAPES[name]
    act = 1 if discrete else {"spread": 5, "tag": 5, "ant": 2, "smac": 1}[name]
    return O.DiffusionCfg(n_agents=A, horizon=Tt - hh, history_horizon=hh, observation_dim=Fd, action_dim=act,
                          n_timesteps=K, discrete_action=discrete, num_actions=14 if discrete else 0,
                          hidden_dim=hidden, returns_condition=True, condition_guidance_w=1.2, share_inv=share_inv)


def full_params(dcfg, c):
    den = O.closed_form_state({"model." + k: v for k, v in O.denoiser_param_shapes(dcfg).items()})
    inv = O.closed_form_state(O.inv_param_shapes(c))
    return {**den, **inv}


def test_posemb(golden):
    g = golden("g1_modules")
    y = O.sinusoidal_embedding(T(g["posemb.t"]), 32)
    assert rel_err(y, g["posemb.y"]) < 1e-6


@pytest.mark.parametrize("tag,key", [("cblock", "cblock"), ("rtb_proj", "rtb_proj"), ("rtb_same", "rtb_same")])
def test_leaf_blocks(golden, tag, key):
    g = golden("g1_modules")
    names = [k[len(tag) + 4:] for k in g if k.startswith(tag + ".gp.")]
    p = {key + "." + n: O.closed_form_param(key + "." + n, g[f"{tag}.gp.{n}"].shape) for n in names}
    x = T(g[tag + ".in0"])
    if tag == "cblock":
        y = O.conv_gn_mish(p, key, x)
    else:
        y = O.residual_block(p, key, x, T(g[tag + ".in1"]))
    assert rel_err(y, g[tag + ".y"]) < TOL


def test_attention_leaf(golden):
    g = golden("g1_modules")
    names = [k[len("attn.gp."):] for k in g if k.startswith("attn.gp.")]
    p = {"attn." + n: O.closed_form_param("attn." + n, g[f"attn.gp.{n}"].shape) for n in names}
    y = O.agent_attention(p, "attn", T(g["attn.in0"]), T(g["attn.in1"]))
    assert rel_err(y, g["attn.y"]) < TOL
    assert float(np.abs(g["attn.y"] - g["attn.in0"]).max()) > 1e-2  # gamma != 0: the layer is not an identity


@pytest.mark.parametrize("name", list(SHAPES))
def test_denoiser_small(golden, name):
    g = golden("g2_denoiser_w16")
    cfg = den_cfg(name, 16)
    p = O.closed_form_state({"model." + k: v for k, v in O.denoiser_param_shapes(cfg).items()})
    p = {k[len("model."):]: v for k, v in p.items()}
    x, t, r = T(g[name + ".x"]), T(g[name + ".t"]), T(g[name + ".returns"])
    assert rel_err(O.denoiser_forward(p, cfg, x, t, r), g[name + ".y_cond"]) < TOL
    assert rel_err(O.denoiser_forward(p, cfg, x, t, r, None, True), g[name + ".y_uncond"]) < TOL
    assert rel_err(O.denoiser_forward(p, cfg, x, t, r, T(g[name + ".dropout_mask"])), g[name + ".y_drop"]) < TOL
    # gradients (autograd through the oracle) against the reference's digests
    pg = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xg = x.clone().requires_grad_(True)
    y = O.denoiser_forward(pg, cfg, xg, t, r, T(g[name + ".dropout_mask"]))
    y.backward(T(g[name + ".gy"]))
    assert rel_err(xg.grad, g[name + ".gx"]) < 1e-5
    dig = {n: row for n, row in zip(g[name + ".gp_names"], g[name + ".gp_digest"])}
    for n in g[name + ".nograd_names"]:
        assert pg[str(n)].grad is None and str(n).startswith("self_attn.3.")
    worst = 0.0
    for n, row in dig.items():
        gr = pg[str(n)].grad.double()
        # key biases have an analytically ZERO gradient (softmax shift invariance): allow fp32 noise
        worst = max(worst, abs(gr.pow(2).sum().sqrt().item() - row[2]) / max(row[2], 2e-2))
    assert worst < 5e-4, worst   # gradients: fp32 vs fp32 with different op grouping (outputs are held to TOL)
    for k in g:
        if k.startswith(name + ".gp.") :
            assert rel_err(pg[k[len(name) + 4:]].grad, g[k]) < 2e-5, k


def test_denoiser_full_width(golden):
    g = golden("g3_denoiser_w128")
    cfg = den_cfg("spread", 128)
    shapes = O.denoiser_param_shapes(cfg)
    assert sum(int(np.prod(s)) for s in shapes.values()) == 64342550  # 64.34 M parameters
    p = {k: O.closed_form_param("model." + k, s) for k, s in shapes.items()}
    x, t, r = T(g["spread.x"]), T(g["spread.t"]), T(g["spread.returns"])
    assert rel_err(O.denoiser_forward(p, cfg, x, t, r), g["spread.y_cond"]) < 5e-6
    assert rel_err(O.denoiser_forward(p, cfg, x, t, r, None, True), g["spread.y_uncond"]) < 5e-6


@pytest.mark.parametrize("tag,name,discrete,share", [("spread", "spread", False, True), ("ant", "ant", False, True),
                                                      ("smac", "smac", True, True),
                                                      ("spread_indep", "spread", False, False)])
def test_loss(golden, tag, name, discrete, share):
    g = golden("g4_loss")
    A, Tt, Fd, hh = SHAPES[name]
    dcfg, c = den_cfg(name, 16), dif_cfg(name, 200, discrete, share)
    p = full_params(dcfg, c)
    batch = O.synthetic_batch(3, Tt, A, Fd, c.action_dim, hh, discrete, c.num_actions, seed=11, visible=g[tag + ".visible"])
    assert rel_err(O.loss_weights(c), g[tag + ".loss_weights"]) == 0.0
    pg = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    loss, info, _ = O.diffusion_loss(pg, dcfg, c, O.DDPMSchedule(200), batch["x"], batch["cond"], batch["loss_masks"],
                                     T(g[tag + ".t"]), T(g[tag + ".noise"]), batch["returns"],
                                     T(g[tag + ".dropout_mask"]), batch.get("legal_actions"))
    assert loss.dtype == torch.float64                       # loss_masks is float64 (sequence.py:314)
    assert abs(loss.item() - float(g[tag + ".loss"])) < 2e-6 * abs(float(g[tag + ".loss"]))
    for k in ("a0_loss", "inv_loss", "inv_acc"):
        if f"{tag}.info.{k}" in g:
            assert abs(float(info[k]) - float(g[f"{tag}.info.{k}"])) <= 2e-6 * abs(float(g[f"{tag}.info.{k}"])) + 1e-12, k
    loss.backward()
    for n, row in zip(g[tag + ".gp_names"], g[tag + ".gp_digest"]):
        gr = pg[str(n)].grad.double()
        assert abs(gr.pow(2).sum().sqrt().item() - row[2]) <= 5e-4 * row[2] + 2e-6, n


@pytest.mark.parametrize("tag,name,width,K", [("small_k200", "spread", 16, 200), ("smac_k20", "smac", 16, 20),
                                               ("full_k8", "spread", 128, 8)])
def test_sampling(golden, tag, name, width, K):
    g = golden("g6_sample")
    A, Tt, Fd, hh = SHAPES[name]
    dcfg, c = den_cfg(name, width), dif_cfg(name, K, hidden=64 if width < 128 else 256)
    p = full_params(dcfg, c)
    shape = (2, Tt, A, Fd)
    x_init = 0.5 * T(O.closed_form_normal(f"{tag}.0", shape))
    z = [T(O.closed_form_normal(f"{tag}.{i}", shape)) for i in range(1, K)]
    cond = {"x": T(g[tag + ".cond_x"]), "masks": T(g[tag + ".cond_masks"])}
    with torch.no_grad():
        x, chain = O.conditional_sample(p, dcfg, c, O.DDPMSchedule(K), cond, T(g[tag + ".returns"]), x_init, z, True)
    sel = g[tag + ".chain_idx"]
    assert rel_err(chain[:, sel], g[tag + ".chain"]) < 1e-4
    assert rel_err(x, g[tag + ".x"]) < 1e-4
    m = g[tag + ".cond_masks"]
    assert np.array_equal(x.numpy()[m], g[tag + ".cond_x"][m])   # inpainting exactly enforced


def test_ddim(golden):
    g = golden("g6_sample")
    tag = "small_k200"
    A, Tt, Fd, hh = SHAPES["spread"]
    dcfg, c = den_cfg("spread", 16), dif_cfg("spread", 200)
    p = full_params(dcfg, c)
    x_init = 0.5 * T(O.closed_form_normal("ddim15.0", (2, Tt, A, Fd)))
    cond = {"x": T(g[tag + ".cond_x"]), "masks": T(g[tag + ".cond_masks"])}
    with torch.no_grad():
        x = O.conditional_sample(p, dcfg, c, O.DDIMSchedule(200, 15), cond, T(g[tag + ".returns"]), x_init, [])
    assert rel_err(x, g["ddim15.x"]) < 1e-4


# ---------------------------------------------------------------- round-2 fixtures: full width with gradients, env_ts
SHAPES["tag4"] = (4, 24, 16, 0)          # BASELINE.json configs[1] as named (4 agents): the shape bench.py runs


def sample_index(numel, n=96):
    """Positions oracle/gen_golden.py sampled every parameter gradient at."""
    i = np.arange(n, dtype=np.uint64)
    return ((i * np.uint64(2654435761) + np.uint64(40503)) % np.uint64(numel)).astype(np.int64)


def test_env_ts_condition(golden):
    """env_ts_condition=True (ma_temporal.py:355-362): embedding lookup at env_timestep[:, history_horizon]."""
    g = golden("g8_env_ts")
    A, Tt, Fd, hh = SHAPES["ant"]
    cfg = O.DenoiserCfg(horizon=Tt, transition_dim=Fd, dim=16, history_horizon=hh, dim_mults=(1, 4, 8), n_agents=A,
                        returns_condition=True, env_ts_condition=True, condition_dropout=0.25,
                        max_path_length=int(g["max_path_length"]))
    p = {k: O.closed_form_param("model." + k, s) for k, s in O.denoiser_param_shapes(cfg).items()}
    x, t, r, ets = T(g["x"]), T(g["t"]), T(g["returns"]), T(g["env_ts"])
    assert rel_err(O.denoiser_forward(p, cfg, x, t, r, env_timestep=ets), g["y_cond"]) < TOL
    assert rel_err(O.denoiser_forward(p, cfg, x, t, r, None, True, env_timestep=ets), g["y_uncond"]) < TOL
    pg = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    y = O.denoiser_forward(pg, cfg, x, t, r, T(g["dropout_mask"]), env_timestep=ets)
    assert rel_err(y.detach(), g["y_drop"]) < TOL
    y.backward(T(g["gy"]))
    for k in g:
        if k.startswith("gp."):
            assert rel_err(pg[k[3:]].grad, g[k]) < 2e-5, k
    assert int(g["env_ts"].max()) == int(g["max_path_length"])      # the pad id is exercised


@pytest.mark.parametrize("name", ["spread", "tag4"])
def test_denoiser_full_width_gradients(golden, name):
    g = golden("g3g_denoiser_w128_grads")
    cfg = den_cfg(name, 128)
    p = {k: O.closed_form_param("model." + k, s).requires_grad_(True) for k, s in O.denoiser_param_shapes(cfg).items()}
    x, t, r = T(g[name + ".x"]).requires_grad_(True), T(g[name + ".t"]), T(g[name + ".returns"])
    y = O.denoiser_forward(p, cfg, x, t, r, T(g[name + ".dropout_mask"]))
    assert rel_err(y.detach(), g[name + ".y_drop"]) < 5e-6
    y.backward(T(g[name + ".gy"]))
    assert rel_err(x.grad, g[name + ".gx"]) < 2e-5
    for n, row, smp in zip(g[name + ".gp_names"], g[name + ".gp_digest"], g[name + ".gp_samples"]):
        gr = p[str(n)].grad
        assert abs(gr.double().pow(2).sum().sqrt().item() - row[2]) <= 5e-4 * row[2] + 2e-6, n
        got = gr.reshape(-1)[torch.from_numpy(sample_index(gr.numel()))].numpy()
        assert np.abs(got - smp).max() <= 2e-4 * np.abs(gr.numpy()).max() + 3e-6, n   # (+ fp32 noise: key biases have zero gradient)
    for n in g[name + ".nograd_names"]:
        assert p[str(n)].grad is None


def test_loss_full_width(golden):
    g = golden("g4w_loss_w128")
    tag = "tag4"
    A, Tt, Fd, hh = SHAPES[tag]
    dcfg, c = den_cfg(tag, 128), dif_cfg("tag", 200, hidden=256)
    c.n_agents = A
    p = full_params(dcfg, c)
    batch = O.synthetic_batch(4, Tt, A, Fd, c.action_dim, hh, False, 0, seed=13)
    pg = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    loss, info, _ = O.diffusion_loss(pg, dcfg, c, O.DDPMSchedule(200), batch["x"], batch["cond"], batch["loss_masks"],
                                     T(g[tag + ".t"]), T(g[tag + ".noise"]), batch["returns"], T(g[tag + ".dropout_mask"]))
    assert abs(loss.item() - float(g[tag + ".loss"])) < 5e-6 * abs(float(g[tag + ".loss"]))
    loss.backward()
    for n, row in zip(g[tag + ".gp_names"], g[tag + ".gp_digest"]):
        gr = pg[str(n)].grad.double()
        assert abs(gr.pow(2).sum().sqrt().item() - row[2]) <= 1e-3 * row[2] + 2e-6, n


# ---------------------------------------------------------------- round-3 fixture: full width on the SMAC / ant shapes
def U(name, shape, lo=-1.0, hi=1.0):
    """Closed-form inputs of oracle/gen_golden.py gen_round3 (its u())."""
    return torch.from_numpy(O.closed_form_uniform("input:" + name, shape, lo, hi).astype(np.float32))


@pytest.mark.parametrize("name", ["smac", "ant"])
def test_denoiser_full_width_smac_ant(golden, name):
    """dim 128 on BASELINE configs[3] (A=8, T=24, F=88, d_qk 64 / d_v 256 attention) and configs[2] (A=4, T=28: 28/14/7)."""
    g = golden("g11_w128_smac_ant")
    A, Tt, Fd, hh = SHAPES[name]
    cfg = den_cfg(name, 128)
    p = {k: O.closed_form_param("model." + k, s).requires_grad_(True) for k, s in O.denoiser_param_shapes(cfg).items()}
    pre = f"den.{name}"
    x = U(f"g11.{name}.x", (2, Tt, A, Fd)).requires_grad_(True)
    r = U(f"g11.{name}.ret", (2, 1, A), 0.0, 1.0)
    y = O.denoiser_forward(p, cfg, x, T(g[pre + ".t"]), r, T(g[pre + ".dropout_mask"]))
    assert rel_err(y.detach(), g[pre + ".y_drop"]) < 5e-6
    y.backward(U(f"g11.{name}.gy", tuple(y.shape)))
    assert rel_err(x.grad, g[pre + ".gx"]) < 2e-5
    # key biases have an analytically zero gradient (softmax shift invariance): what both sides hold there is fp32
    # summation noise, which grows with the agent count -- the absolute floor scales with the largest gradient
    floor = max(2e-6, 1e-6 * float(g[pre + ".gp_digest"][:, 2].max()))
    for n, row, smp in zip(g[pre + ".gp_names"], g[pre + ".gp_digest"], g[pre + ".gp_samples"]):
        gr = p[str(n)].grad
        assert abs(gr.double().pow(2).sum().sqrt().item() - row[2]) <= 5e-4 * row[2] + floor, n
        got = gr.reshape(-1)[torch.from_numpy(sample_index(gr.numel()))].numpy()
        assert np.abs(got - smp).max() <= 2e-4 * np.abs(gr.numpy()).max() + floor, n
    for n in g[pre + ".nograd_names"]:
        assert p[str(n)].grad is None


@pytest.mark.parametrize("name,discrete", [("smac", True), ("ant", False)])
def test_loss_full_width_smac_ant(golden, name, discrete):
    g = golden("g11_w128_smac_ant")
    pre = f"loss.{name}"
    A, Tt, Fd, hh = SHAPES[name]
    dcfg, c = den_cfg(name, 128), dif_cfg(name, 200, discrete=discrete, hidden=256)
    pg = {k: v.clone().requires_grad_(True) for k, v in full_params(dcfg, c).items()}
    batch = O.synthetic_batch(2, Tt, A, Fd, c.action_dim, hh, discrete, 14 if discrete else 0, seed=17,
                              visible=g[pre + ".visible"])
    loss, info, _ = O.diffusion_loss(pg, dcfg, c, O.DDPMSchedule(200), batch["x"], batch["cond"], batch["loss_masks"],
                                     T(g[pre + ".t"]), T(g[pre + ".noise"]), batch["returns"], T(g[pre + ".dropout_mask"]),
                                     legal_actions=batch.get("legal_actions"))
    assert abs(loss.item() - float(g[pre + ".loss"])) < 5e-6 * abs(float(g[pre + ".loss"]))
    for k in ("a0_loss", "inv_loss", "inv_acc"):
        if f"{pre}.info.{k}" in g:
            assert abs(float(info[k]) - float(g[f"{pre}.info.{k}"])) <= 5e-6 * abs(float(g[f"{pre}.info.{k}"])) + 1e-12, k
    assert (f"{pre}.info.inv_acc" in g) == discrete
    loss.backward()
    floor = max(2e-6, 1e-6 * float(g[pre + ".gp_digest"][:, 2].max()))
    for n, row in zip(g[pre + ".gp_names"], g[pre + ".gp_digest"]):
        gr = pg[str(n)].grad.double()
        assert abs(gr.pow(2).sum().sqrt().item() - row[2]) <= 1e-3 * row[2] + floor, n


def test_sampling_full_width_smac(golden):
    g = golden("g11_w128_smac_ant")
    A, Tt, Fd, hh = SHAPES["smac"]
    K, tag = 8, "g11.samp.smac"
    dcfg, c = den_cfg("smac", 128), dif_cfg("smac", K, hidden=256)
    p = full_params(dcfg, c)
    batch = O.synthetic_batch(2, Tt, A, Fd, c.action_dim, hh, False, 0, seed=19)
    shape = (2, Tt, A, Fd)
    x_init = 0.5 * T(O.closed_form_normal(f"{tag}.0", shape))
    z = [T(O.closed_form_normal(f"{tag}.{i}", shape)) for i in range(1, K)]
    with torch.no_grad():
        x, chain = O.conditional_sample(p, dcfg, c, O.DDPMSchedule(K), batch["cond"], batch["returns"], x_init, z, True)
    assert rel_err(chain[:, g["samp.smac.chain_idx"]], g["samp.smac.chain"]) < 1e-4
    assert rel_err(x, g["samp.smac.x"]) < 1e-4
    m = batch["cond"]["masks"].numpy()
    assert np.array_equal(x.numpy()[m], batch["cond"]["x"].numpy()[m])
