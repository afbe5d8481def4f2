"""Generate tests/golden/*.npz by running the REFERENCE's own model files (build container only).

TEST INFRASTRUCTURE.  Run as ``python oracle/gen_golden.py`` from the repo root in a container that
has /root/reference.  The reference package cannot be imported normally here (gym, ml_logger,
diffusers, ... are absent: ordinary ModuleNotFoundError), so its model files are loaded one by one
with importlib under small stub modules.  The only arithmetic that does NOT
come from the reference is the third-party diffusers==0.24.0 scheduler, stubbed with
oracle.madiff_oracle.DDPMSchedule (parity unpinned for that part).

Nothing from the reference is written to the repo except numeric inputs/outputs (.npz, no pickles).
Weights are the closed-form fill of oracle.madiff_oracle.closed_form_param, so fixtures only hold
inputs, recorded random draws, outputs and gradient digests.
"""
from __future__ import annotations

import contextlib
import importlib.util
import io
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import madiff_oracle as O  # noqa: E402

REF = "/root/reference/diffuser"
OUT = os.path.join(ROOT, "tests", "golden")


# ---------------------------------------------------------------- stubs + reference loading
class _SchedOut:
    def __init__(self, prev_sample):
        self.prev_sample = prev_sample


class _Cfg:
    pass


class StubDDPMScheduler(O.DDPMSchedule):
    def __init__(self, num_train_timesteps, clip_sample=True, prediction_type="epsilon",
                 beta_schedule="squaredcos_cap_v2"):
        assert clip_sample and prediction_type == "epsilon" and beta_schedule == "squaredcos_cap_v2"
        super().__init__(num_train_timesteps)
        self.config = _Cfg()
        self.config.num_train_timesteps = num_train_timesteps

    def step(self, model_output, t, x):
        z = torch.randn(x.shape) if int(t) > 0 else None   # recorded by the randn wrapper
        return _SchedOut(O.DDPMSchedule.step(self, model_output, int(t), x, z))


class StubDDIMScheduler(StubDDPMScheduler):
    def set_timesteps(self, n):
        self._ddim = O.DDIMSchedule(self.K, n)
        self.timesteps = self._ddim.timesteps

    def step(self, model_output, t, x):
        return _SchedOut(self._ddim.step(model_output, int(t), x))


def load_reference():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class Encoder:
        def __call__(self, d):
            raise NotImplementedError

    class IdentityEncoder(Encoder):
        def __call__(self, d):
            return d

    class Silent:
        def __init__(self, *a, **k):
            pass

        def update(self, *a, **k):
            pass

        def close(self):
            pass

    d = mod("diffuser")
    d.__path__ = [REF]
    u = mod("diffuser.utils", Encoder=Encoder, IdentityEncoder=IdentityEncoder, Progress=Silent, Silent=Silent,
            to_np=lambda x: x.detach().cpu().numpy(), to_torch=lambda x, **k: torch.as_tensor(x))
    d.utils = u
    m = mod("diffuser.models")
    m.__path__ = [REF + "/models"]
    mod("diffusers")
    mod("diffusers.schedulers")
    mod("diffusers.schedulers.scheduling_ddpm", DDPMScheduler=StubDDPMScheduler)
    mod("diffusers.schedulers.scheduling_ddim", DDIMScheduler=StubDDIMScheduler)
    out = {}
    for name in ("helpers", "temporal", "ma_temporal", "ma_temporal_wrappers", "diffusion"):
        spec = importlib.util.spec_from_file_location(f"diffuser.models.{name}", f"{REF}/models/{name}.py")
        mm = importlib.util.module_from_spec(spec)
        sys.modules[f"diffuser.models.{name}"] = mm
        spec.loader.exec_module(mm)
        out[name] = mm
    return out


class Recorder:
    """Record every random draw the reference makes (order preserved)."""

    def __init__(self, closed_form_randn: str = ""):
        self.draws = []
        self._orig = {}
        self._cf = closed_form_randn   # if set: torch.randn draw i is REPLACED by closed_form_normal(f"{tag}.{i}")
        self._n = 0

    def __enter__(self):
        rec = self

        def wrap(fn, tag):
            def inner(*a, **k):
                r = fn(*a, **k)
                rec.draws.append((tag, r.detach().clone()))
                return r
            return inner

        self._orig = {"randn": torch.randn, "randn_like": torch.randn_like, "randint": torch.randint,
                      "bern": torch.distributions.Bernoulli.sample}
        if self._cf:
            def cf_randn(*shape, **k):
                shape = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
                r = torch.from_numpy(O.closed_form_normal(f"{rec._cf}.{rec._n}", shape))
                rec._n += 1
                rec.draws.append(("randn", r.clone()))
                return r
            torch.randn = cf_randn
        else:
            torch.randn = wrap(torch.randn, "randn")
        torch.randn_like = wrap(torch.randn_like, "randn_like")
        torch.randint = wrap(torch.randint, "randint")
        torch.distributions.Bernoulli.sample = wrap(torch.distributions.Bernoulli.sample, "bern")
        return self

    def __exit__(self, *exc):
        torch.randn = self._orig["randn"]
        torch.randn_like = self._orig["randn_like"]
        torch.randint = self._orig["randint"]
        torch.distributions.Bernoulli.sample = self._orig["bern"]

    def of(self, tag):
        return [d for (t, d) in self.draws if t == tag]


def fill_closed_form(module, prefix=""):
    sd = module.state_dict()
    new = {k: O.closed_form_param(prefix + k, tuple(v.shape)) if v.dtype.is_floating_point and "loss_fn" not in k
           else v for k, v in sd.items()}
    module.load_state_dict(new)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def grad_digest(named_grads):
    """Per-parameter (sum, abs-sum, l2) so a 1M-parameter gradient fits in a few KB."""
    names, rows = [], []
    for k, g in named_grads:
        names.append(k)
        g = g.double()
        rows.append([g.sum().item(), g.abs().sum().item(), g.pow(2).sum().sqrt().item()])
    return np.array(names), np.array(rows, dtype=np.float64)


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    conv = {}
    for k, v in arrays.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        conv[k] = v
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def u(name, shape, lo=-1.0, hi=1.0):
    return torch.from_numpy(O.closed_form_uniform("input:" + name, shape, lo, hi).astype(np.float32))


# ---------------------------------------------------------------- G1: leaf modules
def gen_modules(R):
    H, T = R["helpers"], R["temporal"]
    out = {}

    def run(tag, module, inputs, prefix):
        fill_closed_form(module, prefix)
        ins = [i.clone().requires_grad_(i.dtype.is_floating_point) for i in inputs]
        y = module(*ins)
        gy = u(tag + ".gy", tuple(y.shape))
        y.backward(gy)
        out[tag + ".y"] = y
        out[tag + ".gy"] = gy
        for j, i in enumerate(ins):
            out[f"{tag}.in{j}"] = i.detach()
            if i.grad is not None:
                out[f"{tag}.gin{j}"] = i.grad
        for k, p in module.named_parameters():
            out[f"{tag}.gp.{k}"] = p.grad

    run("cblock", H.Conv1dBlock(12, 32, 5), [u("cblock.x", (3, 12, 8))], "cblock.")
    run("rtb_proj", T.ResidualTemporalBlock(12, 32, embed_dim=20), [u("rtb_proj.x", (3, 12, 8)), u("rtb_proj.t", (3, 20))], "rtb_proj.")
    run("rtb_same", T.ResidualTemporalBlock(32, 32, embed_dim=20), [u("rtb_same.x", (3, 32, 6)), u("rtb_same.t", (3, 20))], "rtb_same.")
    run("down", H.Downsample1d(16), [u("down.x", (3, 16, 14))], "down.")
    run("down_odd", H.Downsample1d(16), [u("down_odd.x", (3, 16, 7))], "down_odd.")
    run("up", H.Upsample1d(16), [u("up.x", (3, 16, 7))], "up.")
    att = T.TemporalSelfAttention(64, 4, 16, embed_dim=20, residual=True)
    run("attn", att, [u("attn.x", (2, 5, 64, 6)), u("attn.t", (2, 5, 20))], "attn.")
    emb = H.SinusoidalPosEmb(32)
    out["posemb.t"] = torch.tensor([[0, 1, 7], [100, 199, 42]])
    out["posemb.y"] = emb(out["posemb.t"])
    save("g1_modules", **out)


# ---------------------------------------------------------------- G2/G3: whole denoiser
SHAPES = {  # name: (A, T, F, hh)
    "spread": (3, 24, 18, 0), "tag": (3, 24, 16, 0), "ant": (4, 28, 54, 18), "smac": (8, 24, 88, 20),
    "tag4": (4, 24, 16, 0),     # BASELINE.json configs[1] as named ("4 agents"): the shape bench.py runs
}


def gen_denoiser(R, width, shapes, tag, with_grads):
    M = R["ma_temporal"]
    out = {}
    for name in shapes:
        A, T, Fd, hh = SHAPES[name]
        B = 2
        net = quiet(M.SharedConvAttentionDeconv, horizon=T, transition_dim=Fd, dim=width, history_horizon=hh,
                    dim_mults=(1, 4, 8), n_agents=A, returns_condition=True, condition_dropout=0.25)
        fill_closed_form(net, "model.")
        x = u(f"{tag}.{name}.x", (B, T, A, Fd))
        t = torch.tensor([3, 151])
        ret = u(f"{tag}.{name}.ret", (B, 1, A), 0.0, 1.0)
        with torch.no_grad():
            y_c = net(x, t, returns=ret, use_dropout=False)
            y_u = net(x, t, returns=ret, use_dropout=False, force_dropout=True)
        with Recorder() as rec:
            xg = x.clone().requires_grad_(True)
            y_d = net(xg, t, returns=ret, use_dropout=True)
        mask = rec.of("bern")[0]
        out.update({f"{name}.x": x, f"{name}.t": t, f"{name}.returns": ret, f"{name}.y_cond": y_c,
                    f"{name}.y_uncond": y_u, f"{name}.dropout_mask": mask, f"{name}.y_drop": y_d})
        if with_grads:
            gy = u(f"{tag}.{name}.gy", tuple(y_d.shape))
            y_d.backward(gy)
            names, rows = grad_digest([(k, p.grad) for k, p in net.named_parameters() if p.grad is not None])
            out[f"{name}.gy"] = gy
            out[f"{name}.gx"] = xg.grad
            out[f"{name}.gp_names"] = names
            out[f"{name}.gp_digest"] = rows
            out[f"{name}.nograd_names"] = np.array([k for k, p in net.named_parameters() if p.grad is None])
            for k in ("net.final_conv.1.bias", "self_attn.0.gamma", "self_attn.2.gamma",
                      "net.mid_block1.blocks.0.block.2.weight", "net.downs.0.0.time_mlp.1.bias",
                      "net.returns_mlp.0.weight"):
                out[f"{name}.gp.{k}"] = dict(net.named_parameters())[k].grad
    save(tag, **out)


# ---------------------------------------------------------------- G4/G6: diffusion wrapper
def build_diffusion(R, name, width, K, discrete=False, share_inv=True):
    M, D = R["ma_temporal"], R["diffusion"]
    A, T, Fd, hh = SHAPES[name]
    act = 1 if discrete else {"spread": 5, "tag": 5, "tag4": 5, "ant": 2, "smac": 1}[name]
    nact = 14 if discrete else 0
    net = quiet(M.SharedConvAttentionDeconv, horizon=T, transition_dim=Fd, dim=width, history_horizon=hh,
                dim_mults=(1, 4, 8), n_agents=A, returns_condition=True, condition_dropout=0.25)
    dif = quiet(D.GaussianDiffusion, net, n_agents=A, horizon=T - hh, history_horizon=hh, observation_dim=Fd,
                action_dim=act, discrete_action=discrete, num_actions=nact, n_timesteps=K,
                returns_condition=True, condition_guidance_w=1.2, hidden_dim=64 if width < 128 else 256,
                share_inv=share_inv, clip_denoised=True, predict_epsilon=True, loss_discount=1.0)
    fill_closed_form(dif)
    return dif, (A, T, Fd, hh, act, nact)


def gen_loss(R):
    out = {}
    for tag, name, discrete, share in (("spread", "spread", False, True), ("ant", "ant", False, True),
                                       ("smac", "smac", True, True), ("spread_indep", "spread", False, False)):
        dif, (A, T, Fd, hh, act, nact) = build_diffusion(R, name, 16, 200, discrete, share)
        vis = np.ones(A, dtype=bool)
        if hh > 0:
            vis[-1] = False
        batch = O.synthetic_batch(3, T, A, Fd, act, hh, discrete, nact, seed=11, visible=vis)
        with Recorder() as rec:
            loss, info = dif.loss(**batch)
            loss.backward()
        out[f"{tag}.t"] = rec.of("randint")[0]
        out[f"{tag}.noise"] = rec.of("randn_like")[0]
        out[f"{tag}.dropout_mask"] = rec.of("bern")[0]
        out[f"{tag}.loss"] = loss.detach()
        for k, v in info.items():
            out[f"{tag}.info.{k}"] = torch.as_tensor(v).detach()
        out[f"{tag}.visible"] = vis
        out[f"{tag}.loss_weights"] = dif.loss_fn.weights
        names, rows = grad_digest([(k, p.grad) for k, p in dif.named_parameters() if p.grad is not None])
        out[f"{tag}.gp_names"] = names
        out[f"{tag}.gp_digest"] = rows
        for k in ("inv_model.4.bias", "model.net.final_conv.1.bias", "inv_model.0.4.bias"):
            pd = dict(dif.named_parameters())
            if k in pd:
                out[f"{tag}.gp.{k}"] = pd[k].grad
    save("g4_loss", **out)


def gen_sample(R):
    out = {}
    for tag, name, width, K, B in (("small_k200", "spread", 16, 200, 2), ("smac_k20", "smac", 16, 20, 2),
                                   ("full_k8", "spread", 128, 8, 2)):
        dif, (A, T, Fd, hh, act, nact) = build_diffusion(R, name, width, K)
        vis = np.ones(A, dtype=bool)
        batch = O.synthetic_batch(B, T, A, Fd, act, hh, False, 0, seed=5, visible=vis)
        cond, returns = batch["cond"], batch["returns"]
        # draw 0 = initial x_K (tempered by 0.5, diffusion.py:239), draws 1..K-1 = posterior z (none at t == 0);
        # they are closed_form_normal(f"{tag}.{i}", shape) so the fixture need not store them.
        with Recorder(closed_form_randn=tag) as rec:
            x, chain = dif.conditional_sample(cond, returns=returns, verbose=False, return_diffusion=True)
        assert len(rec.of("randn")) == K
        out[f"{tag}.x"] = x
        sel = [0, 1, K // 2, K - 1, K]
        out[f"{tag}.chain_idx"] = np.array(sel)
        out[f"{tag}.chain"] = chain[:, sel]
        out[f"{tag}.cond_x"] = cond["x"]
        out[f"{tag}.cond_masks"] = cond["masks"]
        out[f"{tag}.returns"] = returns
        if tag == "small_k200":
            dif.set_ddim_scheduler(15)
            with Recorder(closed_form_randn="ddim15") as rec:
                xd = dif.conditional_sample(cond, returns=returns, verbose=False)
            assert len(rec.of("randn")) == 1
            out["ddim15.x"] = xd
    save("g6_sample", **out)


def gen_masks():
    """g7: conditioning masks of the reference's own MultiAgentMaskGenerator (utils/mask_generator.py:11-59, a
    numpy/einops-only file, loaded directly) for the shapes and agent-visibility patterns the evaluator builds
    (evaluator.py:56-72 decentralized: one visible agent; :117-120 centralized: all visible)."""
    spec = importlib.util.spec_from_file_location("ref_mask_generator", REF + "/utils/mask_generator.py")
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    out = {}
    for name, (A, T, F, hh) in {"spread": (3, 24, 18, 0), "ant": (4, 28, 54, 18), "smac": (8, 24, 88, 20)}.items():
        gen = mg.MultiAgentMaskGenerator(action_dim=5, observation_dim=F, history_horizon=hh, action_visible=False)
        out[f"{name}.all"] = gen((2, T, A, F), np.ones(A))
        for a_idx in (0, A - 1):
            am = np.zeros(A)
            am[a_idx] = 1.0
            out[f"{name}.only{a_idx}"] = gen((2, T, A, F), am)
    save("g7_masks", **{k: np.packbits(v.astype(np.uint8), axis=None) for k, v in out.items()},
         **{k + ".shape": np.asarray(v.shape) for k, v in out.items()})


# ---------------------------------------------------------------- round 2: full width WITH gradients, env_ts
N_SAMPLES = 96


def sample_index(numel: int) -> np.ndarray:
    """Closed-form pseudo-random positions inside a flattened tensor (tests/test_gpu_parity.py recomputes them)."""
    i = np.arange(N_SAMPLES, dtype=np.uint64)
    return ((i * np.uint64(2654435761) + np.uint64(40503)) % np.uint64(numel)).astype(np.int64)


def grad_samples(named_grads):
    """[n_params, N_SAMPLES] gradient elements at sample_index positions: pins individual elements of the big weight
    gradients (a digest alone would accept a permuted or mis-tiled gradient with the right norm)."""
    rows = []
    for _k, g in named_grads:
        flat = g.detach().reshape(-1)
        rows.append(flat[torch.from_numpy(sample_index(flat.numel()))].float().numpy())
    return np.stack(rows)


T_DRAWS = [3, 151, 0, 199, 57, 100, 12, 180]


def gen_denoiser_grads_w128(R):
    """g3g: dim-128 denoiser forward + backward (what the bf16 fused conv+GroupNorm kernel, the all-taps data/weight
    gradient kernels and the GEMM-layout gradient sink compute) for spread (A=3) and the benchmarked tag shape (A=4)."""
    M = R["ma_temporal"]
    out = {}
    for name, B in (("spread", 6), ("tag4", 8)):
        A, T, Fd, hh = SHAPES[name]
        net = quiet(M.SharedConvAttentionDeconv, horizon=T, transition_dim=Fd, dim=128, history_horizon=hh,
                    dim_mults=(1, 4, 8), n_agents=A, returns_condition=True, condition_dropout=0.25)
        fill_closed_form(net, "model.")
        x = u(f"g3g.{name}.x", (B, T, A, Fd))
        t = torch.tensor(T_DRAWS[:B])
        ret = u(f"g3g.{name}.ret", (B, 1, A), 0.0, 1.0)
        with Recorder() as rec:
            xg = x.clone().requires_grad_(True)
            y = net(xg, t, returns=ret, use_dropout=True)
        gy = u(f"g3g.{name}.gy", tuple(y.shape))
        y.backward(gy)
        named = [(k, p.grad) for k, p in net.named_parameters() if p.grad is not None]
        names, rows = grad_digest(named)
        out.update({f"{name}.x": x, f"{name}.t": t, f"{name}.returns": ret, f"{name}.dropout_mask": rec.of("bern")[0],
                    f"{name}.y_drop": y, f"{name}.gy": gy, f"{name}.gx": xg.grad, f"{name}.gp_names": names,
                    f"{name}.gp_digest": rows, f"{name}.gp_samples": grad_samples(named),
                    f"{name}.nograd_names": np.array([k for k, p in net.named_parameters() if p.grad is None])})
        for k in ("net.final_conv.1.bias", "net.final_conv.0.block.0.bias", "self_attn.0.gamma", "self_attn.1.gamma",
                  "self_attn.2.gamma", "net.mid_block1.blocks.0.block.2.weight", "net.mid_block1.blocks.0.block.2.bias",
                  "net.downs.2.1.time_mlp.1.bias", "net.ups.0.0.blocks.0.block.0.bias", "net.returns_mlp.0.weight"):
            out[f"{name}.gp.{k}"] = dict(net.named_parameters())[k].grad
    save("g3g_denoiser_w128_grads", **out)


def gen_loss_w128(R):
    """g4w: GaussianDiffusion.loss + backward at dim 128 / hidden 256 on the benchmarked tag shape (A=4)."""
    out = {}
    tag, name, B = "tag4", "tag4", 4
    dif, (A, T, Fd, hh, act, nact) = build_diffusion(R, name, 128, 200)
    batch = O.synthetic_batch(B, T, A, Fd, act, hh, False, 0, seed=13)
    with Recorder() as rec:
        loss, info = dif.loss(**batch)
        loss.backward()
    named = [(k, p.grad) for k, p in dif.named_parameters() if p.grad is not None]
    names, rows = grad_digest(named)
    out.update({f"{tag}.t": rec.of("randint")[0], f"{tag}.noise": rec.of("randn_like")[0],
                f"{tag}.dropout_mask": rec.of("bern")[0], f"{tag}.loss": loss.detach(), f"{tag}.gp_names": names,
                f"{tag}.gp_digest": rows, f"{tag}.gp_samples": grad_samples(named)})
    for k, v in info.items():
        out[f"{tag}.info.{k}"] = torch.as_tensor(v).detach()
    save("g4w_loss_w128", **out)


def gen_env_ts(R):
    """g8: env_ts_condition=True (ma_temporal.py:355-362): an nn.Embedding(max_path_length + 1) lookup of
    env_timestep[:, history_horizon] -> MLP, concatenated to the embedding.  Small width, forward + gradients."""
    M = R["ma_temporal"]
    out = {}
    name, B, mpl = "ant", 3, 40
    A, T, Fd, hh = SHAPES[name]
    net = quiet(M.SharedConvAttentionDeconv, horizon=T, transition_dim=Fd, dim=16, history_horizon=hh,
                dim_mults=(1, 4, 8), n_agents=A, returns_condition=True, env_ts_condition=True, condition_dropout=0.25,
                max_path_length=mpl)
    fill_closed_form(net, "model.")
    x = u("g8.x", (B, T, A, Fd))
    t = torch.tensor(T_DRAWS[:B])
    ret = u("g8.ret", (B, 1, A), 0.0, 1.0)
    start = torch.tensor([0, 7, mpl - hh])                 # env step of the window start; the last one reaches the pad id
    env_ts = (start[:, None] + torch.arange(T)[None, :] - hh)
    env_ts[(env_ts < 0) | (env_ts >= mpl)] = mpl           # datasets/sequence.py / evaluator.py:42-44 convention
    with torch.no_grad():
        y_c = net(x, t, returns=ret, env_timestep=env_ts, use_dropout=False)
        y_u = net(x, t, returns=ret, env_timestep=env_ts, use_dropout=False, force_dropout=True)
    with Recorder() as rec:
        xg = x.clone().requires_grad_(True)
        y = net(xg, t, returns=ret, env_timestep=env_ts, use_dropout=True)
    gy = u("g8.gy", tuple(y.shape))
    y.backward(gy)
    named = [(k, p.grad) for k, p in net.named_parameters() if p.grad is not None]
    names, rows = grad_digest(named)
    out.update({"x": x, "t": t, "returns": ret, "env_ts": env_ts, "max_path_length": np.array(mpl), "y_cond": y_c,
                "y_uncond": y_u, "dropout_mask": rec.of("bern")[0], "y_drop": y, "gy": gy, "gx": xg.grad,
                "gp_names": names, "gp_digest": rows})
    for k, p in net.named_parameters():
        if k.startswith("net.env_ts_mlp"):
            out["gp." + k] = p.grad
    save("g8_env_ts", **out)


def gen_alternates(R):
    """g9: the non-temporal SelfAttention (helpers.py:77-124), TemporalUnet.forward
    (temporal.py:348-406), the three wrapper denoisers (ma_temporal_wrappers.py) and the SMAC data encoder
    (utils/data_encoder.py:11-20) inside GaussianDiffusion -- all from the reference's own files, small width."""
    H, T, W, D = R["helpers"], R["temporal"], R["ma_temporal_wrappers"], R["diffusion"]
    out = {}
    # --- SelfAttention as a module
    att = H.SelfAttention(64, 4, 16, residual=True)
    fill_closed_form(att, "selfattn.")
    x = u("g9.selfattn.x", (2, 5, 64, 6)).requires_grad_(True)
    y = att(x)
    gy = u("g9.selfattn.gy", tuple(y.shape))
    y.backward(gy)
    out.update({"selfattn.x": x.detach(), "selfattn.y": y, "selfattn.gy": gy, "selfattn.gx": x.grad})
    for k, p in att.named_parameters():
        out["selfattn.gp." + k] = p.grad
    # --- U-Net alone and the wrappers
    B, Tn = 3, 24

    def run(tag, net, xshape, rshape, n_masks):
        fill_closed_form(net, "model.")
        x = u(f"g9.{tag}.x", xshape)
        t = torch.tensor(T_DRAWS[:B])
        ret = u(f"g9.{tag}.ret", rshape, 0.0, 1.0)
        with torch.no_grad():
            y_c = net(x, t, returns=ret, use_dropout=False)
            y_u = net(x, t, returns=ret, use_dropout=False, force_dropout=True)
        with Recorder() as rec:
            xg = x.clone().requires_grad_(True)
            y = net(xg, t, returns=ret, use_dropout=True)
        masks = rec.of("bern")
        assert len(masks) == n_masks, (tag, len(masks))
        gy = u(f"g9.{tag}.gy", tuple(y.shape))
        y.backward(gy)
        names, rows = grad_digest([(k, p.grad) for k, p in net.named_parameters() if p.grad is not None])
        out.update({f"{tag}.x": x, f"{tag}.t": t, f"{tag}.returns": ret, f"{tag}.y_cond": y_c, f"{tag}.y_uncond": y_u,
                    f"{tag}.dropout_masks": torch.stack(masks), f"{tag}.y_drop": y, f"{tag}.gy": gy, f"{tag}.gx": xg.grad,
                    f"{tag}.gp_names": names, f"{tag}.gp_digest": rows})

    kw = dict(dim=16, dim_mults=(1, 4, 8), returns_condition=True, condition_dropout=0.25)
    run("unet", quiet(T.TemporalUnet, horizon=Tn, transition_dim=18, **kw), (B, Tn, 18), (B, 1), 1)
    run("concat", quiet(W.ConcatenatedTemporalUnet, n_agents=3, horizon=Tn, transition_dim=6, **kw), (B, Tn, 3, 6), (B, 1, 3), 1)
    run("indep", quiet(W.IndependentTemporalUnet, n_agents=2, horizon=Tn, transition_dim=10, **kw), (B, Tn, 2, 10), (B, 1, 2), 2)
    run("shared_indep", quiet(W.SharedIndependentTemporalUnet, n_agents=3, horizon=Tn, history_horizon=0, transition_dim=10,
                              **kw), (B, Tn, 3, 10), (B, 1, 3), 1)
    # --- SMAC data encoder inside GaussianDiffusion: loss and a K = 8 sample
    spec = importlib.util.spec_from_file_location("ref_data_encoder", REF + "/utils/data_encoder.py")
    de = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(de)
    M = R["ma_temporal"]
    A, Tt, Fd, hh, act, K = 3, 24, 18, 0, 5, 8
    net = quiet(M.SharedConvAttentionDeconv, horizon=Tt, transition_dim=Fd, dim=16, history_horizon=hh, dim_mults=(1, 4, 8),
                n_agents=A, returns_condition=True, condition_dropout=0.25)
    dif = quiet(D.GaussianDiffusion, net, n_agents=A, horizon=Tt, history_horizon=hh, observation_dim=Fd, action_dim=act,
                n_timesteps=K, returns_condition=True, condition_guidance_w=1.2, hidden_dim=64,
                data_encoder=de.SMAC3mEncoder())
    fill_closed_form(dif)
    batch = O.synthetic_batch(3, Tt, A, Fd, act, hh, False, 0, seed=21)
    with Recorder() as rec:
        loss, info = dif.loss(**batch)
        loss.backward()
    names, rows = grad_digest([(k, p.grad) for k, p in dif.named_parameters() if p.grad is not None])
    out.update({"enc.t": rec.of("randint")[0], "enc.noise": rec.of("randn_like")[0], "enc.dropout_mask": rec.of("bern")[0],
                "enc.loss": loss.detach(), "enc.a0_loss": info["a0_loss"].detach(), "enc.gp_names": names,
                "enc.gp_digest": rows})
    sb = O.synthetic_batch(2, Tt, A, Fd, act, hh, False, 0, seed=22)
    with Recorder(closed_form_randn="g9enc") as rec:
        xs = dif.conditional_sample(sb["cond"], returns=sb["returns"], verbose=False)
    assert len(rec.of("randn")) == K
    out.update({"enc.sample": xs, "enc.cond_x": sb["cond"]["x"], "enc.cond_masks": sb["cond"]["masks"],
                "enc.returns": sb["returns"]})
    save("g9_alternates", **out)


def gen_planner():
    """g10: the evaluator's planning step, `MADEvaluatorWorker._generate_samples`
    (utils/evaluator.py:39-135), called as a plain function on a stand-in `self` -- the reference's own mask generator
    (utils/mask_generator.py) and array helpers (utils/arrays.py), a Config with the evaluator's fields and a recording
    stand-in for `trainer.ema_model` whose conditional_sample returns a closed-form tensor.  Pins what the sampler is CALLED
    with (conditioning trajectories and masks, the (b a) expansion of returns / env_ts, the env_ts clamp) and the gather
    of the decentralized plans; the sampler itself is pinned elsewhere (g6).  ml_logger and launcher_util (params_proto)
    are absent third-party / control-plane imports of evaluator.py and are stubbed; the environment loop is not run."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m

    mod("ml_logger", logger=types.SimpleNamespace(print=print))
    d = mod("diffuser")
    d.__path__ = [REF]
    u = mod("diffuser.utils")
    u.__path__ = [REF + "/utils"]
    d.utils = u
    load("diffuser.utils.arrays", REF + "/utils/arrays.py")
    mod("diffuser.utils.launcher_util", build_config_from_dict=lambda specs, Config=None: types.SimpleNamespace(**specs))
    mg = load("diffuser.utils.mask_generator", REF + "/utils/mask_generator.py")
    ev = load("diffuser.utils.evaluator", REF + "/utils/evaluator.py")
    out = {}
    for tag, (A, horizon, hh, F, B, dec) in {"cen": (3, 6, 2, 4, 2, False), "dec": (3, 6, 2, 4, 2, True),
                                             "dec_nohist": (2, 8, 0, 5, 3, True)}.items():
        calls = []

        class FakeEMA:
            def conditional_sample(self, conditions, returns=None, env_ts=None, attention_masks=None):
                calls.append({"x": conditions["x"].clone(), "masks": conditions["masks"].clone(), "returns": returns.clone(),
                              "env_ts": env_ts.clone(), "attention_masks": attention_masks.clone()})
                return torch.from_numpy(np.asarray(O.closed_form_normal(f"g10.{tag}.plan", tuple(conditions["x"].shape)),
                                                   dtype=np.float32))

        cfg = types.SimpleNamespace(max_path_length=25, horizon=horizon, history_horizon=hh, n_agents=A,
                                    decentralized_execution=dec, device="cpu")
        fake = types.SimpleNamespace(Config=cfg, trainer=types.SimpleNamespace(ema_model=FakeEMA()),
                                     mask_generator=mg.MultiAgentMaskGenerator(action_dim=2, observation_dim=F,
                                                                               history_horizon=hh, action_visible=False))
        obs = np.asarray(O.closed_form_normal(f"g10.{tag}.obs", (B, hh + 1, A, F)), dtype=np.float32)
        returns = 0.9 * torch.ones(B, 1, A)
        env_ts = (torch.arange(horizon + hh) - hh)[None].repeat(B, 1) + torch.tensor([0, 21, 3][:B])[:, None]
        samples = ev.MADEvaluatorWorker._generate_samples(fake, obs, returns, env_ts)
        assert len(calls) == 1
        c = calls[0]
        out.update({f"{tag}.obs": obs, f"{tag}.returns": returns.numpy(), f"{tag}.env_ts": env_ts.numpy(),
                    f"{tag}.cond_x": c["x"].numpy(), f"{tag}.cond_masks": c["masks"].numpy().astype(np.uint8),
                    f"{tag}.call_returns": c["returns"].numpy(), f"{tag}.call_env_ts": c["env_ts"].numpy(),
                    f"{tag}.samples": samples.numpy(), f"{tag}.geom": np.asarray([A, horizon, hh, F, B, int(dec)])})
    save("g10_planner", **out)


# ---------------------------------------------------------------- round 3: full width on the SMAC and ant shapes
def gen_round3(R):
    """g11: the shapes BASELINE.json configs[2] / configs[3] name, at the width the benchmarks run (dim 128), from the
    reference's own classes: `smac` (A=8, T=24, F=88, history 20; d_qk=64 / d_v=256 attention over 8 agents) and `ant`
    (A=4, T=28, F=54, history 18; levels 28/14/7).  Three parts per the g3g / g4w / g6 fixtures:
      den.<name>  denoiser forward with dropout + backward: y, input gradient, per-parameter digest + 96 sampled elements
      loss.smac   GaussianDiffusion.loss + backward, discrete actions (legal-action mask, inv_acc), hidden 256
      loss.ant    ... continuous actions
      samp.smac   K=8 guided conditional_sample with closed-form draws
    Inputs x / gy / returns are closed-form (tests recompute them); recorded draws and outputs are stored."""
    M = R["ma_temporal"]
    out = {}
    B = 2
    for name in ("smac", "ant"):
        A, T, Fd, hh = SHAPES[name]
        net = quiet(M.SharedConvAttentionDeconv, horizon=T, transition_dim=Fd, dim=128, history_horizon=hh,
                    dim_mults=(1, 4, 8), n_agents=A, returns_condition=True, condition_dropout=0.25)
        fill_closed_form(net, "model.")
        x = u(f"g11.{name}.x", (B, T, A, Fd))
        t = torch.tensor(T_DRAWS[:B])
        ret = u(f"g11.{name}.ret", (B, 1, A), 0.0, 1.0)
        with Recorder() as rec:
            xg = x.clone().requires_grad_(True)
            y = net(xg, t, returns=ret, use_dropout=True)
        gy = u(f"g11.{name}.gy", tuple(y.shape))
        y.backward(gy)
        named = [(k, p.grad) for k, p in net.named_parameters() if p.grad is not None]
        names, rows = grad_digest(named)
        pre = f"den.{name}"
        out.update({f"{pre}.t": t, f"{pre}.dropout_mask": rec.of("bern")[0], f"{pre}.y_drop": y, f"{pre}.gx": xg.grad,
                    f"{pre}.gp_names": names, f"{pre}.gp_digest": rows, f"{pre}.gp_samples": grad_samples(named),
                    f"{pre}.nograd_names": np.array([k for k, p in net.named_parameters() if p.grad is None])})
        for k in ("self_attn.0.gamma", "self_attn.1.gamma", "self_attn.2.gamma", "net.final_conv.1.bias",
                  "net.downs.2.1.time_mlp.1.bias", "self_attn.0.value_time_mlp.1.bias"):
            out[f"{pre}.gp.{k}"] = dict(net.named_parameters())[k].grad
        del net
    for name, discrete in (("smac", True), ("ant", False)):
        dif, (A, T, Fd, hh, act, nact) = build_diffusion(R, name, 128, 200, discrete)
        vis = np.ones(A, dtype=bool)
        vis[-1] = False
        batch = O.synthetic_batch(B, T, A, Fd, act, hh, discrete, nact, seed=17, visible=vis)
        with Recorder() as rec:
            loss, info = dif.loss(**batch)
            loss.backward()
        named = [(k, p.grad) for k, p in dif.named_parameters() if p.grad is not None]
        names, rows = grad_digest(named)
        pre = f"loss.{name}"
        out.update({f"{pre}.t": rec.of("randint")[0], f"{pre}.noise": rec.of("randn_like")[0],
                    f"{pre}.dropout_mask": rec.of("bern")[0], f"{pre}.loss": loss.detach(), f"{pre}.visible": vis,
                    f"{pre}.gp_names": names, f"{pre}.gp_digest": rows, f"{pre}.gp_samples": grad_samples(named)})
        for k, v in info.items():
            out[f"{pre}.info.{k}"] = torch.as_tensor(v).detach()
        del dif
    name, K, tag = "smac", 8, "g11.samp.smac"
    dif, (A, T, Fd, hh, act, nact) = build_diffusion(R, name, 128, K)
    batch = O.synthetic_batch(B, T, A, Fd, act, hh, False, 0, seed=19)
    cond, returns = batch["cond"], batch["returns"]
    with Recorder(closed_form_randn=tag) as rec:
        x, chain = dif.conditional_sample(cond, returns=returns, verbose=False, return_diffusion=True)
    assert len(rec.of("randn")) == K
    sel = [1, K // 2]
    out.update({"samp.smac.x": x, "samp.smac.chain_idx": np.array(sel), "samp.smac.chain": chain[:, sel]})
    save("g11_w128_smac_ant", **out)



def main():
    if len(sys.argv) > 1 and sys.argv[1] == "planner":
        gen_planner()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "alternates":
        torch.manual_seed(1234)
        torch.set_num_threads(8)
        gen_alternates(load_reference())
        return
    if len(sys.argv) > 1 and sys.argv[1] == "masks":
        gen_masks()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "round3":      # only the fixture added in round 3
        torch.manual_seed(1234)
        torch.set_num_threads(8)
        gen_round3(load_reference())
        return
    if len(sys.argv) > 1 and sys.argv[1] == "round2":      # only the fixtures added in round 2
        torch.manual_seed(1234)
        torch.set_num_threads(8)
        R = load_reference()
        gen_denoiser_grads_w128(R)
        gen_loss_w128(R)
        gen_env_ts(R)
        return
    torch.manual_seed(1234)
    torch.set_num_threads(8)
    R = load_reference()
    gen_masks()
    gen_modules(R)
    gen_denoiser(R, 16, ["spread", "tag", "ant", "smac"], "g2_denoiser_w16", True)
    gen_denoiser(R, 128, ["spread"], "g3_denoiser_w128", False)
    gen_loss(R)
    gen_sample(R)
    gen_denoiser_grads_w128(R)
    gen_loss_w128(R)
    gen_env_ts(R)
    gen_alternates(R)
    gen_round3(R)


if __name__ == "__main__":
    main()
