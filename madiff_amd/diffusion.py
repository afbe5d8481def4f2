"""GaussianDiffusion on the HIP kernels: same constructor, attributes, ``state_dict`` keys and call surface as
/root/reference/diffuser/models/diffusion.py:13-499 (``loss(**batch)``, ``conditional_sample``, ``forward``,
``set_ddim_scheduler``, ``inv_model``); ``p_sample_loop`` is an alias of ``conditional_sample``.

Differences that are additions, not changes of behaviour:
* every random draw of the reference may be supplied explicitly (``t``, ``noise``, ``dropout_mask`` to
  ``loss``; ``x_init``/``step_noise`` to ``conditional_sample``) so results can be compared bit-for-bit in
  draw order with the reference; without them torch's RNG is used in the reference's draw order;
* the whole K-step loop is captured into ONE hipGraph (``use_graph=True``, the default on a GPU): K x [inpaint ->
  one 2B-batch guided forward -> fused posterior step], driven by a device-side step counter, the scheduler's
  noise drawn on the device (Philox) or read from a host-supplied table -- one host launch per call;
* the classifier-free-guidance pair runs as one 2B batch.
"""
from __future__ import annotations

import os
import weakref

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .modules import SharedConvAttentionDeconv, rows_to_cl
from .ops import LINEAR, MaskedStateLoss, conv_gemm, pad32
from .schedulers import DDIMScheduler, DDPMScheduler


class Encoder(object):
    def __call__(self, data):
        raise NotImplementedError


class IdentityEncoder(Encoder):
    """diffuser/utils/data_encoder.py:6-8."""

    def __call__(self, data):
        return data


class _FirstKEncoder(Encoder):
    """utils/data_encoder.py:11-20: the first k observation features (the agent's own unit features in SMAC) of every
    later time step are overwritten by those of time step 0.  Callable on any [..., T, A, F] tensor like the reference's
    (in place); GaussianDiffusion applies it with one HIP launch (madiff_encode_first_k)."""

    first_k = 0

    def __call__(self, data):
        k = self.first_k
        data[..., 1:, :, :k] = data[..., 0:1, :, :k]
        return data


class SMAC5m6mEncoder(_FirstKEncoder):
    first_k = 5


class SMAC3mEncoder(_FirstKEncoder):
    first_k = 3


class InvDynMLP(nn.Sequential):
    """Linear-ReLU-Linear-ReLU-Linear[-Softmax] inverse-dynamics head (diffusion.py:100-137) whose three
    contractions run on the MFMA GEMM kernel (ReLU fused in the epilogue).  Children keep the reference's
    nn.Sequential indices 0, 2, 4 so checkpoints load unchanged; callable on [..., in_dim] like the original."""

    def __init__(self, in_dim, hidden_dim, out_dim, softmax=False):
        mods = [nn.Linear(in_dim, hidden_dim), nn.ReLU(), nn.Linear(hidden_dim, hidden_dim), nn.ReLU(),
                nn.Linear(hidden_dim, out_dim)]
        if softmax:
            mods.append(nn.Softmax(dim=-1))
        else:
            mods.append(nn.Identity())
        super().__init__(*mods)
        self.out_dim = out_dim
        self.softmax = softmax

    def logits_cl(self, h):
        """cl rows [R, pad32(in_dim)] -> fp32 [R, pad32(out_dim)] (no softmax): the fused loss reads this directly."""
        h = conv_gemm(h, [self[0].weight], [self[0].bias], LINEAR, 1, act=1)
        h = conv_gemm(h, [self[2].weight], [self[2].bias], LINEAR, 1, act=1)
        return conv_gemm(h, [self[4].weight], [self[4].bias], LINEAR, 1, perm="f32")

    def forward(self, x):
        lead = x.shape[:-1]
        h = rows_to_cl(x.reshape(-1, x.shape[-1]).to(torch.float32))
        h = conv_gemm(h, [self[0].weight], [self[0].bias], LINEAR, 1, act=1)
        h = conv_gemm(h, [self[2].weight], [self[2].bias], LINEAR, 1, act=1)
        h = conv_gemm(h, [self[4].weight], [self[4].bias], LINEAR, 1, perm="f32")[:, : self.out_dim]
        if self.softmax:
            h = torch.softmax(h, dim=-1)
        return h.reshape(*lead, self.out_dim)


INV_SIDE_STREAM = os.environ.get("MADIFF_INV_SIDE_STREAM", "1") != "0"
FUSED_LOSS = os.environ.get("MADIFF_FUSED_LOSS", "1") != "0"
# GaussianDiffusion -> its captured denoising step (kept outside the module: deepcopy / pickling of the module stay plain)
_SAMPLE_CACHE: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


class WeightedStateL2(nn.Module):
    """helpers.py:238-251,292-294 (the ``state_l2`` loss): holds the ``weights`` buffer [T, A, F]."""

    def __init__(self, weights):
        super().__init__()
        self.register_buffer("weights", weights)

    def forward(self, pred, targ):
        loss = F.mse_loss(pred, targ, reduction="none") * self.weights
        return loss, {"a0_loss": loss.mean()}


class GaussianDiffusion(nn.Module):
    def __init__(self, model, n_agents: int, horizon: int, history_horizon: int, observation_dim: int,
                 action_dim: int, use_inv_dyn: bool = True, discrete_action: bool = False, num_actions: int = 0,
                 n_timesteps: int = 1000, clip_denoised: bool = False, predict_epsilon: bool = True,
                 action_weight: float = 1.0, hidden_dim: int = 256, loss_discount: float = 1.0,
                 loss_weights: np.ndarray = None, state_loss_weight: float = None,
                 opponent_loss_weight: float = None, returns_condition: bool = False,
                 condition_guidance_w: float = 1.2, returns_loss_guided: bool = False, loss_guidence_w: float = 0.1,
                 value_diffusion_model: nn.Module = None, train_only_inv: bool = False, share_inv: bool = True,
                 joint_inv: bool = False, data_encoder: Encoder = IdentityEncoder(), **kwargs):
        assert action_dim > 0
        assert not returns_condition or not returns_loss_guided, \
            "Can't do both returns conditioning and returns loss guidence"
        if not use_inv_dyn or not predict_epsilon or returns_loss_guided or value_diffusion_model is not None:
            raise NotImplementedError("only the configuration every exp_spec uses is on the hot path: "
                                      "use_inv_dyn=True, predict_epsilon=True, no value guidance (SURVEY.md section 2)")
        if opponent_loss_weight is not None:
            # diffusion.py:315-325 re-weights the loss of the opponents of cond["agent_idx"]; no exp_spec sets it, and
            # silently ignoring it would train a different objective.  (state_loss_weight is stored and never read by the
            # reference either, diffusion.py:149-172.)
            raise NotImplementedError("opponent_loss_weight is not on the hot path (no exp_spec sets it)")
        super().__init__()
        self.n_agents = n_agents
        self.horizon = horizon
        self.history_horizon = history_horizon
        self.observation_dim = observation_dim
        self.action_dim = action_dim
        self.state_loss_weight = state_loss_weight
        self.opponent_loss_weight = opponent_loss_weight
        self.discrete_action = discrete_action
        self.num_actions = num_actions
        self.transition_dim = observation_dim + action_dim
        self.model = model
        self.use_inv_dyn = use_inv_dyn
        self.train_only_inv = train_only_inv
        self.share_inv = share_inv
        self.joint_inv = joint_inv
        self.data_encoder = data_encoder
        self.inv_model = self._build_inv_model(hidden_dim, output_dim=action_dim if not discrete_action else num_actions)
        self.returns_condition = returns_condition
        self.condition_guidance_w = condition_guidance_w
        self.returns_loss_guided = returns_loss_guided
        self.loss_guidence_w = loss_guidence_w
        self.value_diffusion_model = None
        self.n_timesteps = int(n_timesteps)
        self.clip_denoised = clip_denoised
        self.predict_epsilon = predict_epsilon
        self.noise_scheduler = DDPMScheduler(num_train_timesteps=self.n_timesteps, clip_sample=True,
                                             prediction_type="epsilon", beta_schedule="squaredcos_cap_v2")
        self.use_ddim_sample = False
        self.loss_fn = WeightedStateL2(self.get_loss_weights(loss_discount, action_weight))
        self._tables: dict = {}
        self._graphs: dict = {}

    # ------------------------------------------------------------------ construction helpers
    def _build_inv_model(self, hidden_dim: int, output_dim: int):
        if self.joint_inv:
            return InvDynMLP(self.n_agents * 2 * self.observation_dim, hidden_dim, self.n_agents * output_dim)
        if self.share_inv:
            return InvDynMLP(2 * self.observation_dim, hidden_dim, output_dim)
        return nn.ModuleList([InvDynMLP(2 * self.observation_dim, hidden_dim, output_dim, softmax=self.discrete_action)
                              for _ in range(self.n_agents)])

    def set_ddim_scheduler(self, n_ddim_steps: int = 15):
        self.ddim_noise_scheduler = DDIMScheduler(num_train_timesteps=self.n_timesteps, clip_sample=True,
                                                  prediction_type="epsilon", beta_schedule="squaredcos_cap_v2")
        self.ddim_noise_scheduler.set_timesteps(n_ddim_steps)
        self.use_ddim_sample = True
        self._graphs.clear()

    def get_loss_weights(self, discount: float, action_weight: Optional[float] = None):
        """diffusion.py:149-172 with use_inv_dyn=True: discount**t over the prediction horizon, zeros over the
        history, broadcast over agents and observation dims -> [T, A, F]."""
        dim_weights = torch.ones(self.observation_dim, dtype=torch.float32)
        discounts = discount ** torch.arange(self.horizon, dtype=torch.float)
        discounts = discounts / discounts.mean()
        discounts = torch.cat([torch.zeros(self.history_horizon), discounts])
        w = torch.einsum("h,t->ht", discounts, dim_weights)
        return w.unsqueeze(1).expand(-1, self.n_agents, -1).clone()

    def _sched_tables(self, sched, device):
        """Device-resident tables of a scheduler: sqrt(abar), sqrt(1-abar), posterior coefficient rows, timesteps."""
        key = (id(sched), str(device), tuple(sched.timesteps.tolist()))
        tb = self._tables.get(key)
        if tb is None:
            ac = sched.alphas_cumprod
            tb = {"sqrt_ac": (ac ** 0.5).to(device).contiguous(), "sqrt_1mac": ((1 - ac) ** 0.5).to(device).contiguous(),
                  "coef": sched.coef_table().to(device).contiguous(),
                  "timesteps": sched.timesteps.to(device=device, dtype=torch.int64).contiguous()}
            self._tables = {key: tb}
        return tb

    # ------------------------------------------------------------------ sampling
    def get_model_output(self, x, t, returns=None, env_ts=None, attention_masks=None, states=None):
        """Guided epsilon (diffusion.py:176-212); the blend itself is fused into the posterior kernel during
        sampling, this entry point returns the blended tensor for callers that want it."""
        if self.returns_condition:
            if isinstance(self.model, SharedConvAttentionDeconv):
                ec, eu = self.model.forward_guided_pair(ops.pack_traj(x.contiguous(), pad32(x.shape[-1])), x.shape[0],
                                                        x.shape[1], t, returns, env_ts)
            else:
                ec = self.model(x, t, returns=returns, env_timestep=env_ts, attention_masks=attention_masks,
                                use_dropout=False)
                eu = self.model(x, t, returns=returns, env_timestep=env_ts, attention_masks=attention_masks,
                                use_dropout=False, force_dropout=True)
            return eu + self.condition_guidance_w * (ec - eu)
        return self.model(x, t, env_timestep=env_ts, attention_masks=attention_masks)

    def _denoise_step(self, st, returns, env_ts, bsz, t_len):
        """One loop iteration on persistent buffers (graph-capturable, device-driven: the step index lives in st["state"]):
        eps from x (already inpainted), then the fused posterior step -- guidance blend, x0, clip, mean, + sigma z with z
        read from the step's row of st["z_all"] or drawn on the device (Philox), inpainting -- writes the next x in place."""
        x = st["x"]
        lib = ops.lib
        lib.call("madiff_sampler_fill_time", st["state"].data_ptr(), st["tsteps"].data_ptr(), st["ts"].data_ptr(), bsz,
                 ops._stream())
        x = self._encode_(x)                                # diffusion.py:252 (x arrives inpainted)
        if isinstance(self.model, SharedConvAttentionDeconv):
            xcl = ops.pack_traj(x, pad32(x.shape[-1]))
            if self.returns_condition:
                ec, eu = self.model.forward_guided_pair(xcl, bsz, t_len, st["ts"], returns, env_ts, proj_R=st.get("R"))
            else:
                temb = self.model.net.embed(st["ts"], None, self.n_agents, False, False, None, env_ts)
                ec, eu = self.model.forward_cl(xcl, bsz, t_len, temb), None
        elif self.returns_condition:                        # any other denoiser: two passes (diffusion.py:187-202)
            ec = self.model(x, st["ts"], returns=returns, env_timestep=env_ts, attention_masks=None, use_dropout=False)
            # use_dropout=False: the reference's Bernoulli draw here is multiplied by zero (temporal.py:377-381), and a
            # host-side draw cannot be captured into the sampling graph
            eu = self.model(x, st["ts"], returns=returns, env_timestep=env_ts, attention_masks=None, use_dropout=False,
                            force_dropout=True)
        else:
            ec, eu = self.model(x, st["ts"], env_timestep=env_ts, attention_masks=None), None
        ops.posterior_step_rng(x, ec.contiguous(), None if eu is None else eu.contiguous(), self.condition_guidance_w,
                               st["z_all"], st["rng"] if st["z_all"] is None else None, st["cond_x"], st["mask"], st["coef"], x,
                               step_ptr=st["state"])
        lib.call("madiff_sampler_advance", st["state"].data_ptr(), ops._stream())

    @staticmethod
    def _graph_chunk(n_steps: int) -> int:
        """Denoising steps held by ONE captured graph.  Default: all of them -- the K-step loop is one graph launch
        (diffusion.py:249-263 without the host between steps).  MADIFF_SAMPLE_GRAPH_STEPS=<n> caps it (the graph then holds
        the largest divisor of K below the cap and is replayed K / chunk times; 1 = round 2's one-step graph)."""
        cap = int(os.environ.get("MADIFF_SAMPLE_GRAPH_STEPS") or 0)
        if cap <= 0 or cap >= n_steps:
            return n_steps
        return max(d for d in range(1, cap + 1) if n_steps % d == 0)

    @torch.no_grad()
    def conditional_sample(self, cond: Dict[str, torch.Tensor], returns: Optional[torch.Tensor] = None,
                           env_ts: Optional[torch.Tensor] = None, horizon: int = None,
                           attention_masks: Optional[torch.Tensor] = None, verbose: bool = True,
                           return_diffusion: bool = False, x_init: Optional[torch.Tensor] = None,
                           step_noise: Optional[torch.Tensor] = None, use_graph: Optional[bool] = None):
        """diffusion.py:214-275.  ``x_init`` = the tempered initial draw (0.5*randn); ``step_noise`` [n_steps, B,T,A,F]
        = the N(0,1) draws of the scheduler steps (row i is consumed by loop iteration i).  Without ``step_noise`` the
        scheduler's draws are made on the device (Philox4x32-10 keyed by a seed taken from torch's device generator, so
        ``torch.manual_seed`` fixes them; eager and graph runs of one seed give the same trajectories).
        On a GPU the whole K-step loop is ONE hipGraph launch (``use_graph``, default unless ``return_diffusion``)."""
        batch_size = cond["x"].shape[0]
        t_len = horizon or self.horizon + self.history_horizon
        shape = (batch_size, t_len, self.n_agents, self.observation_dim)
        device = cond["x"].device
        sched = self.ddim_noise_scheduler if self.use_ddim_sample else self.noise_scheduler
        tb = self._sched_tables(sched, device)
        n_steps = tb["timesteps"].shape[0]
        if use_graph is None:
            use_graph = not return_diffusion

        needs_z = not self.use_ddim_sample
        explicit = needs_z and step_noise is not None
        x0 = (0.5 * torch.randn(shape, device=device)) if x_init is None else x_init.to(device, torch.float32)
        if returns is not None:
            returns = returns.to(device, torch.float32).contiguous()
        # {seed, stream id} of the device-side draws: two words from torch's DEVICE generator (where the reference's
        # scheduler draws its noise too), made outside any graph
        rng_host = (torch.randint(0, 2 ** 62, (2,), dtype=torch.int64, device=device)
                    if (needs_z and not explicit) else None)

        # The captured loop is kept across calls (the evaluator samples once per environment step, evaluator.py:272-277): the
        # graph reads persistent buffers, so a later call with the same shapes and unchanged weights only copies its inputs
        # in and replays.  One entry is kept (its activations are the memory cost).
        chunk = self._graph_chunk(n_steps)
        key = (shape, str(device), id(sched), n_steps, returns is not None, None if env_ts is None else tuple(env_ts.shape),
               ops._PRECISION, ops._EPOCH, sum(p_._version for p_ in self.model.parameters()), self.condition_guidance_w,
               explicit, chunk)
        ent = _SAMPLE_CACHE.get(self)
        if not (use_graph and ent is not None and ent["key"] == key):
            ent = None
        if ent is not None:
            st, returns_s, env_ts_s, graph = ent["st"], ent["returns"], ent["env_ts"], ent["graph"]
            st["cond_x"].copy_(cond["x"])
            st["mask"].copy_(cond["masks"])
            if returns_s is not None:
                returns_s.copy_(returns)
            if env_ts_s is not None:
                env_ts_s.copy_(env_ts)
        else:
            _SAMPLE_CACHE.pop(self, None)
            st = {"x": torch.empty(shape, dtype=torch.float32, device=device),
                  "cond_x": cond["x"].to(device, torch.float32).contiguous().clone(),
                  "mask": cond["masks"].to(device=device, dtype=torch.uint8).contiguous().clone(),
                  "coef": tb["coef"], "tsteps": tb["timesteps"],
                  "state": torch.zeros(1, dtype=torch.int32, device=device),
                  "ts": torch.zeros(batch_size, dtype=torch.int64, device=device),
                  "rng": torch.zeros(2, dtype=torch.int64, device=device),
                  "z_all": torch.zeros((n_steps,) + shape, dtype=torch.float32, device=device) if explicit else None}
            returns_s = None if returns is None else returns.clone()
            env_ts_s = None if env_ts is None else env_ts.to(device).contiguous().clone()
            graph = None

        def reset():
            st["x"].copy_(x0)
            st["state"].zero_()
            ops.inpaint_(st["x"], st["cond_x"], st["mask"])       # step 1 of the first iteration (diffusion.py:251)

        if explicit:
            rows = min(n_steps, step_noise.shape[0])               # (the last step, t = 0, has sigma = 0: its row is never read)
            st["z_all"][:rows].copy_(step_noise[:rows].to(device, torch.float32).reshape((rows,) + shape))
        elif rng_host is not None:
            st["rng"].copy_(rng_host)
        reset()
        # the returns half of every time projection is constant over the K steps: once per call (modules.proj_static)
        if isinstance(self.model, SharedConvAttentionDeconv) and self.returns_condition and returns_s is not None \
                and env_ts_s is None:
            R = self.model.proj_static(returns_s, batch_size)
            if R is not None and st.get("R") is not None:
                st["R"].copy_(R)                                   # (the captured loop reads this buffer)
            elif graph is None:
                st["R"] = R
        chain = [st["x"].clone()] if return_diffusion else None
        if use_graph and n_steps > 1:
            if graph is None:
                # one eager iteration first (it fills the weight-pack cache, tunes kernel variants and warms the allocator),
                # then the loop is captured: `chunk` iterations per graph (all K by default), each reading the step index
                # from the device.  Capture does not execute; the replays below do.
                self._denoise_step(st, returns_s, env_ts_s, batch_size, t_len)
                reset()
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with ops.graph_capture(graph):
                    for _ in range(chunk):
                        self._denoise_step(st, returns_s, env_ts_s, batch_size, t_len)
                _SAMPLE_CACHE[self] = {"key": key, "st": st, "returns": returns_s, "env_ts": env_ts_s, "graph": graph,
                                       "chunk": chunk}
            for _ in range(n_steps // chunk):
                graph.replay()
            self.last_sample_launches = n_steps // chunk          # host launches of the K-step loop (1 = whole-loop graph)
            x = st["x"].clone()                                    # (the persistent buffer is reused by the next call)
        else:
            for i in range(n_steps):
                self._denoise_step(st, returns_s, env_ts_s, batch_size, t_len)
                if return_diffusion:
                    chain.append(st["x"].clone())
            self.last_sample_launches = None
            x = st["x"]
        # the posterior kernel already enforced the final conditioning (diffusion.py:268); then the encoder (:269)
        x = self._encode_(x)
        if return_diffusion:
            return x, torch.stack(chain, dim=1)
        return x

    p_sample_loop = conditional_sample

    # ------------------------------------------------------------------ training
    def _predict_noise(self, x_start, cond, t, returns=None, env_ts=None, noise=None, dropout_mask=None):
        """diffusion.py:290-302: draw the noise, add_noise + inpainting + layout change (one kernel), denoiser forward.
        -> (epsilon [B,T,A,F] fp32, noise)."""
        if noise is None:
            noise = torch.randn(x_start.shape, dtype=x_start.dtype, device=x_start.device)
        device = x_start.device
        tb = self._sched_tables(self.noise_scheduler, device)
        B, T, A, Fd = x_start.shape
        mask = cond["masks"]
        if mask.dtype != torch.uint8:
            mask = mask.view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8)
        mask = mask.to(device).contiguous()
        identity = isinstance(self.data_encoder, IdentityEncoder)
        fast = isinstance(self.model, SharedConvAttentionDeconv)
        x_noisy = None if (identity and fast) else torch.empty(B, T, A, Fd, dtype=torch.float32, device=device)
        xcl = ops.pack_traj(x_start.to(torch.float32), pad32(Fd), noise=noise.contiguous(), t=t.contiguous(),
                            sqrt_ac=tb["sqrt_ac"], sqrt_1mac=tb["sqrt_1mac"],
                            cond=cond["x"].to(device, torch.float32).contiguous(), mask=mask, x_noisy=x_noisy)
        if not identity:                       # diffusion.py:294 (after the conditioning)
            x_noisy = self._encode_(x_noisy)
            xcl = ops.pack_traj(x_noisy, pad32(Fd)) if fast else None
        if fast:
            temb = self.model.net.embed(t, returns if self.model.returns_condition else None, A, True, False,
                                        dropout_mask, env_ts)
            epsilon = self.model.forward_cl(xcl, B, T, temb)
        else:                                  # any other denoiser with the reference's call surface (the wrappers)
            kw = {} if dropout_mask is None else {"dropout_mask": dropout_mask}
            epsilon = self.model(x_noisy, t, returns=returns, env_timestep=env_ts, attention_masks=None, **kw)
        assert noise.shape == epsilon.shape
        return epsilon, noise

    def _encode_(self, x):
        """self.data_encoder applied to a [B,T,A,F] fp32 trajectory, in place where the encoder allows it."""
        enc = self.data_encoder
        if isinstance(enc, IdentityEncoder):
            return x
        k = getattr(enc, "first_k", None)
        if k is not None and x.is_cuda and x.is_contiguous() and x.dtype == torch.float32:
            B, T, A, Fd = x.shape
            ops.lib.call("madiff_encode_first_k", x.data_ptr(), B, T, A, Fd, int(k), ops._stream())
            return x
        return enc(x)

    def p_losses(self, x_start, cond, t, loss_masks, attention_masks=None, returns=None, env_ts=None, states=None,
                 noise=None, dropout_mask=None):
        """diffusion.py:279-337.  add_noise + inpainting + layout change are one kernel; the masked weighted MSE
        and its gradient are one kernel each."""
        epsilon, noise = self._predict_noise(x_start, cond, t, returns, env_ts, noise, dropout_mask)
        B, T, A, Fd = x_start.shape
        device = x_start.device
        m32 = loss_masks.to(device=device, dtype=torch.float32).reshape(B, T, A).contiguous()
        loss, a0 = MaskedStateLoss.apply(epsilon, noise, self.loss_fn.weights.contiguous(), m32)
        if loss_masks.dtype != torch.float64:
            loss = loss.to(loss_masks.dtype)
        return loss, {"a0_loss": a0}

    def compute_inv_loss(self, x, loss_masks, legal_actions=None):
        """diffusion.py:369-436.  The MLP contractions run on the HIP GEMM; the tiny per-row loss algebra on the
        logits stays in torch."""
        info = {}
        ad, od = self.action_dim, self.observation_dim
        x_comb_t = torch.cat([x[:, :-1, :, ad:], x[:, 1:, :, ad:]], dim=-1).reshape(-1, x.shape[2], 2 * od)
        a_t = x[:, :-1, :, :ad].reshape(-1, x.shape[2], ad)
        masks_t = loss_masks[:, 1:].reshape(-1, loss_masks.shape[2])
        if legal_actions is not None:
            legal_actions_t = legal_actions[:, :-1].reshape(-1, *legal_actions.shape[2:])
        if self.joint_inv or self.share_inv:
            if self.joint_inv:
                pred_a_t = self.inv_model(x_comb_t.reshape(x_comb_t.shape[0], -1)).reshape(
                    x_comb_t.shape[0], x_comb_t.shape[1], -1)
            else:
                pred_a_t = self.inv_model(x_comb_t)
            if legal_actions is not None:
                pred_a_t = torch.where(legal_actions_t == 0, torch.full_like(pred_a_t, -1e10), pred_a_t)
            if self.discrete_action:
                inv_loss = (F.cross_entropy(pred_a_t.reshape(-1, pred_a_t.shape[-1]), a_t.reshape(-1).long(),
                                            reduction="none") * masks_t.reshape(-1)).mean() / masks_t.mean()
                inv_acc = ((pred_a_t.argmax(dim=-1, keepdim=True) == a_t).to(dtype=float).squeeze(-1)
                           * masks_t).mean() / masks_t.mean()
                info["inv_acc"] = inv_acc
            else:
                inv_loss = (F.mse_loss(pred_a_t, a_t, reduction="none") * masks_t.unsqueeze(-1)).mean() / masks_t.mean()
        else:
            inv_loss = 0.0
            for i in range(self.n_agents):
                pred_a_t = self.inv_model[i](x_comb_t[:, i])
                if self.discrete_action:
                    inv_loss += (F.cross_entropy(pred_a_t, a_t[:, i].reshape(-1).long(), reduction="none")
                                 * masks_t[:, i]).mean() / masks_t[:, i].mean()
                else:
                    inv_loss += (F.mse_loss(pred_a_t, a_t[:, i]) * masks_t[:, i].unsqueeze(-1)).mean() \
                        / masks_t[:, i].mean()
        return inv_loss, info

    def loss(self, x, cond, loss_masks, attention_masks=None, returns=None, env_ts=None, states=None,
             legal_actions=None, t=None, noise=None, dropout_mask=None, micro_batches: int = 1):
        """diffusion.py:438-496; called as ``loss(**batch)``.

        micro_batches = S > 1 (fused GPU loss only): the batch is S micro-batches of one optimizer step concatenated along
        the batch axis; the returned loss is the mean of the S losses the reference's accumulation loop would compute
        (training.py:147-152; the inverse-dynamics term is normalised per micro-batch), so ONE backward gives the
        accumulated gradient.  info then holds the LAST micro-batch's values plus "micro_loss" = its loss."""
        # The inverse-dynamics branch shares nothing with the denoiser branch until the final sum, and it is a chain of
        # small launches: on the GPU it runs on a second stream (forward here, backward there too, since autograd replays
        # a node on its forward stream), i.e. as a parallel branch of the captured train-step graph.
        fused = (x.is_cuda and FUSED_LOSS and not self.train_only_inv and (self.share_inv or self.joint_inv)
                 and x.dtype == torch.float32 and loss_masks.dtype in (torch.float32, torch.float64))
        if fused:
            return self._loss_fused(x, cond, loss_masks, returns, env_ts, legal_actions, t, noise, dropout_mask,
                                    int(micro_batches))
        if micro_batches != 1:
            raise NotImplementedError("micro_batches > 1 needs the fused GPU loss (share_inv or joint_inv, fp32 batch on the "
                                      "GPU): call loss() once per micro-batch instead")
        side = None
        if x.is_cuda and not self.train_only_inv and INV_SIDE_STREAM:
            cur = torch.cuda.current_stream()
            side = ops.aux_stream(x.device)
            ops.wait_stream(side, cur)
            with torch.cuda.stream(side):
                inv_loss, inv_info = self.compute_inv_loss(x, loss_masks, legal_actions)
        if self.train_only_inv:
            info = {}
        else:
            batch_size = len(x)
            if t is None:
                t = torch.randint(0, self.noise_scheduler.config.num_train_timesteps, (batch_size,),
                                  device=x.device).long()
            diffuse_loss, info = self.p_losses(x[..., self.action_dim:], cond, t, loss_masks, attention_masks, returns,
                                               env_ts, states, noise=noise, dropout_mask=dropout_mask)
        if side is None:
            inv_loss, inv_info = self.compute_inv_loss(x, loss_masks, legal_actions)
        else:
            ops.wait_stream(cur, side)
            for v in [inv_loss, *inv_info.values()]:
                if torch.is_tensor(v):
                    v.record_stream(cur)
        info = {**info, **inv_info}
        info["inv_loss"] = inv_loss
        if self.train_only_inv:
            return inv_loss, info
        return (1 / 2) * (diffuse_loss + inv_loss), info

    def _loss_fused(self, x, cond, loss_masks, returns, env_ts, legal_actions, t, noise, dropout_mask, segments=1):
        """loss() for the shared / joint inverse-dynamics head on the GPU: the MLP's inputs come straight from the batch
        tensor (madiff_inv_pairs), its logits and the denoiser's epsilon go into ONE loss node (ops.FusedLoss: two launches
        forward, one backward) -- no ATen kernel between the two networks and the scalar loss."""
        B, T, A, _ = x.shape
        x = x.contiguous()
        nout = self.num_actions if self.discrete_action else self.action_dim
        cur = torch.cuda.current_stream()
        side = ops.aux_stream(x.device) if INV_SIDE_STREAM else cur
        ops.wait_stream(side, cur)
        with torch.cuda.stream(side):
            logits = self.inv_model.logits_cl(ops.inv_pairs(x, self.action_dim, self.joint_inv))
        if t is None:
            t = torch.randint(0, self.noise_scheduler.config.num_train_timesteps, (B,), device=x.device).long()
        epsilon, noise = self._predict_noise(x[..., self.action_dim:], cond, t, returns, env_ts, noise, dropout_mask)
        ops.wait_stream(cur, side)
        logits.record_stream(cur)
        if B % segments:
            raise ValueError(f"batch of {B} does not split into {segments} equal micro-batches")
        loss, a0, inv_loss, inv_acc, last = ops.FusedLoss.apply(
            epsilon, noise, self.loss_fn.weights, loss_masks.reshape(B, T, A), logits, x, legal_actions,
            (self.action_dim, nout, self.discrete_action, self.joint_inv, segments))
        info = {"a0_loss": a0}
        if segments > 1:
            info["micro_loss"] = last
        if self.discrete_action:
            info["inv_acc"] = inv_acc
        info["inv_loss"] = inv_loss
        return loss, info

    def forward(self, cond, *args, **kwargs):
        return self.conditional_sample(cond=cond, *args, **kwargs)
