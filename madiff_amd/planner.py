"""The evaluator's planning step with every buffer on the device.

Reference: ``MADEvaluatorWorker._generate_samples`` (diffuser/utils/evaluator.py:38-135) builds the conditioning
trajectories and masks in numpy on the host for every environment step (centralized: all agents visible; decentralized:
one copy of each episode per agent, ``(b a)`` expansion, copy ``a`` sees only agent ``a``'s history), moves them to the
device, calls ``ema_model.conditional_sample`` and slices the plan; the episode loop (:272-296) then feeds the first two
planned observations to the inverse-dynamics model and picks actions.  Here the same arithmetic runs on persistent
device buffers: the masks are built once (they depend only on the shapes, utils/mask_generator.py:26-59), the
observation history is written into the conditioning buffer by one indexed copy, the sampler replays its kept hipGraph
(GaussianDiffusion.conditional_sample), the inverse-dynamics MLP runs on the HIP GEMM, and the legal-action arg-max is a
device op; the only host transfer left per environment step is the action tensor the simulator needs.
"""
from __future__ import annotations

from typing import Optional

import torch


class DevicePlanner:
    def __init__(self, diffusion, decentralized_execution: bool = False, max_path_length: Optional[int] = None):
        """diffusion: the (EMA) GaussianDiffusion.  Shapes come from it: n_agents, horizon, history_horizon,
        observation_dim; share_inv / joint_inv / discrete_action select the action head (evaluator.py:282-296)."""
        self.d = diffusion
        self.A, self.h, self.hh, self.F = diffusion.n_agents, diffusion.horizon, diffusion.history_horizon, \
            diffusion.observation_dim
        self.T = self.h + self.hh
        self.decentralized = bool(decentralized_execution)
        self.max_path_length = max_path_length
        self._buf = {}
        self._hist = None

    # ------------------------------------------------------------------ conditions (evaluator.py:50-132)
    def _buffers(self, B: int, device):
        key = (B, str(device))
        b = self._buf.get(key)
        if b is not None:
            return b
        A, T, F, hh = self.A, self.T, self.F, self.hh
        steps = torch.arange(T, device=device)
        vis = (steps < hh + 1).view(T, 1, 1).expand(T, A, F)        # visible agents: history and the current observation
        inv = (steps < hh).view(T, 1, 1).expand(T, A, F)           # invisible agents: (zeroed) history only
        if not self.decentralized:
            masks = vis.unsqueeze(0).expand(B, T, A, F).contiguous()
            cond_x = torch.zeros(B, T, A, F, device=device)
        else:
            eye = torch.eye(A, dtype=torch.bool, device=device)                       # [copy, agent]
            m = torch.where(eye.view(A, 1, A, 1), vis.unsqueeze(0), inv.unsqueeze(0))  # [copy, T, A, F]
            masks = m.unsqueeze(0).expand(B, A, T, A, F).reshape(B * A, T, A, F).contiguous()
            cond_x = torch.zeros(B * A, T, A, F, device=device)
        b = self._buf[key] = {"cond_x": cond_x, "masks": masks}
        return b

    def conditions(self, obs: torch.Tensor):
        """obs [B, hh+1, A, F] on the device (normalised history, newest last) -> {"x", "masks"} (persistent buffers)."""
        B = obs.shape[0]
        assert obs.shape[1:] == (self.hh + 1, self.A, self.F), obs.shape
        b = self._buffers(B, obs.device)
        cx = b["cond_x"]
        if not self.decentralized:
            cx[:, : self.hh + 1].copy_(obs)
        else:
            v = cx.view(B, self.A, self.T, self.A, self.F)
            idx = torch.arange(self.A, device=obs.device)
            v[:, idx, : self.hh + 1, idx] = obs.permute(2, 0, 1, 3)      # copy a, agent a <- agent a's history
        return {"x": cx, "masks": b["masks"]}

    # ------------------------------------------------------------------ observation history (evaluator.py:259-275)
    def reset_history(self, first_obs: torch.Tensor, use_zero_padding: bool = True) -> None:
        """Start an episode batch: the evaluator's `deque(maxlen=history_horizon + 1)` pre-filled with history_horizon
        entries -- zeros (`use_zero_padding`) or copies of the first normalised observation [B, A, F] -- kept on the
        device as one [B, hh+1, A, F] buffer (slot 0 is overwritten by the first push)."""
        B = first_obs.shape[0]
        assert first_obs.shape[1:] == (self.A, self.F), first_obs.shape
        h = torch.zeros(B, self.hh + 1, self.A, self.F, device=first_obs.device, dtype=torch.float32)
        if not use_zero_padding and self.hh > 0:
            h[:, 1:] = first_obs.to(torch.float32).unsqueeze(1)
        self._hist = h

    def push_obs(self, obs: torch.Tensor) -> torch.Tensor:
        """`obs_queue.append(obs); np.stack(list(obs_queue), axis=1)` (evaluator.py:273-275) on the device: drops the
        oldest entry, appends the normalised observation [B, A, F], returns the history [B, hh+1, A, F] (newest last)."""
        h = getattr(self, "_hist", None)
        if h is None:
            raise RuntimeError("DevicePlanner.push_obs: call reset_history(first_obs) at the start of an episode batch")
        if self.hh > 0:
            h[:, :-1] = h[:, 1:].clone()
        h[:, -1] = obs.to(h.dtype)
        return h.clone()          # a fresh tensor per step, like the evaluator's np.stack: callers may keep it

    # ------------------------------------------------------------------ sampling (evaluator.py:85-134)
    @torch.no_grad()
    def generate_samples(self, obs: torch.Tensor, returns: Optional[torch.Tensor] = None,
                         env_ts: Optional[torch.Tensor] = None, **sample_kwargs) -> torch.Tensor:
        """-> planned observations [B, horizon, A, F] (the history part is cut off, evaluator.py:134)."""
        B, A = obs.shape[0], self.A
        cond = self.conditions(obs)
        if env_ts is not None and self.max_path_length is not None:               # evaluator.py:42-44
            env_ts = env_ts.clone()
            env_ts[(env_ts < 0) | (env_ts >= self.max_path_length)] = self.max_path_length
        if self.decentralized:
            if returns is not None:
                returns = returns.repeat_interleave(A, dim=0)
            if env_ts is not None:
                env_ts = env_ts.repeat_interleave(A, dim=0)
        x = self.d.conditional_sample(cond, returns=returns, env_ts=env_ts, verbose=False, **sample_kwargs)
        if self.decentralized:                                                    # evaluator.py:107-114
            x = x.view(B, A, self.T, A, self.F)
            idx = torch.arange(A, device=x.device)
            x = x[:, idx, :, idx].permute(1, 2, 0, 3)                              # [B, T, A, F]: copy a -> agent a
        return x[:, self.hh:]

    # ------------------------------------------------------------------ actions (evaluator.py:279-313)
    @torch.no_grad()
    def act(self, obs: torch.Tensor, returns: Optional[torch.Tensor] = None, env_ts: Optional[torch.Tensor] = None,
            legal_actions: Optional[torch.Tensor] = None, **sample_kwargs):
        """-> (actions, samples).  Continuous: normalised actions [B, A, act_dim] (the caller un-normalises with its
        dataset normaliser); discrete: action ids [B, A] = arg-max over the legal actions."""
        d = self.d
        samples = self.generate_samples(obs, returns, env_ts, **sample_kwargs)
        B, A = samples.shape[0], self.A
        comb = torch.cat([samples[:, 0], samples[:, 1]], dim=-1).reshape(-1, A, 2 * self.F)
        if d.joint_inv:
            actions = d.inv_model(comb.reshape(B, -1)).reshape(B, A, -1)
        elif d.share_inv:
            actions = d.inv_model(comb)
        else:
            actions = torch.stack([d.inv_model[i](comb[:, i]) for i in range(A)], dim=1)
        if d.discrete_action:
            actions = actions.float()
            if legal_actions is not None:
                actions = actions.masked_fill(legal_actions.to(actions.device) == 0, float("-inf"))
            actions = actions.argmax(dim=-1)
        return actions, samples

    @staticmethod
    def update_return_to_go(rtg: torch.Tensor, reward: torch.Tensor, returns_scale: float, discount: float) -> torch.Tensor:
        """evaluator.py:218-223, on the device."""
        reward = torch.as_tensor(reward, device=rtg.device, dtype=rtg.dtype).reshape(1, -1)
        return ((rtg * returns_scale - reward) / discount) / returns_scale
