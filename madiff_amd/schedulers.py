"""DDPM / DDIM noise schedulers with the call surface MADiff uses from diffusers==0.24.0.

The reference delegates this arithmetic to the third-party package (requirements.txt:5; call sites
diffuser/models/diffusion.py:87-92,139-147,246,261,292,456), which is neither vendored nor installed here;
the formulas below restate the published algorithm: squaredcos_cap_v2 betas
(float64 python math stored as float32), epsilon prediction, clip_sample=True with range 1, variance type
"fixed_small", DDIM with eta=0, "leading" spacing and set_alpha_to_one.  Parity for this part is unpinned
by any reference fixture; tests/test_scheduler.py freezes known-answer values.

Besides the diffusers-like host API (``add_noise``, ``step(...).prev_sample``, ``timesteps``,
``config.num_train_timesteps``) each scheduler exports ``coef_table()``: the per-step coefficient rows
consumed by the fused HIP posterior kernel (madiff_posterior_step).
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch


class DDPMScheduler:
    def __init__(self, num_train_timesteps=1000, clip_sample=True, prediction_type="epsilon",
                 beta_schedule="squaredcos_cap_v2", **_ignored):
        if not clip_sample or prediction_type != "epsilon" or beta_schedule != "squaredcos_cap_v2":
            raise NotImplementedError("only the configuration MADiff uses is implemented")
        K = int(num_train_timesteps)
        self.config = SimpleNamespace(num_train_timesteps=K, clip_sample=True, prediction_type="epsilon",
                                      beta_schedule=beta_schedule)

        def abar(s):
            return math.cos((s + 0.008) / 1.008 * math.pi / 2) ** 2

        self.betas = torch.tensor([min(1 - abar((i + 1) / K) / abar(i / K), 0.999) for i in range(K)],
                                  dtype=torch.float32)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.timesteps = torch.arange(K - 1, -1, -1)
        self._dev_cache = {}

    # ---- host API (eager, any device) -----------------------------------------------------------
    def add_noise(self, original_samples, noise, timesteps):
        ac = self.alphas_cumprod.to(device=original_samples.device, dtype=original_samples.dtype)
        t = timesteps.to(original_samples.device)
        shp = (-1,) + (1,) * (original_samples.dim() - 1)
        return (ac[t] ** 0.5).reshape(shp) * original_samples + ((1 - ac[t]) ** 0.5).reshape(shp) * noise

    def _prev(self, t: int) -> int:
        return t - 1

    def _row(self, t: int):
        ac = self.alphas_cumprod
        a_t = ac[t]
        tp = self._prev(t)
        a_prev = ac[tp] if tp >= 0 else torch.tensor(1.0)
        b_t, b_prev = 1 - a_t, 1 - a_prev
        cur_a = a_t / a_prev
        cur_b = 1 - cur_a
        c0 = a_prev ** 0.5 * cur_b / b_t
        ct = cur_a ** 0.5 * b_prev / b_t
        sigma = torch.clamp(b_prev / b_t * cur_b, min=1e-20) ** 0.5 if t > 0 else torch.tensor(0.0)
        return [b_t ** 0.5, 1.0 / a_t ** 0.5, c0, ct, torch.tensor(0.0), sigma]

    def coef_table(self) -> torch.Tensor:
        """[len(timesteps), 6] fp32 rows {sqrt(1-abar), 1/sqrt(abar), c_x0, c_xt, c_eps, sigma}, in loop order."""
        return torch.stack([torch.stack([torch.as_tensor(v, dtype=torch.float32) for v in self._row(int(t))])
                            for t in self.timesteps.tolist()])

    def step(self, model_output, timestep, sample, generator=None, noise=None):
        r = [float(v) for v in self._row(int(timestep))]
        x0 = ((sample - r[0] * model_output) * r[1]).clamp(-1.0, 1.0)
        prev = r[2] * x0 + r[3] * sample + r[4] * model_output
        if r[5] != 0.0:
            if noise is None:
                noise = torch.randn(sample.shape, generator=generator, device=sample.device, dtype=sample.dtype)
            prev = prev + r[5] * noise
        return SimpleNamespace(prev_sample=prev, pred_original_sample=x0)


class DDIMScheduler(DDPMScheduler):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self._ratio = 1

    def set_timesteps(self, num_inference_steps: int):
        K = self.config.num_train_timesteps
        self._ratio = K // int(num_inference_steps)
        self.timesteps = (torch.arange(int(num_inference_steps)) * self._ratio).flip(0)

    def _prev(self, t: int) -> int:
        return t - self._ratio

    def _row(self, t: int):
        ac = self.alphas_cumprod
        a_t = ac[t]
        tp = self._prev(t)
        a_prev = ac[tp] if tp >= 0 else torch.tensor(1.0)
        z = torch.tensor(0.0)
        return [(1 - a_t) ** 0.5, 1.0 / a_t ** 0.5, a_prev ** 0.5, z, (1 - a_prev) ** 0.5, z]
