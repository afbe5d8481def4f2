"""Known-answer freeze of the restated diffusers==0.24.0 DDPM/DDIM schedule.

PARITY UNPINNED: no reference test/fixture covers the scheduler (the package is absent); these
values freeze the spec so that the oracle and the product cannot drift apart silently.
"""
import math

import numpy as np
import torch

from oracle.madiff_oracle import DDIMSchedule, DDPMSchedule


def test_alpha_bar_known_answers():
    s = DDPMSchedule(200)
    ac = s.alphas_cumprod
    assert ac.dtype == torch.float32
    np.testing.assert_allclose(ac[0].item(), 0.9997450113, rtol=2e-7)
    np.testing.assert_allclose(ac[1].item(), 0.9993687272, rtol=2e-7)
    np.testing.assert_allclose(ac[100].item(), 0.4860524535, rtol=2e-6)
    np.testing.assert_allclose(ac[199].item(), 6.0717e-08, rtol=1e-3)
    assert abs(s.betas[199].item() - 0.999) < 1e-7          # the cap is active on the last step
    assert s.timesteps.tolist() == list(range(199, -1, -1))


def test_posterior_coefficients_self_consistent():
    s = DDPMSchedule(200)
    for t in (0, 1, 100, 199):
        sb, sa, c0, ct, sigma = [float(v) for v in s.posterior_coeffs(t, torch.float64)]
        ac = s.alphas_cumprod.double()
        a_t = ac[t].item()
        a_p = ac[t - 1].item() if t > 0 else 1.0
        assert math.isclose(sb ** 2 + sa ** 2, 1.0, rel_tol=1e-12)
        beta = 1 - a_t / a_p
        assert math.isclose(c0, math.sqrt(a_p) * beta / (1 - a_t), rel_tol=1e-12)
        assert math.isclose(ct, math.sqrt(a_t / a_p) * (1 - a_p) / (1 - a_t), rel_tol=1e-12)
        if t == 0:
            assert sigma == 0.0 and ct == 0.0
        else:
            assert math.isclose(sigma ** 2, max((1 - a_p) / (1 - a_t) * beta, 1e-20), rel_tol=1e-12)


def test_step_and_add_noise():
    s = DDPMSchedule(200)
    x0 = torch.linspace(-1, 1, 24).reshape(2, 3, 4)
    n = torch.linspace(1, -1, 24).reshape(2, 3, 4)
    t = torch.tensor([0, 199])
    xn = s.add_noise(x0, n, t)
    np.testing.assert_allclose(xn[0].numpy(), (s.alphas_cumprod[0] ** 0.5 * x0[0] + (1 - s.alphas_cumprod[0]) ** 0.5 * n[0]).numpy(), rtol=1e-6)
    # exact noise -> the step at t=0 returns the (clipped) x0
    out = s.step(n[:1], 0, xn[:1], None)
    np.testing.assert_allclose(out.numpy(), x0[:1].numpy(), atol=2e-6)
    # clipping is always on (clip_sample=True, range 1)
    big = s.step(torch.zeros(1, 3, 4), 0, 10 * torch.ones(1, 3, 4), None)
    assert float(big.max()) <= 1.0


def test_ddim_timesteps():
    d = DDIMSchedule(200, 15)
    assert d.ratio == 13
    assert d.timesteps.tolist() == [13 * i for i in range(14, -1, -1)]


def test_philox4x32_10_known_answers():
    """The oracle's restatement of the counter-based generator behind the sampler's device-side noise, against the
    known-answer vectors published with the algorithm (Random123 kat_vectors, philox4x32 10 rounds)."""
    from oracle.madiff_oracle import philox4x32_10, philox_normal
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = philox4x32_10(np.array([ctr], dtype=np.uint32), np.array(key, dtype=np.uint32))[0]
        assert tuple(int(v) for v in got) == want
    z = philox_normal(seed=0x1234567890ABCDEF, stream=5, step=17, numel=(1 << 18) + 3)
    assert z.shape == ((1 << 18) + 3,)
    assert abs(z.mean()) < 6e-3 and abs(z.std() - 1) < 6e-3 and abs((z ** 4).mean() - 3) < 0.08
    z2 = philox_normal(seed=0x1234567890ABCDEF, stream=5, step=18, numel=1 << 18)
    assert abs(np.corrcoef(z[: 1 << 18], z2)[0, 1]) < 8e-3         # another step: another stream of draws
