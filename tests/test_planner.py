"""Planning step of the evaluator (madiff_amd/planner.py; reference diffuser/utils/evaluator.py:38-135,218-223,277-313).
CPU part: the oracle's mask rule against masks produced by the reference's own MultiAgentMaskGenerator
(tests/golden/g7_masks.npz, written by oracle/gen_golden.py), and the device planner's condition builder against the
oracle.  GPU part: conditions -> guided K-step sample -> inverse dynamics -> actions against the oracle, centralized and
decentralized, continuous and discrete (legal-action masking), fp32 build."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import madiff_oracle as O

SHAPES = {"spread": (3, 24, 18, 0), "ant": (4, 28, 54, 18), "smac": (8, 24, 88, 20)}


def test_oracle_mask_rule_matches_reference_generator(golden):
    g = golden("g7_masks")
    for name, (A, T, F, hh) in SHAPES.items():
        for tag, am in (("all", np.ones(A)), ("only0", np.eye(A)[0]), (f"only{A - 1}", np.eye(A)[A - 1])):
            shp = tuple(int(v) for v in g[f"{name}.{tag}.shape"])
            ref = np.unpackbits(g[f"{name}.{tag}"])[: int(np.prod(shp))].reshape(shp).astype(bool)
            assert np.array_equal(O.agent_condition_mask(shp, am, hh), ref), (name, tag)


class _Shape:
    def __init__(self, A, h, hh, F):
        self.n_agents, self.horizon, self.history_horizon, self.observation_dim = A, h, hh, F


def test_history_is_a_fresh_tensor_per_step_and_needs_reset():
    """evaluator.py:273-275 builds a new array per environment step (np.stack(list(obs_queue))): a history a caller keeps
    must not change under it at the next push; and pushing before reset_history says so."""
    from madiff_amd.planner import DevicePlanner
    pl = DevicePlanner(_Shape(3, 4, 2, 5), False)
    with pytest.raises(RuntimeError, match="reset_history"):
        pl.push_obs(torch.zeros(2, 3, 5))
    pl.reset_history(torch.zeros(2, 3, 5))
    h1 = pl.push_obs(torch.full((2, 3, 5), 1.0))
    kept = h1.clone()
    h2 = pl.push_obs(torch.full((2, 3, 5), 2.0))
    assert torch.equal(h1, kept) and h1.data_ptr() != h2.data_ptr()
    assert torch.equal(h2[:, -1], torch.full((2, 3, 5), 2.0)) and torch.equal(h2[:, -2], torch.full((2, 3, 5), 1.0))


@pytest.mark.parametrize("tag", ["cen", "dec", "dec_nohist"])
def test_planner_calls_sampler_like_reference_evaluator(golden, tag):
    """tests/golden/g10_planner.npz was recorded from the reference's own MADEvaluatorWorker._generate_samples
    (evaluator.py:39-135) driving a recording stand-in sampler: the conditioning trajectories / masks it builds, the
    returns / env_ts it passes (clamp at max_path_length, (b a) expansion) and the plans it hands back (decentralized
    gather, history cut) -- DevicePlanner must call its sampler with the same tensors and return the same plans, both
    paths fed with and returning the same closed-form stand-in plan."""
    from madiff_amd.planner import DevicePlanner
    g = golden("g10_planner")
    A, h, hh, F, B, dec = (int(v) for v in g[f"{tag}.geom"])
    calls = []

    class Recording(_Shape):
        def conditional_sample(self, cond, returns=None, env_ts=None, verbose=False, **kw):
            calls.append((cond["x"].clone(), cond["masks"].clone(), returns.clone(), env_ts.clone()))
            return torch.from_numpy(np.asarray(O.closed_form_normal(f"g10.{tag}.plan", tuple(cond["x"].shape)), dtype=np.float32))

    pl = DevicePlanner(Recording(A, h, hh, F), bool(dec), max_path_length=25)
    out = pl.generate_samples(torch.from_numpy(g[f"{tag}.obs"]), torch.from_numpy(g[f"{tag}.returns"]),
                              torch.from_numpy(g[f"{tag}.env_ts"]))
    cx, cm, r, ets = calls[0]
    assert np.array_equal(cx.numpy(), g[f"{tag}.cond_x"])
    assert np.array_equal(cm.numpy().astype(np.uint8), g[f"{tag}.cond_masks"])
    assert np.array_equal(r.numpy(), g[f"{tag}.call_returns"])
    assert np.array_equal(ets.numpy(), g[f"{tag}.call_env_ts"])
    assert np.array_equal(out.numpy(), g[f"{tag}.samples"])


@pytest.mark.parametrize("decentralized", [False, True])
@pytest.mark.parametrize("hh", [0, 5])
def test_planner_conditions_match_oracle(decentralized, hh):
    from madiff_amd.planner import DevicePlanner
    A, h, F, B = 4, 7, 6, 3
    obs = np.random.default_rng(hh).normal(size=(B, hh + 1, A, F)).astype(np.float32)
    pl = DevicePlanner(_Shape(A, h, hh, F), decentralized)
    for _ in range(2):                                    # second call: the persistent buffers are rewritten in place
        c = pl.conditions(torch.from_numpy(obs))
        cx, cm = O.planner_conditions(obs, h, hh, decentralized)
        assert np.array_equal(c["x"].numpy(), cx) and np.array_equal(c["masks"].numpy(), cm)
        obs = obs * 0.5 + 1.0


@pytest.mark.parametrize("zero_pad", [True, False])
@pytest.mark.parametrize("hh", [0, 3])
def test_observation_history_matches_evaluator_queue(zero_pad, hh):
    """The evaluator's deque (evaluator.py:263-275): maxlen hh + 1, pre-filled with hh zeros or hh copies of the first
    normalised observation; each environment step appends and stacks."""
    from collections import deque
    from madiff_amd.planner import DevicePlanner
    A, F, B = 3, 4, 2
    rng = np.random.default_rng(3)
    obs_seq = [rng.normal(size=(B, A, F)).astype(np.float32) for _ in range(6)]
    q = deque(maxlen=hh + 1)
    q.extend([np.zeros_like(obs_seq[0]) if zero_pad else obs_seq[0] for _ in range(hh)])
    pl = DevicePlanner(_Shape(A, 5, hh, F), False)
    pl.reset_history(torch.from_numpy(obs_seq[0]), use_zero_padding=zero_pad)
    for o in obs_seq:
        q.append(o)
        ref = np.stack(list(q), axis=1)
        got = pl.push_obs(torch.from_numpy(o))
        assert np.array_equal(got.numpy(), ref)


def test_update_return_to_go():
    from madiff_amd.planner import DevicePlanner
    rtg = torch.rand(5, 1, 3)
    r = np.array([0.3, -0.2, 1.0])
    assert torch.allclose(DevicePlanner.update_return_to_go(rtg, r, 400.0, 0.99), O.update_return_to_go(rtg, r, 400.0, 0.99))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["central", "decentral", "central_discrete", "decentral_indep_inv"])
def test_plan_actions_match_oracle(mode):
    import diffuser.models as DM
    from madiff_amd import ops
    from madiff_amd.planner import DevicePlanner
    dev = torch.device("cuda:0")
    ops.set_precision("fp32")
    try:
        dec = mode.startswith("decentral")
        discrete = mode.endswith("discrete")
        share_inv = not mode.endswith("indep_inv")
        A, h, hh, F, K, B = 3, 6, 6, 10, 8, 2
        T, act, nact = h + hh, (1 if discrete else 4), (6 if discrete else 0)
        net = DM.SharedConvAttentionDeconv(horizon=T, transition_dim=F, dim=16, history_horizon=hh, dim_mults=(1, 4, 8),
                                           n_agents=A, returns_condition=True, condition_dropout=0.25)
        dif = DM.GaussianDiffusion(net, n_agents=A, horizon=h, history_horizon=hh, observation_dim=F, action_dim=act,
                                   discrete_action=discrete, num_actions=nact, n_timesteps=K, returns_condition=True,
                                   hidden_dim=32, share_inv=share_inv).to(dev)
        sd = {k: (O.closed_form_param(k, tuple(v.shape)) if "loss_fn" not in k else v) for k, v in dif.state_dict().items()}
        dif.load_state_dict(sd)
        params = {k: v.clone() for k, v in sd.items() if "loss_fn" not in k}
        dcfg = O.DenoiserCfg(horizon=T, transition_dim=F, dim=16, dim_mults=(1, 4, 8), n_agents=A, returns_condition=True,
                             condition_dropout=0.25, history_horizon=hh)
        c = O.DiffusionCfg(A, h, hh, F, act, K, hidden_dim=32, returns_condition=True, discrete_action=discrete,
                           num_actions=nact, share_inv=share_inv)
        rng = np.random.default_rng(3)
        obs = rng.uniform(-1, 1, size=(B, hh + 1, A, F)).astype(np.float32)
        returns = torch.from_numpy(rng.uniform(0, 1, size=(B, 1, A)).astype(np.float32))
        legal = (rng.uniform(size=(B, A, nact)) < 0.6).astype(np.float32) if discrete else None
        if discrete:
            legal[..., 0] = 1.0
        Bx = B * A if dec else B
        shape = (Bx, T, A, F)
        x_init = 0.5 * torch.from_numpy(O.closed_form_normal(f"plan.{mode}.0", shape))
        z = [torch.from_numpy(O.closed_form_normal(f"plan.{mode}.{i}", shape)) for i in range(1, K)]
        with torch.no_grad():
            ref_a, ref_s = O.plan_actions(params, dcfg, c, O.DDPMSchedule(K), obs, returns, x_init, z, dec, legal)
        pl = DevicePlanner(dif, decentralized_execution=dec)
        for rep in range(2):                              # the second call replays the sampler's kept graph
            a, s = pl.act(torch.from_numpy(obs).to(dev), returns.to(dev), None,
                          None if legal is None else torch.from_numpy(legal).to(dev),
                          x_init=x_init.to(dev), step_noise=torch.stack(z).to(dev))
            assert s.shape == (B, h, A, F)
            assert rel_err(s.cpu(), ref_s) < 1e-3, (mode, rep)
            if discrete:
                assert torch.equal(a.cpu(), ref_a), (mode, rep)
            else:
                assert rel_err(a.cpu(), ref_a) < 1e-3, (mode, rep)
    finally:
        ops.set_precision("bf16")
