"""Driver-level boundary: what /root/reference/run_scripts/train.py:41-215 does with the plugin surface, end to end on
the GPU -- the utils.Config objects built with exactly the keyword arguments the driver passes (class paths resolved
through `diffuser.`), pickled and unpickled as the evaluator worker does (diffuser/utils/evaluator.py:382-395), the
"Testing forward" smoke step (train.py:200-204), trainer.train(n), save, a second trainer rebuilt from the pickles that
loads the checkpoint, and ema_model.conditional_sample on conditions shaped like the evaluator's.

The dataset is synthetic (the simulators / .npy episodes are not in the image): a torch Dataset with the constructor
keywords train.py:41-72 passes and the item contract of diffuser/datasets/sequence.py:271-353 (numpy arrays; `loss_masks`
float64; `returns` float32 [1, A]; conditioning masks by the MultiAgentMaskGenerator rule).  The spec values are the
constants of exp_specs/mpe/spread/mad_mpe_spread_attn_exp.yaml with a narrower network (dim 32) and fewer steps so the
test stays short; ml_logger / params_proto are absent, so `logger.root` is a temporary directory."""
import os
import pickle

import numpy as np
import pytest
import torch

from oracle import madiff_oracle as O

SPEC = dict(  # exp_specs/mpe/spread/mad_mpe_spread_attn_exp.yaml `constants` (+ the swept `variables` at their first value)
    seed=100, env_type="mpe", n_agents=3, use_action=True, discrete_action=False, residual_attn=True,
    decentralized_execution=False, use_zero_padding=False, pred_future_padding=True, use_ddim_sample=False, n_ddim_steps=15,
    model="models.SharedConvAttentionDeconv", diffusion="models.GaussianDiffusion", share_inv=True, history_horizon=0,
    n_diffusion_steps=200, action_weight=10, loss_weights=None, loss_discount=1, dim_mults=[1, 4, 8], returns_condition=True,
    predict_epsilon=True, calc_energy=False, dim=128, hidden_dim=256, condition_dropout=0.25, condition_guidance_w=1.2,
    train_only_inv=False, clip_denoised=True, test_ret=0.9, renderer="utils.MPERenderer", loader="datasets.SequenceDataset",
    normalizer="CDFNormalizer", max_n_episodes=200000, preprocess_fns=[], use_padding=True, discount=0.99, max_path_length=25,
    termination_penalty=0.0, n_steps_per_epoch=10000, n_train_steps=1000000, batch_size=32, learning_rate=0.0002,
    gradient_accumulate_every=2, ema_decay=0.995, log_freq=1000, save_freq=50000, sample_freq=10000, n_saves=5,
    save_parallel=False, n_reference=3, save_checkpoints=True, evaluator="utils.MADEvaluator", num_eval=10, eval_freq=100000,
    continue_training=True, horizon=24, returns_scale=700, dataset="simple_spread-expert",
)


class _Env:
    num_actions = 0


class SyntheticSequenceDataset(torch.utils.data.Dataset):
    """Constructor keywords = train.py:41-72; items = sequence.py:326-353."""

    def __init__(self, env_type, env, n_agents, horizon, history_horizon, normalizer, preprocess_fns, max_n_episodes,
                 use_padding, use_action, discrete_action, max_path_length, include_returns, include_env_ts, returns_scale,
                 discount, termination_penalty, agent_share_parameters, use_seed_dataset, seed, use_inv_dyn,
                 decentralized_execution, use_zero_padding, agent_condition_type, pred_future_padding,
                 observation_dim=18, action_dim=5, n_items=256):
        assert agent_share_parameters is True and use_inv_dyn and use_action and not discrete_action
        self.n_agents, self.horizon, self.history_horizon = n_agents, horizon, history_horizon
        self.observation_dim, self.action_dim = observation_dim, action_dim
        self.include_returns, self.include_env_ts, self.max_path_length = include_returns, include_env_ts, max_path_length
        self.agent_condition_type = agent_condition_type
        self.env = _Env()
        rng = np.random.default_rng(seed)
        T = horizon + history_horizon
        # smooth trajectories in [-1, 1] (normalised data lives there), so a few optimizer steps visibly reduce the loss
        phase = rng.uniform(0, 2 * np.pi, size=(n_items, 1, n_agents, action_dim + observation_dim))
        freq = rng.uniform(0.1, 0.6, size=(n_items, 1, n_agents, action_dim + observation_dim))
        self.traj = np.sin(phase + freq * np.arange(T)[None, :, None, None]).astype(np.float32)
        self.ret = rng.uniform(0, 1, size=(n_items, 1, n_agents)).astype(np.float32)

    def __len__(self):
        return len(self.traj) * (self.n_agents if self.agent_condition_type == "single" else 1)

    def __getitem__(self, idx):
        hh, A = self.history_horizon, self.n_agents
        if self.agent_condition_type == "single":
            agent_mask = np.zeros(A, dtype=bool)
            agent_mask[idx % A] = True
            idx //= A
        else:
            agent_mask = np.ones(A, dtype=bool)
        traj = self.traj[idx]
        obs = traj[..., self.action_dim:]
        cond_x = obs.copy()
        cond_x[:hh, ~agent_mask] = 0.0
        masks = O.agent_condition_mask((1,) + obs.shape, agent_mask, hh)[0]
        loss_masks = np.zeros((obs.shape[0], A, 1))            # float64, as numpy makes it (sequence.py:314)
        loss_masks[hh:] = 1.0
        loss_masks[hh, agent_mask] = 0.0
        attention_masks = np.ones((obs.shape[0], A, 1))
        batch = {"x": traj, "cond": {"x": cond_x, "masks": masks}, "loss_masks": loss_masks,
                 "attention_masks": attention_masks}
        if self.include_returns:
            batch["returns"] = self.ret[idx]
        if self.include_env_ts:
            batch["env_ts"] = np.arange(-hh, self.horizon)
        return batch


class NullRenderer:
    def __init__(self, env_type, env):
        self.env_type, self.env = env_type, env


def _configs(C, tmp, dim):
    """run_scripts/train.py:17-159 with `Config` = the spec object C; the loader / renderer are handed over as classes
    (utils.Config takes either, config.py:8-9) because the reference's datasets / renderers are not in this image."""
    import diffuser.utils as utils
    C.discrete_action = getattr(C, "discrete_action", False)
    C.state_loss_weight = getattr(C, "state_loss_weight", None)
    C.opponent_loss_weight = getattr(C, "opponent_loss_weight", None)
    C.use_seed_dataset = getattr(C, "use_seed_dataset", False)
    C.residual_attn = getattr(C, "residual_attn", True)
    C.use_temporal_attention = getattr(C, "use_temporal_attention", True)
    C.env_ts_condition = getattr(C, "env_ts_condition", False)
    C.use_return_to_go = getattr(C, "use_return_to_go", False)
    C.joint_inv = getattr(C, "joint_inv", False)
    C.use_zero_padding = getattr(C, "use_zero_padding", True)
    C.use_inv_dyn = getattr(C, "use_inv_dyn", True)
    C.pred_future_padding = getattr(C, "pred_future_padding", False)
    if not hasattr(C, "agent_condition_type"):
        C.agent_condition_type = "single" if C.decentralized_execution else "all"
    dataset_config = utils.Config(
        SyntheticSequenceDataset, savepath="dataset_config.pkl", env_type=C.env_type, env=C.dataset, n_agents=C.n_agents,
        horizon=C.horizon, history_horizon=C.history_horizon, normalizer=C.normalizer, preprocess_fns=C.preprocess_fns,
        max_n_episodes=C.max_n_episodes, use_padding=C.use_padding, use_action=C.use_action,
        discrete_action=C.discrete_action, max_path_length=C.max_path_length, include_returns=C.returns_condition,
        include_env_ts=C.env_ts_condition, returns_scale=C.returns_scale, discount=C.discount,
        termination_penalty=C.termination_penalty,
        agent_share_parameters=utils.import_class(C.model).agent_share_parameters,
        use_seed_dataset=C.use_seed_dataset, seed=C.seed, use_inv_dyn=C.use_inv_dyn,
        decentralized_execution=C.decentralized_execution, use_zero_padding=C.use_zero_padding,
        agent_condition_type=C.agent_condition_type, pred_future_padding=C.pred_future_padding)
    render_config = utils.Config(NullRenderer, savepath="render_config.pkl", env_type=C.env_type, env=C.dataset)
    data_encoder_config = utils.Config(getattr(C, "data_encoder", "utils.IdentityEncoder"), savepath="data_encoder_config.pkl")
    dataset = dataset_config()
    data_encoder = data_encoder_config()
    observation_dim, action_dim = dataset.observation_dim, dataset.action_dim
    model_config = utils.Config(
        C.model, savepath="model_config.pkl", n_agents=C.n_agents, horizon=C.horizon + C.history_horizon,
        history_horizon=C.history_horizon, transition_dim=observation_dim, dim_mults=C.dim_mults,
        returns_condition=C.returns_condition, env_ts_condition=C.env_ts_condition, dim=dim,
        condition_dropout=C.condition_dropout, residual_attn=C.residual_attn, max_path_length=C.max_path_length,
        use_temporal_attention=C.use_temporal_attention, device=C.device)
    diffusion_config = utils.Config(
        C.diffusion, savepath="diffusion_config.pkl", n_agents=C.n_agents, horizon=C.horizon,
        history_horizon=C.history_horizon, observation_dim=observation_dim, action_dim=action_dim,
        discrete_action=C.discrete_action, num_actions=getattr(dataset.env, "num_actions", 0),
        n_timesteps=C.n_diffusion_steps, clip_denoised=C.clip_denoised, predict_epsilon=C.predict_epsilon,
        hidden_dim=C.hidden_dim, train_only_inv=C.train_only_inv, share_inv=C.share_inv, joint_inv=C.joint_inv,
        action_weight=C.action_weight, loss_weights=C.loss_weights, state_loss_weight=C.state_loss_weight,
        opponent_loss_weight=C.opponent_loss_weight, loss_discount=C.loss_discount, returns_condition=C.returns_condition,
        condition_guidance_w=C.condition_guidance_w, data_encoder=data_encoder, use_inv_dyn=C.use_inv_dyn, device=C.device)
    trainer_config = utils.Config(
        utils.Trainer, savepath="trainer_config.pkl", train_batch_size=C.batch_size, train_lr=C.learning_rate,
        gradient_accumulate_every=C.gradient_accumulate_every, ema_decay=C.ema_decay, sample_freq=C.sample_freq,
        save_freq=C.save_freq, log_freq=C.log_freq, label_freq=int(C.n_train_steps // C.n_saves), eval_freq=C.eval_freq,
        save_parallel=C.save_parallel, bucket=str(tmp), n_reference=C.n_reference, train_device=C.device,
        save_checkpoints=C.save_checkpoints)
    return dict(dataset=dataset_config, render=render_config, model=model_config, diffusion=diffusion_config,
                trainer=trainer_config), dataset


def test_spec_values_resolve_through_the_plugin_path():
    """CPU: the class-path strings of the YAML resolve to the build's classes and the class attribute train.py:61-63
    reads before construction exists."""
    import diffuser.utils as utils
    import madiff_amd.diffusion
    import madiff_amd.modules
    assert utils.import_class(SPEC["model"]) is madiff_amd.modules.SharedConvAttentionDeconv
    assert utils.import_class(SPEC["diffusion"]) is madiff_amd.diffusion.GaussianDiffusion
    assert utils.import_class(SPEC["model"]).agent_share_parameters is True
    assert utils.import_class("utils.IdentityEncoder") is madiff_amd.diffusion.IdentityEncoder
    ds = SyntheticSequenceDataset("mpe", "x", 3, 24, 0, "CDFNormalizer", [], 10, True, True, False, 25, True, False, 700, 0.99,
                                  0.0, True, False, 1, True, False, False, "all", True, n_items=4)
    item = ds[1]
    assert item["x"].shape == (24, 3, 23) and item["x"].dtype == np.float32
    assert item["loss_masks"].dtype == np.float64 and item["loss_masks"].shape == (24, 3, 1)
    assert item["cond"]["masks"].dtype == bool and item["returns"].shape == (1, 3) and item["returns"].dtype == np.float32


@pytest.mark.gpu
def test_train_script_flow_on_the_gpu(tmp_path):
    import diffuser.utils as utils
    from madiff_amd import ops
    ops.set_precision("bf16")
    dev = "cuda:0"
    C = utils.build_config_from_dict({**SPEC, "device": dev, "save_freq": 4, "eval_freq": 0, "log_freq": 2, "sample_freq": 0,
                                      "n_train_steps": 6, "n_steps_per_epoch": 6})
    utils.set_seed(C.seed)
    configs, dataset = _configs(C, tmp_path, dim=32)
    # the evaluator worker rebuilds everything from these pickles (evaluator.py:382-395)
    blobs = {k: pickle.dumps(v) for k, v in configs.items()}
    model = configs["model"]()
    diffusion = configs["diffusion"](model)
    trainer = configs["trainer"](diffusion, dataset, configs["render"]())
    assert next(diffusion.parameters()).is_cuda
    utils.report_parameters(model)
    batch = utils.batchify(dataset[0], dev)                    # train.py:200-204 "Testing forward"
    loss, _ = diffusion.loss(**batch)
    loss.backward()
    assert bool(torch.isfinite(loss))
    diffusion.zero_grad(set_to_none=True)
    try:
        n_epochs = int((C.n_train_steps - trainer.step) // C.n_steps_per_epoch)
        for _ in range(n_epochs):
            trainer.train(n_train_steps=C.n_steps_per_epoch)
        trainer.finish_training()
        assert trainer.step == 6 and trainer._graphed is not None          # steps 3.. were replayed from the hipGraph
        torch.cuda.synchronize()
        ckpt = utils.discover_latest_checkpoint_path(os.path.join(str(tmp_path), "checkpoint"))
        assert ckpt is not None and ckpt.endswith("state_4.pt")
        trained = {k: v.detach().clone() for k, v in trainer.model.state_dict().items()}
    finally:
        ops.SINK = None
        ops.clear_pack_cache()
    # ---- a second process' view: everything from the pickles + the checkpoint (train.py:180-192 / evaluator.py:143-167)
    c2 = {k: pickle.loads(b) for k, b in blobs.items()}
    assert c2["model"]._class is configs["model"]._class and c2["model"]._dict == configs["model"]._dict
    assert c2["trainer"]._dict == configs["trainer"]._dict and c2["diffusion"]["n_timesteps"] == C.n_diffusion_steps
    ds2 = c2["dataset"]()
    dif2 = c2["diffusion"](c2["model"]())
    try:
        tr2 = c2["trainer"](dif2, ds2, c2["render"]())
        state = torch.load(ckpt, map_location=dev, weights_only=True)
        assert set(state.keys()) == {"step", "model", "ema"} and state["step"] == 4
        tr2.step = state["step"]
        tr2.model.load_state_dict(state["model"])
        tr2.ema_model.load_state_dict(state["ema"])
        ops.params_changed()
        assert set(state["model"].keys()) == set(trained.keys())
        moved = max(float((state["model"][k].float() - v.float()).abs().max()) for k, v in trained.items()
                    if v.dtype.is_floating_point)
        assert 0.0 < moved < 10 * C.learning_rate                            # one more Adam step after the checkpoint
        item = utils.batchify(ds2[3], dev)
        cond = {"x": item["cond"]["x"], "masks": item["cond"]["masks"].bool()}
        x = tr2.ema_model.conditional_sample(cond, returns=item["returns"], verbose=False)
        assert x.shape == (1, C.horizon + C.history_horizon, C.n_agents, ds2.observation_dim) and bool(torch.isfinite(x).all())
        m = cond["masks"]
        assert torch.equal(x[m], cond["x"][m])                              # first-observation inpainting enforced
        tr2.train(n_train_steps=2)                                           # resumed training steps
        assert tr2.step == 6
    finally:
        ops.SINK = None
        ops.clear_pack_cache()
