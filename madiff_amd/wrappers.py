"""The ablation denoisers without cross-agent attention (diffuser/models/ma_temporal_wrappers.py:10-249): one TemporalUnet
over the concatenated agents, one per agent, or one shared across agents.  Same constructor arguments, attributes,
``state_dict`` keys and call surface; everything runs on the same HIP kernels as SharedConvAttentionDeconv (the U-Net is
``TemporalUnet.forward_cl``).  No exp_spec selects them.

Additions for parity tests only: ``dropout_mask`` (the explicit Bernoulli draws of the ``use_dropout`` branch)."""
from __future__ import annotations

from typing import Tuple

import torch
from torch import nn

from .modules import TemporalUnet


def _unet(horizon, history_horizon, transition_dim, dim, dim_mults, returns_condition, env_ts_condition, condition_dropout,
          kernel_size, max_path_length):
    return TemporalUnet(horizon=horizon, history_horizon=history_horizon, transition_dim=transition_dim, dim=dim,
                        dim_mults=dim_mults, returns_condition=returns_condition, env_ts_condition=env_ts_condition,
                        condition_dropout=condition_dropout, kernel_size=kernel_size, max_path_length=max_path_length)


class ConcatenatedTemporalUnet(nn.Module):
    """ma_temporal_wrappers.py:10-79: agents concatenated along the feature axis, returns averaged over agents."""

    agent_share_parameters = False

    def __init__(self, n_agents: int, horizon: int, transition_dim: int, dim: int = 128, history_horizon: int = 0,
                 dim_mults: Tuple[int] = (1, 2, 4, 8), returns_condition: bool = False, env_ts_condition: bool = False,
                 condition_dropout: float = 0.1, kernel_size: int = 5, residual_attn: bool = False,
                 use_layer_norm: bool = False, max_path_length: int = 100, use_temporal_attention: bool = False):
        super().__init__()
        self.n_agents = n_agents
        self.history_horizon = history_horizon
        self.use_temporal_attention = use_temporal_attention
        self.returns_condition = returns_condition
        self.env_ts_condition = env_ts_condition
        self.net = _unet(horizon, history_horizon, transition_dim * n_agents, dim, dim_mults, returns_condition,
                         env_ts_condition, condition_dropout, kernel_size, max_path_length)

    def forward(self, x, time, returns=None, env_timestep=None, attention_masks=None, use_dropout: bool = True,
                force_dropout: bool = False, dropout_mask=None, **kwargs):
        """x : [ batch x horizon x agent x transition ];  returns : [ batch x 1 x agent ]."""
        assert x.shape[2] == self.n_agents, f"{x.shape}, {self.n_agents}"
        b, h, a, f = x.shape
        y = self.net(x.reshape(b, h, a * f), time=time, returns=returns.mean(dim=2) if returns is not None else None,
                     env_timestep=env_timestep, use_dropout=use_dropout, force_dropout=force_dropout,
                     dropout_mask=dropout_mask)
        return y.reshape(b, h, a, f)


class IndependentTemporalUnet(nn.Module):
    """ma_temporal_wrappers.py:82-158: one U-Net per agent (keys nets.{a}.*)."""

    agent_share_parameters = False

    def __init__(self, n_agents: int, horizon: int, transition_dim: int, dim: int = 128, history_horizon: int = 0,
                 dim_mults: Tuple[int] = (1, 2, 4, 8), returns_condition: bool = False, env_ts_condition: bool = False,
                 condition_dropout: float = 0.1, kernel_size: int = 5, residual_attn: bool = False,
                 max_path_length: int = 100, use_temporal_attention: bool = False):
        super().__init__()
        self.n_agents = n_agents
        self.history_horizon = history_horizon
        self.use_temporal_attention = use_temporal_attention
        self.returns_condition = returns_condition
        self.env_ts_condition = env_ts_condition
        self.nets = nn.ModuleList([_unet(horizon, history_horizon, transition_dim, dim, dim_mults, returns_condition,
                                         env_ts_condition, condition_dropout, kernel_size, max_path_length)
                                   for _ in range(n_agents)])

    def forward(self, x, time, returns=None, env_timestep=None, attention_masks=None, use_dropout: bool = True,
                force_dropout: bool = False, dropout_mask=None, **kwargs):
        """dropout_mask: one [batch, 1] mask per agent (list), in the order the reference draws them (agent 0 first)."""
        assert x.shape[2] == self.n_agents, f"{x.shape}, {self.n_agents}"
        outs = []
        for a_idx in range(self.n_agents):
            outs.append(self.nets[a_idx](x[:, :, a_idx, :], time=time,
                                         returns=returns[:, :, a_idx] if returns is not None else None,
                                         env_timestep=env_timestep, use_dropout=use_dropout, force_dropout=force_dropout,
                                         dropout_mask=None if dropout_mask is None else dropout_mask[a_idx]))
        return torch.stack(outs, dim=2)


class SharedIndependentTemporalUnet(nn.Module):
    """ma_temporal_wrappers.py:161-234: ONE U-Net applied to every agent's sequence.  The reference flattens the sequences
    as (b a) but stacks time / returns / env_timestep agent-major (`torch.cat([time for _ in range(a)])`, :214-227): row
    i = b*A + a of the batch is conditioned on entry i of the agent-major lists.  Reproduced as is."""

    agent_share_parameters = True

    def __init__(self, n_agents: int, horizon: int, history_horizon: int, transition_dim: int, dim: int = 128,
                 dim_mults: Tuple[int] = (1, 2, 4, 8), returns_condition: bool = False, env_ts_condition: bool = False,
                 condition_dropout: float = 0.1, kernel_size: int = 5, residual_attn: bool = False,
                 max_path_length: int = 100):
        super().__init__()
        self.n_agents = n_agents
        self.returns_condition = returns_condition
        self.env_ts_condition = env_ts_condition
        self.history_horizon = history_horizon
        self.net = _unet(horizon, history_horizon, transition_dim, dim, dim_mults, returns_condition, env_ts_condition,
                         condition_dropout, kernel_size, max_path_length)

    def forward(self, x, time, returns=None, env_timestep=None, attention_masks=None, use_dropout=True,
                force_dropout=False, dropout_mask=None, **kwargs):
        """dropout_mask: [batch * agents, 1]."""
        assert x.shape[2] == self.n_agents, f"{x.shape}, {self.n_agents}"
        b, t, a, f = x.shape
        xs = x.permute(0, 2, 1, 3).reshape(b * a, t, f)
        y = self.net(xs, time=torch.cat([time for _ in range(a)], dim=0),
                     returns=torch.cat([returns[:, :, i] for i in range(a)], dim=0) if returns is not None else None,
                     env_timestep=torch.cat([env_timestep for _ in range(a)], dim=0) if env_timestep is not None else None,
                     use_dropout=use_dropout, force_dropout=force_dropout, dropout_mask=dropout_mask)
        return y.reshape(b, a, t, f).permute(0, 2, 1, 3)
