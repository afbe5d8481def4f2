"""Host-side mirror of the reference's ``diffuser.models`` denoiser surface, executing on the HIP kernels.

Same class names, constructor arguments, attributes, ``state_dict`` keys and parameter shapes as
/root/reference/diffuser/models/{helpers,temporal,ma_temporal}.py (cited per class), so reference checkpoints
load unchanged and ``utils.Config`` / ``copy.deepcopy`` / ``Adam(model.parameters())`` keep working.  The
nn.Conv1d / nn.Linear / nn.GroupNorm children are parameter holders only: their own ``forward`` is never used
on the hot path; every contraction, normalisation and the attention core run through ``madiff_amd.ops``.

Internally activations are channels-last ("cl") matrices [(sequence, time), pad32(channels)].
"""
from __future__ import annotations

import os

import math
from typing import List, Optional, Tuple

import torch
import torch.nn as nn
from torch.distributions import Bernoulli

from . import ops
from .ops import CONV1, CONV5, DOWN, LINEAR, UP, AttnBlock, AttnCore, GNMish, PackTraj, conv_gemm, mish, pad32


# ------------------------------------------------------------------------------------------------
# layout glue for the module-level ([N, C, T]) interfaces
# ------------------------------------------------------------------------------------------------
def to_cl(x: torch.Tensor) -> torch.Tensor:
    """[N, C, T] -> cl [N*T, pad32(C)] (differentiable torch glue; only used by the per-module entry points)."""
    n, c, t = x.shape
    y = x.permute(0, 2, 1).reshape(n * t, c)
    if pad32(c) != c:
        y = torch.nn.functional.pad(y, (0, pad32(c) - c))
    return y.to(ops.act_dtype()).contiguous()


def from_cl(y: torch.Tensor, n: int, t: int, c: int) -> torch.Tensor:
    return y[:, :c].reshape(n, t, c).permute(0, 2, 1).float()


def rows_to_cl(x: torch.Tensor) -> torch.Tensor:
    """[rows, C] fp32 -> cl [rows, pad32(C)]."""
    c = x.shape[-1]
    if pad32(c) != c:
        x = torch.nn.functional.pad(x, (0, pad32(c) - c))
    return x.to(ops.act_dtype()).contiguous()


class _Tag(nn.Identity):
    """Parameter-free placeholder keeping the reference's nn.Sequential indices (Rearrange / Mish slots)."""


# ------------------------------------------------------------------------------------------------
# leaves (helpers.py)
# ------------------------------------------------------------------------------------------------
class SinusoidalPosEmb(nn.Module):
    """helpers.py:17-29."""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def freq(self, device) -> torch.Tensor:
        """exp(-j ln(1e4)/(half-1)) evaluated on the host in fp32 like the reference, cached per device (no H2D copy
        inside a captured graph)."""
        cache = self.__dict__.setdefault("_freq_cache", {})
        key = str(device)
        if key not in cache:
            half = self.dim // 2
            cache[key] = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1))).to(device)
        return cache[key]

    def __deepcopy__(self, memo):
        return SinusoidalPosEmb(self.dim)

    def forward(self, x):
        shape = x.shape
        ts = x.reshape(-1).to(torch.int64).contiguous()
        out = ops.sinusoid(ts, self.freq(x.device), 1)
        return out[:, : self.dim].float().reshape(*shape, self.dim)


class Downsample1d(nn.Module):
    """helpers.py:32-38: Conv1d(dim, dim, 3, stride 2, pad 1)."""

    def __init__(self, dim):
        super().__init__()
        self.conv = nn.Conv1d(dim, dim, 3, 2, 1)

    def forward_cl(self, x, t):
        return conv_gemm(x, [self.conv.weight], [self.conv.bias], DOWN, t), DOWN.out_len(t)

    def forward(self, x):
        n, c, t = x.shape
        y, t2 = self.forward_cl(to_cl(x), t)
        return from_cl(y, n, t2, c)


class Upsample1d(nn.Module):
    """helpers.py:41-47: ConvTranspose1d(dim, dim, 4, stride 2, pad 1)."""

    def __init__(self, dim):
        super().__init__()
        self.conv = nn.ConvTranspose1d(dim, dim, 4, 2, 1)

    def forward_cl(self, x, t):
        return conv_gemm(x, [self.conv.weight], [self.conv.bias], UP, t), UP.out_len(t)

    def forward(self, x):
        n, c, t = x.shape
        y, t2 = self.forward_cl(to_cl(x), t)
        return from_cl(y, n, t2, c)


class Conv1dBlock(nn.Module):
    """helpers.py:50-74: Conv1d -> GroupNorm(8) -> Mish.  keys: block.0.{weight,bias}, block.2.{weight,bias}."""

    def __init__(self, inp_channels, out_channels, kernel_size, mish=True, n_groups=8):
        super().__init__()
        if not mish:
            raise NotImplementedError("the reference always builds Mish blocks (temporal.py:225)")
        self.block = nn.Sequential(
            nn.Conv1d(inp_channels, out_channels, kernel_size, padding=kernel_size // 2),
            _Tag(), nn.GroupNorm(n_groups, out_channels), _Tag(), _Tag())
        self.out_channels = out_channels
        self.n_groups = n_groups
        self.spec = CONV5 if kernel_size == 5 else ops.ConvSpec("conv", kernel_size, 1, kernel_size // 2, 0)

    def forward_cl(self, x, t, x2=None, rowbias=None, res=None):
        conv, gn = self.block[0], self.block[2]
        return ops.conv_block(x, conv.weight, conv.bias, gn.weight, gn.bias, self.spec, t, self.n_groups, x2=x2,
                              rowbias=rowbias, res=res)

    def forward(self, x):
        n, _, t = x.shape
        return from_cl(self.forward_cl(to_cl(x), t), n, t, self.out_channels)


# ------------------------------------------------------------------------------------------------
# temporal.py
# ------------------------------------------------------------------------------------------------
PROJ_SPLIT = os.environ.get("MADIFF_PROJ_SPLIT", "1") != "0"
PROJ_SPLIT_MIN_SEQ = int(os.environ.get("MADIFF_PROJ_SPLIT_MIN_SEQ", "512"))
EMBED_FUSED = os.environ.get("MADIFF_EMBED_FUSED", "1") != "0"
RES_BLOCK_NODE = os.environ.get("MADIFF_RES_BLOCK", "1") != "0"
ATTN_BLOCK_NODE = os.environ.get("MADIFF_ATTN_BLOCK", "1") != "0"


def _chain_tag():
    """A fresh ops.ChainTag for a pair of consecutive ResidualTemporalBlocks (None when the chained backward is off)."""
    return ops.ChainTag() if (ops.FUSE_GNB_CHAIN and RES_BLOCK_NODE and torch.is_grad_enabled()) else None


class ResidualTemporalBlock(nn.Module):
    """temporal.py:161-200.  keys: blocks.{0,1}.block.*, time_mlp.1.*, residual_conv.*"""

    def __init__(self, inp_channels, out_channels, embed_dim, kernel_size=5, mish=True):
        super().__init__()
        self.blocks = nn.ModuleList([Conv1dBlock(inp_channels, out_channels, kernel_size, mish),
                                     Conv1dBlock(out_channels, out_channels, kernel_size, mish)])
        self.time_mlp = nn.Sequential(_Tag(), nn.Linear(embed_dim, out_channels), _Tag())
        self.residual_conv = nn.Conv1d(inp_channels, out_channels, 1) if inp_channels != out_channels else nn.Identity()
        self.out_channels = out_channels

    def forward_cl(self, x, t, temb_proj, x2=None, chain_out=None, chain_in=None):
        """x cl [N*t, Cin]; temb_proj fp32 [N, >=Cout] = Linear(Mish(t_emb)) (computed by the fused time GEMM).
        chain_out / chain_in: the ops.ChainTag two consecutive blocks share when this block's output feeds the next block
        only / this block's input is the previous block's output and nobody else's (the caller guarantees it)."""
        b0, b1 = self.blocks
        if RES_BLOCK_NODE and b0.spec == CONV5 and b1.spec == CONV5:        # the whole block as one autograd node
            rc = self.residual_conv if isinstance(self.residual_conv, nn.Conv1d) else None
            return ops.res_block(x, x2, temb_proj, t, b0.n_groups,
                                 (b0.block[0].weight, b0.block[0].bias, b0.block[2].weight, b0.block[2].bias),
                                 (b1.block[0].weight, b1.block[0].bias, b1.block[2].weight, b1.block[2].bias),
                                 None if rc is None else rc.weight, None if rc is None else rc.bias,
                                 chain_out=chain_out, chain_in=chain_in)
        h = self.blocks[0].forward_cl(x, t, x2=x2, rowbias=temb_proj)
        if isinstance(self.residual_conv, nn.Conv1d):
            res = conv_gemm(x, [self.residual_conv.weight], [self.residual_conv.bias], CONV1, t, x2=x2)
        else:
            assert x2 is None
            res = x
        return self.blocks[1].forward_cl(h, t, res=res)

    def forward(self, x, t):
        n, _, tl = x.shape
        lin = self.time_mlp[1]
        proj = conv_gemm(mish(rows_to_cl(t)), [lin.weight], [lin.bias], LINEAR, 1, perm="f32")
        return from_cl(self.forward_cl(to_cl(x), tl, proj[:, :self.out_channels]), n, tl, self.out_channels)


class TemporalSelfAttention(nn.Module):
    """temporal.py:57-124: cross-agent attention with timestep-conditioned Q/K/V."""

    def __init__(self, n_channels, qk_n_channels, v_n_channels, embed_dim, nheads=4, residual=False):
        super().__init__()
        self.nheads = nheads
        self.query_layer = nn.Conv1d(n_channels, qk_n_channels * nheads, kernel_size=1)
        self.key_layer = nn.Conv1d(n_channels, qk_n_channels * nheads, kernel_size=1)
        self.value_layer = nn.Conv1d(n_channels, v_n_channels * nheads, kernel_size=1)
        self.query_time_mlp = nn.Sequential(_Tag(), nn.Linear(embed_dim, qk_n_channels * nheads), _Tag())
        self.key_time_mlp = nn.Sequential(_Tag(), nn.Linear(embed_dim, qk_n_channels * nheads), _Tag())
        self.value_time_mlp = nn.Sequential(_Tag(), nn.Linear(embed_dim, v_n_channels * nheads), _Tag())
        self.residual = residual
        if residual:
            self.gamma = nn.Parameter(torch.zeros([1]))
        self.n_channels = n_channels
        self.cq = qk_n_channels * nheads
        self.cv = v_n_channels * nheads
        if self.cv != n_channels:
            raise NotImplementedError("value width must equal the channel count (out.reshape(x.shape), temporal.py:121)")

    def time_layers(self) -> List[nn.Linear]:
        return [self.query_time_mlp[1], self.key_time_mlp[1], self.value_time_mlp[1]]

    def forward_cl(self, x, bsz, n_agents, t, time_proj):
        """x cl [(b a) t, C]; time_proj fp32 [B*A, >= 2cq + C] = [q | k | v] time-MLP outputs."""
        if ATTN_BLOCK_NODE:          # projection + attention core as one autograd node (ops.AttnBlock)
            return AttnBlock.apply(x, time_proj, self.gamma if self.residual else None, bsz, n_agents, t, self.nheads,
                                   self.cq, self.n_channels, self.residual, self.query_layer.weight, self.key_layer.weight,
                                   self.value_layer.weight, self.query_layer.bias, self.key_layer.bias, self.value_layer.bias)
        qkv = conv_gemm(x, [self.query_layer.weight, self.key_layer.weight, self.value_layer.weight],
                        [self.query_layer.bias, self.key_layer.bias, self.value_layer.bias], CONV1, t,
                        rowbias=time_proj)
        return AttnCore.apply(qkv, x, self.gamma if self.residual else None, bsz, n_agents, t, self.nheads, self.cq,
                              self.n_channels, self.residual)

    def forward(self, x, time):
        b, a, c, t = x.shape
        tl = self.time_layers()
        proj = conv_gemm(mish(rows_to_cl(time.reshape(b * a, -1))), [l.weight for l in tl], [l.bias for l in tl],
                         LINEAR, 1, perm="f32")
        y = self.forward_cl(to_cl(x.reshape(b * a, c, t)), b, a, t, proj[:, :2 * self.cq + self.cv])
        return from_cl(y, b * a, t, c).reshape(b, a, c, t)


class SelfAttention(nn.Module):
    """helpers.py:77-124: the cross-agent attention without timestep conditioning (same fused Q/K/V GEMM and attention
    kernel as TemporalSelfAttention, no time row-bias).  In the reference only ConvAttentionDeconv and the value networks
    build it -- classes whose constructors raise -- so it is provided as a module."""

    def __init__(self, n_channels: int, qk_n_channels: int, v_n_channels: int, nheads: int = 4, residual: bool = False):
        super().__init__()
        self.nheads = nheads
        self.query_layer = nn.Conv1d(n_channels, qk_n_channels * nheads, kernel_size=1)
        self.key_layer = nn.Conv1d(n_channels, qk_n_channels * nheads, kernel_size=1)
        self.value_layer = nn.Conv1d(n_channels, v_n_channels * nheads, kernel_size=1)
        self.attend = _Tag()
        self.residual = residual
        if residual:
            self.gamma = nn.Parameter(torch.zeros([1]))
        self.n_channels = n_channels
        self.cq = qk_n_channels * nheads
        self.cv = v_n_channels * nheads
        if self.cv != n_channels:
            raise NotImplementedError("value width must equal the channel count (out.reshape(x.shape), helpers.py:121)")

    def forward_cl(self, x, bsz, n_agents, t):
        if ATTN_BLOCK_NODE:
            return AttnBlock.apply(x, None, self.gamma if self.residual else None, bsz, n_agents, t, self.nheads,
                                   self.cq, self.n_channels, self.residual, self.query_layer.weight, self.key_layer.weight,
                                   self.value_layer.weight, self.query_layer.bias, self.key_layer.bias, self.value_layer.bias)
        qkv = conv_gemm(x, [self.query_layer.weight, self.key_layer.weight, self.value_layer.weight],
                        [self.query_layer.bias, self.key_layer.bias, self.value_layer.bias], CONV1, t)
        return AttnCore.apply(qkv, x, self.gamma if self.residual else None, bsz, n_agents, t, self.nheads, self.cq,
                              self.n_channels, self.residual)

    def forward(self, x, states=None):
        b, a, c, t = x.shape
        y = self.forward_cl(to_cl(x.reshape(b * a, c, t)), b, a, t)
        return from_cl(y, b * a, t, c).reshape(b, a, c, t)


class TemporalUnet(nn.Module):
    """temporal.py:203-346 (constructor = parameter layout).  Its forward is only used for a single agent
    group; SharedConvAttentionDeconv drives the blocks itself."""

    agent_share_parameters = True

    def __init__(self, horizon, transition_dim, history_horizon=0, dim=128, dim_mults=(1, 2, 4, 8),
                 returns_condition=False, env_ts_condition=False, condition_dropout=0.1, kernel_size=5,
                 max_path_length=100):
        super().__init__()
        dims = [transition_dim, *map(lambda m: dim * m, dim_mults)]
        in_out = list(zip(dims[:-1], dims[1:]))
        self.in_out = in_out
        self.time_dim = dim
        self.returns_dim = dim
        self.transition_dim = transition_dim
        self.time_mlp = nn.Sequential(SinusoidalPosEmb(dim), nn.Linear(dim, dim * 4), _Tag(), nn.Linear(dim * 4, dim))
        embed_dim = dim
        self.returns_condition = returns_condition
        self.env_ts_condition = env_ts_condition
        self.condition_dropout = condition_dropout
        self.history_horizon = history_horizon
        if returns_condition:
            self.returns_mlp = nn.Sequential(nn.Linear(1, dim), _Tag(), nn.Linear(dim, dim * 4), _Tag(),
                                             nn.Linear(dim * 4, dim))
            self.mask_dist = Bernoulli(probs=1 - self.condition_dropout)
            embed_dim += dim
        if env_ts_condition:
            self.env_ts_mlp = nn.Sequential(nn.Embedding(max_path_length + 1, dim), nn.Linear(dim, dim * 4), _Tag(),
                                            nn.Linear(dim * 4, dim))
            embed_dim += dim
        self.embed_dim = embed_dim

        self.downs = nn.ModuleList([])
        self.ups = nn.ModuleList([])
        n_res = len(in_out)
        for ind, (d_in, d_out) in enumerate(in_out):
            is_last = ind >= (n_res - 1)
            self.downs.append(nn.ModuleList([
                ResidualTemporalBlock(d_in, d_out, embed_dim=embed_dim, kernel_size=kernel_size),
                ResidualTemporalBlock(d_out, d_out, embed_dim=embed_dim, kernel_size=kernel_size),
                Downsample1d(d_out) if not is_last else nn.Identity()]))
        mid = dims[-1]
        self.mid_block1 = ResidualTemporalBlock(mid, mid, embed_dim=embed_dim, kernel_size=kernel_size)
        self.mid_block2 = ResidualTemporalBlock(mid, mid, embed_dim=embed_dim, kernel_size=kernel_size)
        for ind, (d_in, d_out) in enumerate(reversed(in_out[1:])):
            is_last = ind >= (n_res - 1)
            self.ups.append(nn.ModuleList([
                ResidualTemporalBlock(d_out * 2, d_in, embed_dim=embed_dim, kernel_size=kernel_size),
                ResidualTemporalBlock(d_in, d_in, embed_dim=embed_dim, kernel_size=kernel_size),
                Upsample1d(d_in) if not is_last else nn.Identity()]))
        self.final_conv = nn.Sequential(Conv1dBlock(dim, dim, kernel_size=kernel_size), nn.Conv1d(dim, transition_dim, 1))

    # -- embedding MLPs (ma_temporal.py:341-362 / temporal.py:365-384) ----------------------------------
    def embed(self, time, returns, n_agents, use_dropout, force_dropout, dropout_mask=None, env_timestep=None):
        """-> cl [B*A, pad32(embed_dim)] = cat(time_mlp(t), returns_mlp(r) * mask [, env_ts_mlp])."""
        dim = self.time_dim
        bsz = time.shape[0]
        dev = time.device
        if not torch.is_grad_enabled():
            if self.returns_condition and use_dropout and dropout_mask is None and not force_dropout:
                dropout_mask = self.mask_dist.sample(sample_shape=(bsz, n_agents, 1))   # the reference's draw (CPU generator)
            fused = self.embed_fused(time, returns, n_agents, dropout_mask if use_dropout else None, force_dropout)
            if fused is not None:
                return fused
        e = ops.sinusoid(time.to(torch.int64).contiguous(), self.time_mlp[0].freq(dev), n_agents)
        l1, l3 = self.time_mlp[1], self.time_mlp[3]
        e = conv_gemm(mish(conv_gemm(e, [l1.weight], [l1.bias], LINEAR, 1)), [l3.weight], [l3.bias], LINEAR, 1)
        parts = [e[:, :dim]]
        if self.returns_condition:
            assert returns is not None
            r = returns.to(torch.float32).permute(0, 2, 1).reshape(bsz * n_agents, -1)   # "b t a -> b a t", t == 1
            m0, m2, m4 = self.returns_mlp[0], self.returns_mlp[2], self.returns_mlp[4]
            r = conv_gemm(rows_to_cl(r), [m0.weight], [m0.bias], LINEAR, 1)
            r = conv_gemm(mish(r), [m2.weight], [m2.bias], LINEAR, 1)
            r = conv_gemm(mish(r), [m4.weight], [m4.bias], LINEAR, 1)[:, :dim]
            if use_dropout:
                if dropout_mask is None:   # same draw as the reference: CPU generator, shape (B, A, 1)
                    dropout_mask = self.mask_dist.sample(sample_shape=(bsz, n_agents, 1))
                r = dropout_mask.to(device=dev, dtype=r.dtype).reshape(bsz * n_agents, 1) * r
            if force_dropout:
                r = 0 * r
            parts.append(r)
        if self.env_ts_condition:
            assert env_timestep is not None
            ets = env_timestep.to(dtype=torch.int64)[:, self.history_horizon]
            ee = self.env_ts_mlp[0].weight[ets]
            p1, p3 = self.env_ts_mlp[1], self.env_ts_mlp[3]
            ee = conv_gemm(mish(conv_gemm(rows_to_cl(ee), [p1.weight], [p1.bias], LINEAR, 1)), [p3.weight], [p3.bias],
                           LINEAR, 1)[:, :dim]
            parts.append(ee.repeat_interleave(n_agents, dim=0))
        t = torch.cat(parts, dim=-1)
        if pad32(t.shape[1]) != t.shape[1]:
            t = torch.nn.functional.pad(t, (0, pad32(t.shape[1]) - t.shape[1]))
        return t.contiguous()

    def embed_fused(self, time, returns, n_agents, dropout_mask=None, force_dropout=False, act=False, twin=False):
        """embed() as ONE launch (madiff_embed_mlp_fwd) where that applies: bf16 build, on the GPU, autograd off (the
        sampler, the evaluator, validation), no env-timestep branch, dim a multiple of 32 up to 128.  None otherwise.
        act: the result is Mish(embedding) -- what every consumer applies first.  twin: rows [N, 2N) of the result are the
        classifier-free-guidance twins of rows [0, N): same time half, returns half zero (diffusion.py:195-202)."""
        dim = self.time_dim
        if (not EMBED_FUSED or torch.is_grad_enabled() or ops.get_precision() != "bf16" or not time.is_cuda
                or self.env_ts_condition or dim % 32 or dim > 128):
            return None
        bsz = time.shape[0]
        n = bsz * n_agents
        l1, l3 = self.time_mlp[1], self.time_mlp[3]
        W1, W3 = ops._packed([l1.weight], LINEAR, False), ops._packed([l3.weight], LINEAR, False)
        width = dim * (2 if self.returns_condition else 1)
        out = torch.empty(n * (2 if twin else 1), pad32(width), dtype=ops.act_dtype(), device=time.device)
        if out.shape[1] != width:
            out[:, width:].zero_()
        ret = m0 = m2w = m4w = M2 = M4 = None
        if self.returns_condition:
            assert returns is not None
            ret = returns.to(torch.float32).permute(0, 2, 1).reshape(n).contiguous()      # "b t a -> (b a)", t == 1
            m0, m2w, m4w = self.returns_mlp[0], self.returns_mlp[2], self.returns_mlp[4]
            M2, M4 = ops._packed([m2w.weight], LINEAR, False), ops._packed([m4w.weight], LINEAR, False)
        mask = None
        if dropout_mask is not None and self.returns_condition:
            mask = dropout_mask.to(device=time.device, dtype=torch.float32).reshape(n).contiguous()
        ts = time.to(torch.int64).contiguous()
        p = ops._p
        rc = ops.lib.call_rc(
            "madiff_embed_mlp_fwd", ts.data_ptr(), self.time_mlp[0].freq(time.device).data_ptr(), W1.data_ptr(), W1.stride(0),
            l1.bias.data_ptr(), W3.data_ptr(), W3.stride(0), l3.bias.data_ptr(), p(ret),
            None if m0 is None else m0.weight.data_ptr(), None if m0 is None else m0.bias.data_ptr(), p(M2),
            0 if M2 is None else M2.stride(0), None if m2w is None else m2w.bias.data_ptr(), p(M4),
            0 if M4 is None else M4.stride(0), None if m4w is None else m4w.bias.data_ptr(), p(mask), int(bool(force_dropout)),
            out.data_ptr(), out.stride(0), (out.data_ptr() + n * out.stride(0) * out.element_size()) if twin else None,
            bsz, n_agents, dim, int(bool(act)), ops._stream())
        return out if rc == 0 else None

    def rtb_blocks(self):
        """The residual blocks in execution order (their Linear(Mish(t_emb)) run as one GEMM per forward)."""
        blocks: List[ResidualTemporalBlock] = []
        for r1, r2, _ in self.downs:
            blocks += [r1, r2]
        blocks += [self.mid_block1, self.mid_block2]
        for r1, r2, _ in self.ups:
            blocks += [r1, r2]
        return blocks

    def forward_cl(self, xcl, n_seq, t_len, temb):
        """The U-Net alone on n_seq independent sequences (temporal.py:390-406): xcl cl [n_seq*t, pad32(F)], temb cl
        [n_seq, pad32(E)] -> fp32 [n_seq, T, F]."""
        blocks = self.rtb_blocks()
        layers = [b.time_mlp[1] for b in blocks]
        proj_all = conv_gemm(mish(temb), [l.weight for l in layers], [l.bias for l in layers], LINEAR, 1, perm="f32")
        bproj = ops.split_cols(proj_all, tuple(b.out_channels for b in blocks))
        bi = 0
        h, t = xcl, t_len
        skips = []
        for r1, r2, down in self.downs:
            tag = _chain_tag()                  # r1's output feeds r2 only
            h = r1.forward_cl(h, t, bproj[bi], chain_out=tag); bi += 1
            h = r2.forward_cl(h, t, bproj[bi], chain_in=tag); bi += 1
            skips.append((h, t))
            if isinstance(down, Downsample1d):
                h, t = down.forward_cl(h, t)
        tag = _chain_tag()
        h = self.mid_block1.forward_cl(h, t, bproj[bi], chain_out=tag); bi += 1
        h = self.mid_block2.forward_cl(h, t, bproj[bi], chain_in=tag); bi += 1
        for r1, r2, up in self.ups:
            s, ts = skips.pop()
            assert ts == t, "skip/upsample length mismatch (horizon must be divisible by 2**(levels-1))"
            tag = _chain_tag()
            h = r1.forward_cl(h, t, bproj[bi], x2=s, chain_out=tag); bi += 1
            h = r2.forward_cl(h, t, bproj[bi], chain_in=tag); bi += 1
            if isinstance(up, Upsample1d):
                h, t = up.forward_cl(h, t)
        h = self.final_conv[0].forward_cl(h, t)
        fc = self.final_conv[1]
        out = conv_gemm(h, [fc.weight], [fc.bias], CONV1, t, perm=(t, 1))
        return out.reshape(n_seq, t, self.transition_dim)

    def forward(self, x, time, returns=None, env_timestep=None, attention_masks=None, use_dropout=True,
                force_dropout=False, dropout_mask=None):
        """temporal.py:348-406.  x : [ batch x horizon x transition ];  returns : [ batch x 1 ].  dropout_mask (optional,
        [batch, 1]): the explicit Bernoulli draw of the use_dropout branch (otherwise drawn like the reference)."""
        bsz, t_len = x.shape[0], x.shape[1]
        if self.returns_condition and use_dropout and dropout_mask is None:
            dropout_mask = self.mask_dist.sample(sample_shape=(bsz, 1))
        temb = self.embed(time, None if returns is None else returns.reshape(bsz, 1, 1), 1, use_dropout, force_dropout,
                          None if dropout_mask is None else dropout_mask.reshape(bsz, 1, 1), env_timestep)
        xcl = PackTraj.apply(x.to(torch.float32).unsqueeze(2))
        return self.forward_cl(xcl, bsz, t_len, temb)


# ------------------------------------------------------------------------------------------------
# ma_temporal.py
# ------------------------------------------------------------------------------------------------
class SharedConvAttentionDeconv(nn.Module):
    """ma_temporal.py:229-418: parameter-shared temporal U-Net + cross-agent attention at the bottleneck and
    on every skip connection.  forward(x[B,T,A,F], time[B], returns[B,1,A]) -> eps [B,T,A,F] (fp32)."""

    agent_share_parameters = True

    def __init__(self, horizon: int, transition_dim: int, dim: int = 128, history_horizon: int = 0,
                 dim_mults: Tuple[int] = (1, 2, 4, 8), nhead: int = 4, n_agents: int = 2,
                 returns_condition: bool = False, env_ts_condition: bool = False, condition_dropout: float = 0.1,
                 kernel_size: int = 5, residual_attn: bool = True, use_layer_norm: bool = False,
                 max_path_length: int = 100, use_temporal_attention: bool = True):
        super().__init__()
        if use_layer_norm:
            raise NotImplementedError("use_layer_norm is not selected by any exp_spec (SURVEY.md section 8f, row F4)")
        if not use_temporal_attention:
            # the reference builds TemporalSelfAttention layers regardless (ma_temporal.py:277-294) and then calls them
            # without the time embedding (:382,:399): its forward raises TypeError.  Nothing to be compatible with.
            raise NotImplementedError("use_temporal_attention=False cannot run in the reference either "
                                      "(ma_temporal.py:382 calls TemporalSelfAttention.forward without its time argument)")
        self.n_agents = n_agents
        self.history_horizon = history_horizon
        self.use_temporal_attention = use_temporal_attention
        self.returns_condition = returns_condition
        self.env_ts_condition = env_ts_condition
        self.use_layer_norm = use_layer_norm
        self.transition_dim = transition_dim
        dims = [transition_dim, *map(lambda m: dim * m, dim_mults)]
        in_out = list(zip(dims[:-1], dims[1:]))
        self.net = TemporalUnet(horizon=horizon, history_horizon=history_horizon, transition_dim=transition_dim,
                                dim=dim, dim_mults=dim_mults, returns_condition=returns_condition,
                                env_ts_condition=env_ts_condition, condition_dropout=condition_dropout,
                                max_path_length=max_path_length, kernel_size=kernel_size)
        attn = [TemporalSelfAttention(in_out[-1][1], in_out[-1][1] // 16, in_out[-1][1] // 4, residual=residual_attn,
                                      embed_dim=self.net.embed_dim, nheads=nhead)]
        for d in reversed(in_out):
            attn.append(TemporalSelfAttention(d[1], d[1] // 16, d[1] // 4, residual=residual_attn,
                                              embed_dim=self.net.embed_dim, nheads=nhead))
        self.self_attn = nn.ModuleList(attn)

    # every Linear(Mish(t_emb)) of the live graph, in execution order, evaluated by ONE GEMM per forward
    def _time_layers(self):
        net = self.net
        blocks = net.rtb_blocks()
        layers = [b.time_mlp[1] for b in blocks]
        n_live_attn = 1 + len(net.ups)          # self_attn[len(ups)+1:] are constructed but never called
        for i in range(n_live_attn):
            layers += self.self_attn[i].time_layers()
        return blocks, layers, n_live_attn

    # ---- sampling loop only (diffusion.py:214-275): every trajectory of the batch is at the SAME diffusion step, so with
    # e = [time_mlp(t) | returns_mlp(r_n)] the projection  W Mish(e) + b  of all 21 time layers splits into
    #   v_t = W [Mish(time_mlp(t)) | 0] + b     one vector per denoising step                      (proj_step)
    #   R_n = W [0 | Mish(returns_mlp(r_n))]    one row per sequence, constant over the K steps    (proj_static)
    # and the consumers add  R[n, c] + v_t[c]  (ops.rowvec_of).  The [2 B A, 10496] fp32 projection GEMM of every step
    # (108 us and 258 MB written per step at 1024 episodes) becomes a one-row GEMM.
    def proj_static(self, returns, bsz):
        """R for the classifier-free-guidance pair batch (rows [0, N) conditional, [N, 2N) unconditional = zeros), with the
        q/k/v convolution biases of the attention layers folded in (the per-step vector takes their bias operand).  None
        when the split does not apply (then forward_guided_pair runs the one-GEMM form)."""
        net, A = self.net, self.n_agents
        if (not PROJ_SPLIT or torch.is_grad_enabled() or not net.returns_condition or returns is None
                or 2 * bsz * A < PROJ_SPLIT_MIN_SEQ):
            return None
        both = net.embed_fused(torch.zeros(bsz, dtype=torch.int64, device=returns.device), returns, A, None, False, act=True,
                               twin=True)
        if both is None:
            return None
        both[:, : net.time_dim].zero_()                      # the returns half only
        blocks, layers, n_attn = self._time_layers()
        R = conv_gemm(both, [l.weight for l in layers], [], LINEAR, 1, perm="f32")
        off = sum(b.out_channels for b in blocks)
        for i in range(n_attn):
            a = self.self_attn[i]
            w = 2 * a.cq + a.cv
            R[:, off:off + w] += torch.cat((a.query_layer.bias, a.key_layer.bias, a.value_layer.bias)).detach()
            off += w
        return R

    def proj_step(self, time):
        """v_t fp32 [1, pad32(total)] for the step all of `time` holds (graph-capturable: reads time[0] on the device)."""
        net = self.net
        one = net.embed_fused(time[:1], torch.zeros(1, 1, 1, device=time.device), 1, None, True, act=True, twin=False)
        if one is None:
            return None
        _, layers, _ = self._time_layers()
        return conv_gemm(one, [l.weight for l in layers], [l.bias for l in layers], LINEAR, 1, perm="f32")

    def forward_cl(self, xcl, bsz, t_len, temb, temb_act=False, proj=None):
        """xcl: cl [(b a) t, pad32(F)]; temb: cl [B*A, pad32(E)] (temb_act: already Mish(embedding), as embed_fused can
        produce it), or proj = (R, v_t) from proj_static / proj_step instead of temb.  Returns eps fp32 [B, T, A, F]."""
        net, A = self.net, self.n_agents
        blocks, layers, n_attn = self._time_layers()
        sizes = [b.out_channels for b in blocks] + [2 * self.self_attn[i].cq + self.self_attn[i].cv for i in range(n_attn)]
        if proj is not None:
            proj_all, vec = proj
            assert not torch.is_grad_enabled() and proj_all.shape[0] == bsz * A
        else:
            proj_all = conv_gemm(temb if temb_act else mish(temb), [l.weight for l in layers], [l.bias for l in layers],
                                 LINEAR, 1, perm="f32")
        views = ops.split_cols(proj_all, tuple(sizes))
        if proj is not None:
            off = 0
            for vw, sz in zip(views, sizes):
                vw._madiff_rowvec = vec[0, off:off + sz]
                off += sz
        bproj, aproj = views[: len(blocks)], views[len(blocks):]
        bi = 0
        h, t = xcl, t_len
        skips = []
        for r1, r2, down in net.downs:
            tag = _chain_tag()                  # r1's output feeds r2 only
            h = r1.forward_cl(h, t, bproj[bi], chain_out=tag); bi += 1
            h = r2.forward_cl(h, t, bproj[bi], chain_in=tag); bi += 1
            skips.append((h, t))
            if isinstance(down, Downsample1d):
                h, t = down.forward_cl(h, t)
        tag = _chain_tag()
        h = net.mid_block1.forward_cl(h, t, bproj[bi], chain_out=tag); bi += 1
        h = net.mid_block2.forward_cl(h, t, bproj[bi], chain_in=tag); bi += 1
        h = self.self_attn[0].forward_cl(h, bsz, A, t, aproj[0])
        for j, (r1, r2, up) in enumerate(net.ups):
            s, ts = skips.pop()
            assert ts == t, "skip/upsample length mismatch (horizon must be divisible by 2**(levels-1))"
            s = self.self_attn[j + 1].forward_cl(s, bsz, A, t, aproj[j + 1])
            tag = _chain_tag()
            h = r1.forward_cl(h, t, bproj[bi], x2=s, chain_out=tag); bi += 1
            h = r2.forward_cl(h, t, bproj[bi], chain_in=tag); bi += 1
            if isinstance(up, Upsample1d):
                h, t = up.forward_cl(h, t)
        h = net.final_conv[0].forward_cl(h, t)
        fc = net.final_conv[1]
        out = conv_gemm(h, [fc.weight], [fc.bias], CONV1, t, perm=(t, A))
        return out.reshape(bsz, t, A, self.transition_dim)

    def forward(self, x, time, returns=None, states=None, env_timestep=None, attention_masks=None,
                use_dropout: bool = True, force_dropout: bool = False, dropout_mask=None, **kwargs):
        """x : [ batch x horizon x agent x transition ];  returns : [ batch x 1 x agent ]."""
        assert x.shape[2] == self.n_agents, f"Expected {self.n_agents} agents, but got samples with shape {x.shape}"
        bsz, t_len = x.shape[0], x.shape[1]
        temb = self.net.embed(time, returns if self.returns_condition else None, self.n_agents, use_dropout,
                              force_dropout, dropout_mask, env_timestep)
        return self.forward_cl(PackTraj.apply(x.to(torch.float32)), bsz, t_len, temb)

    def forward_guided_pair(self, xcl, bsz, t_len, time, returns, env_timestep=None, proj_R=None):
        """Classifier-free-guidance pair as ONE 2B batch (weights are read once): rows [0, B) are the
        conditional pass (use_dropout=False), rows [B, 2B) the unconditional one (force_dropout=True)
        (diffusion.py:187-202).  xcl holds the B conditioned inputs; returns (eps_cond, eps_uncond)."""
        A = self.n_agents
        if proj_R is not None and env_timestep is None:          # sampler: returns half precomputed, time half one vector
            vec = self.proj_step(time)
            self.last_proj_split = vec is not None             # (tests)
            if vec is not None:
                out = self.forward_cl(torch.cat((xcl, xcl), dim=0), 2 * bsz, t_len, None, proj=(proj_R, vec))
                return out[:bsz], out[bsz:]
        both = self.net.embed_fused(time, returns, A, None, False, act=True, twin=True) if env_timestep is None else None
        if both is not None:          # one launch: both embeddings of the pair, Mish applied
            out = self.forward_cl(torch.cat((xcl, xcl), dim=0), 2 * bsz, t_len, both, temb_act=True)
            return out[:bsz], out[bsz:]
        tc = self.net.embed(time, returns, A, False, False, None, env_timestep)
        tu = self.net.embed(time, returns, A, False, True, None, env_timestep)
        out = self.forward_cl(torch.cat((xcl, xcl), dim=0), 2 * bsz, t_len, torch.cat((tc, tu), dim=0))
        return out[:bsz], out[bsz:]
