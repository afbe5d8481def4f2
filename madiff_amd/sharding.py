"""Episode sharding for evaluation-time sampling.

Reference call site: ``MADEvaluatorWorker._generate_samples`` (diffuser/utils/evaluator.py:101-106,127-134) samples all
``num_envs`` episodes in one ``conditional_sample`` call on one device.  Episodes are independent, so here every rank
takes a contiguous slice of the episode axis, runs its own K-step loop (its own captured hipGraph) with NO collective
inside the loop, and the plans are brought together by ONE final gather.  The reference has no counterpart (it is
single-process); a single process (no process group) degenerates to the plain call.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_episodes: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of the episodes rank `rank` samples: contiguous, sizes differ by at most one, empty when
    n_episodes < world for the last ranks."""
    assert 0 <= rank < world and n_episodes >= 0
    q, r = divmod(n_episodes, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def _cut(v, lo, hi):
    if v is None:
        return None
    if isinstance(v, dict):
        return {k: _cut(u, lo, hi) for k, u in v.items()}
    return v[lo:hi].contiguous()


def sample_sharded(diffusion, cond: Dict[str, torch.Tensor], returns: Optional[torch.Tensor] = None,
                   env_ts: Optional[torch.Tensor] = None, group=None, dst: Optional[int] = 0, **sample_kwargs):
    """Every rank passes the FULL episode batch (cond["x"], cond["masks"] [B,T,A,F], returns [B,1,A], env_ts [B,T]) —
    as the evaluator holds it — or at least its own slice's rows at the right indices; each rank samples rows
    shard_range(B, rank, world) with `diffusion.conditional_sample`, then one gather collects [B,T,A,F].

    dst = r: gather to rank r (others return None) — evaluator.py's single consumer; dst = None: all-gather.
    Per-episode keyword tensors (x_init [B,...]; step_noise [n_steps,B,...]) are cut to the shard as well."""
    B = cond["x"].shape[0]
    if not (dist.is_available() and dist.is_initialized()):
        return diffusion.conditional_sample(cond, returns=returns, env_ts=env_ts, verbose=False, **sample_kwargs)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_range(B, rank, world)
    kw = dict(sample_kwargs)
    if kw.get("x_init") is not None:
        kw["x_init"] = kw["x_init"][lo:hi].contiguous()
    if kw.get("step_noise") is not None:
        kw["step_noise"] = kw["step_noise"][:, lo:hi].contiguous()
    if hi > lo:
        local = diffusion.conditional_sample(_cut(cond, lo, hi), returns=_cut(returns, lo, hi), env_ts=_cut(env_ts, lo, hi),
                                             verbose=False, **kw)
    else:
        local = cond["x"].new_empty((0,) + tuple(cond["x"].shape[1:]), dtype=torch.float32)
    local = local.contiguous()
    # the one collective of the sampling path.  Shards may be ragged (B % world != 0): pad to the largest shard.
    biggest = shard_range(B, 0, world)
    n_max = biggest[1] - biggest[0]
    if local.shape[0] < n_max:
        pad = local.new_zeros((n_max - local.shape[0],) + tuple(local.shape[1:]))
        local = torch.cat([local, pad], dim=0)
    if dst is None:
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(parts, local, group=group)
    else:
        parts = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
        dist.gather(local, parts, dst=dst, group=group)
        if rank != dst:
            return None
    out = []
    for r, p in enumerate(parts):
        l, h = shard_range(B, r, world)
        out.append(p[: h - l])
    return torch.cat(out, dim=0)
