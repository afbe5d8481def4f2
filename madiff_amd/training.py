"""Drop-in for the reference's training loop object (diffuser/utils/training.py: ``cycle``, ``EMA``, ``Trainer``).

Same constructor arguments, attributes (``model``, ``ema_model``, ``step``, ``optimizer``, ``dataloader`` ...), hooks
(``set_evaluator``, ``finish_training``, ``save``/``load`` with the ``{"step", "model", "ema"}`` checkpoint layout) and
schedule as the reference:

    for step:  gradient_accumulate_every x [ batch -> loss(**batch) / accumulate -> backward ]   (training.py:147-152)
               Adam(lr) step, zero_grad                                                        (:109, :154-155)
               if step % update_ema_every == 0:  EMA := copy of the model while step < step_start_ema,
                                                 else ema*beta + (1-beta)*model                (:134-138, :157-158)
               save / evaluate / log / render hooks                                            (:160-187)

What is different underneath: the optimizer is ``FlatAdamEMA`` (flat fp32 buffers, fused Adam+EMA slices issued while
backward is still running, bucketed RCCL all-reduce when ``torch.distributed`` is initialised with more than one rank),
the whole optimizer step is replayed from one hipGraph once shapes are known (``GraphedTrainStep``), and batches reach
the GPU through a pinned-memory prefetcher on a copy stream instead of a blocking ``batch_to_device`` per micro-batch.
The reference is a single process; under data parallelism every rank draws its own shard of each epoch
(``DistributedSampler``), rank 0 alone logs and saves.
"""
from __future__ import annotations

import copy
import os
import time
from typing import Optional

import torch
import torch.distributed as dist

from . import ops
from .host import batch_to_device
from .trainer import FlatAdamEMA, GraphedTrainStep, train_step

try:                                            # the reference logs through ml_logger (absent from this image)
    from ml_logger import logger as _ml_logger  # type: ignore
except Exception:                               # noqa: BLE001
    _ml_logger = None


class _PrintLogger:
    """The four ml_logger calls the reference's Trainer makes (print / log / prefix / save via torch.save)."""
    prefix = ""

    def print(self, *args, **kwargs):
        kwargs.pop("color", None)
        print(*args, **{k: v for k, v in kwargs.items() if k in ("sep", "end", "file", "flush")})

    def log(self, *args, flush=False, **metrics):
        pass


logger = _ml_logger if _ml_logger is not None else _PrintLogger()


def cycle(dl):
    """training.py:14-17; a sampler with ``set_epoch`` (DistributedSampler) is advanced at every pass."""
    epoch = 0
    while True:
        sampler = getattr(dl, "sampler", None)
        if hasattr(sampler, "set_epoch"):
            sampler.set_epoch(epoch)
        for data in dl:
            yield data
        epoch += 1


class EMA:
    """training.py:20-39 (kept for API parity; the Trainer's own EMA runs inside the fused optimizer kernel)."""

    def __init__(self, beta):
        self.beta = beta

    def update_model_average(self, ma_model, current_model):
        with torch.no_grad():
            for cur, ma in zip(current_model.parameters(), ma_model.parameters()):
                ma.data.copy_(self.update_average(ma.data, cur.data))

    def update_average(self, old, new):
        if old is None:
            return new
        return old * self.beta + (1 - self.beta) * new


class DevicePrefetcher:
    """Iterator over device-resident batches: the next ``depth`` host batches (pinned by the DataLoader) are copied on a
    dedicated stream while the current step runs; ``next()`` makes the compute stream wait for the copy's event only."""

    def __init__(self, it, device, depth: int = 2):
        self.it, self.device, self.depth = it, torch.device(device), depth
        self.cuda = self.device.type == "cuda"
        self.stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.queue = []

    def _fetch(self):
        batch = next(self.it)
        if not self.cuda:
            self.queue.append((batch_to_device(batch, self.device), None))
            return
        with torch.cuda.stream(self.stream):
            mv = lambda v: ({k: mv(u) for k, u in v.items()} if isinstance(v, dict)            # noqa: E731
                            else v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v)
            dev_batch = mv(batch)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.queue.append((dev_batch, ev))

    def __iter__(self):
        return self

    def __next__(self):
        while len(self.queue) < self.depth:
            self._fetch()
        batch, ev = self.queue.pop(0)
        if ev is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            rs = lambda v: ([rs(u) for u in v.values()] if isinstance(v, dict)                # noqa: E731
                            else v.record_stream(cur) if torch.is_tensor(v) else None)
            rs(batch)
        return batch


class Trainer(object):
    def __init__(self, diffusion_model, dataset, renderer, ema_decay=0.995, train_batch_size=32, train_lr=2e-5,
                 gradient_accumulate_every=2, step_start_ema=2000, update_ema_every=10, log_freq=100, sample_freq=1000,
                 save_freq=1000, label_freq=100000, eval_freq=100000, save_parallel=False, n_reference=8, bucket=None,
                 train_device="cuda", save_checkpoints=False, use_graph: Optional[bool] = None, bucket_mb: float = 64.0,
                 shuffle: bool = True, seed: int = 0, allow_eager_fallback: Optional[bool] = None):
        """Reference arguments (training.py:43-64) plus: use_graph (None = replay the step from a hipGraph on a GPU),
        bucket_mb (gradient bucket size of the data-parallel all-reduce / optimizer slices), shuffle, seed.
        The reference's DataLoader keeps the smaller last batch of an epoch; a replayed graph (and equal work per
        rank) needs fixed shapes, so that batch is dropped when use_graph is on or under data parallelism."""
        super().__init__()
        self.model = diffusion_model
        self.ema = EMA(ema_decay)
        self.ema_model = copy.deepcopy(self.model)
        self.update_ema_every = update_ema_every
        self.save_checkpoints = save_checkpoints
        self.step_start_ema = step_start_ema
        assert eval_freq % save_freq == 0, \
            f"eval_freq must be a multiple of save_freq, but got {eval_freq} and {save_freq} respectively"
        self.log_freq, self.sample_freq, self.save_freq = log_freq, sample_freq, save_freq
        self.label_freq, self.eval_freq, self.save_parallel = label_freq, eval_freq, save_parallel
        self.batch_size = train_batch_size
        self.gradient_accumulate_every = gradient_accumulate_every
        self.device = train_device
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        self.dataset = dataset
        on_gpu = torch.device(train_device).type == "cuda"
        self.use_graph = on_gpu if use_graph is None else use_graph
        # a failed capture of the train step raises unless this is set (then all ranks drop to the eager step together)
        self.allow_eager_fallback = (os.environ.get("MADIFF_ALLOW_EAGER_FALLBACK") == "1"
                                     if allow_eager_fallback is None else bool(allow_eager_fallback))
        if dataset is not None:
            sampler = None
            if self.world > 1:
                sampler = torch.utils.data.distributed.DistributedSampler(dataset, num_replicas=self.world, rank=self.rank,
                                                                          shuffle=shuffle, seed=seed)
            self.dataloader = DevicePrefetcher(cycle(torch.utils.data.DataLoader(
                self.dataset, batch_size=train_batch_size, num_workers=0, shuffle=shuffle and sampler is None,
                sampler=sampler, pin_memory=on_gpu, drop_last=self.use_graph or self.world > 1)), train_device)
            self.dataloader_vis = cycle(torch.utils.data.DataLoader(
                self.dataset, batch_size=1, num_workers=0, shuffle=True, pin_memory=on_gpu))
        self.renderer = renderer
        # Adam(lr, torch defaults) + EMA over the same parameters as the reference's torch.optim.Adam (training.py:109)
        self.optimizer = FlatAdamEMA(diffusion_model, self.ema_model, lr=train_lr, ema_decay=ema_decay, bucket_mb=bucket_mb)
        self.optimizer.broadcast_parameters(0)
        self.bucket = bucket
        self.n_reference = n_reference
        self.reset_parameters()
        self.step = 0
        self.evaluator = None
        self._graphed: Optional[GraphedTrainStep] = None
        self._eager_steps = 0

    # ------------------------------------------------------------------ reference hooks
    def set_evaluator(self, evaluator):
        self.evaluator = evaluator

    def finish_training(self):
        if self.step % self.save_freq == 0:
            self.save()
        if self.eval_freq > 0 and self.step % self.eval_freq == 0:
            self.evaluate()
        if self.evaluator is not None:
            del self.evaluator

    def reset_parameters(self):
        self.ema_model.load_state_dict(self.model.state_dict())
        ops.params_changed()

    def ema_mode(self, step: Optional[int] = None) -> int:
        """0 = no EMA update this step, 2 = copy (step < step_start_ema), 1 = lerp (training.py:134-138,157-158)."""
        step = self.step if step is None else step
        if step % self.update_ema_every != 0:
            return 0
        return 2 if step < self.step_start_ema else 1

    def step_ema(self):
        """Reference API (training.py:134-138); train() folds the same update into the optimizer kernel instead."""
        if self.step < self.step_start_ema:
            self.reset_parameters()
            return
        self.ema.update_model_average(self.ema_model, self.model)
        ops.params_changed()

    # ------------------------------------------------------------------ api
    def _optimizer_step(self, batches, mode):
        opt = self.optimizer
        if self._graphed is not None:
            return self._graphed.step(batches, do_ema=mode)
        # two eager steps first (flat buffers materialised, GEMM variants tuned); the third is performed by the
        # constructor of the graphed step (eagerly, on its capture stream), which then captures the step once
        if self.use_graph and opt.use_hip and self._eager_steps >= 2 and batches[0]["x"].is_cuda:
            done_before = opt.step_count
            err, graphed = None, None
            try:
                graphed = GraphedTrainStep(self.model, opt, batches, warm_do_ema=mode)
            except Exception as e:                      # noqa: BLE001
                err = e
                opt._announced = None                   # nothing of an interrupted capture may leak into the next step
                opt._graph_order = []
            ok = err is None
            if self.world > 1:
                # every rank must take the same path: a rank that replays while another issues the eager step would
                # pair its per-bucket all-reduces (capture order) with the other's (hook order)
                import torch.distributed as dist
                flag = torch.tensor([1 if ok else 0], device=batches[0]["x"].device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = bool(int(flag.item()))
            if ok:
                self._graphed = graphed
                return graphed.first
            if not self.allow_eager_fallback:
                # a capture regression must not hide behind a slower run
                raise RuntimeError("hipGraph capture of the train step failed"
                                   + ("" if err is not None else " on another rank")
                                   + "; pass allow_eager_fallback=True (or MADIFF_ALLOW_EAGER_FALLBACK=1) to train with "
                                     "eagerly issued steps instead") from err
            logger.print(f"[ utils/training ] hipGraph capture of the train step failed ({err!r}); staying eager")
            self.use_graph, self._graphed = False, None
            del graphed
            if opt.step_count > done_before:            # the constructor's eager step did happen before the failure
                torch.cuda.synchronize()
                with torch.no_grad():
                    loss, infos = self.model.loss(**batches[-1])
                return loss.detach(), infos
        out = train_step(self.model, opt, batches, do_ema=mode)
        self._eager_steps += 1
        return out

    def train(self, n_train_steps):
        t0 = time.time()
        for _ in range(n_train_steps):
            batches = [next(self.dataloader) for _ in range(self.gradient_accumulate_every)]
            loss, infos = self._optimizer_step(batches, self.ema_mode())
            loss = loss / self.gradient_accumulate_every

            if self.step % self.save_freq == 0:
                self.save()
            if self.eval_freq > 0 and self.step % self.eval_freq == 0:
                self.evaluate()
            if self.step % self.log_freq == 0 and self.rank == 0:
                infos = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in infos.items()}
                infos_str = " | ".join([f"{key}: {float(val):8.4f}" for key, val in infos.items()])
                logger.print(f"{self.step}: {float(loss):8.4f} | {infos_str} | t: {time.time() - t0:8.4f}")
                metrics = {k: float(v.detach().item()) if torch.is_tensor(v) else float(v) for k, v in infos.items()}
                logger.log(step=self.step, loss=float(loss), **metrics, flush=True)
                t0 = time.time()
            if self.renderer is not None and self.sample_freq and self.step == 0 and hasattr(self, "render_reference"):
                self.render_reference(self.n_reference)
            if self.renderer is not None and self.sample_freq and self.step % self.sample_freq == 0 \
                    and hasattr(self, "inv_render_samples"):
                self.inv_render_samples()
            self.step += 1

    def evaluate(self):
        assert self.evaluator is not None, \
            "Method `evaluate` can not be called when `self.evaluator` is None. Set evaluator with `self.set_evaluator` first."
        self.evaluator.evaluate(load_step=self.step)

    def _checkpoint_dir(self):
        return os.path.join(self.bucket or ".", getattr(logger, "prefix", "") or "", "checkpoint")

    def save(self):
        """training.py:197-215: {"step", "model", "ema"} -> <bucket>/<prefix>/checkpoint/state[_<step>].pt (rank 0)."""
        if self.rank != 0:
            return
        data = {"step": self.step, "model": self.model.state_dict(), "ema": self.ema_model.state_dict()}
        savepath = self._checkpoint_dir()
        os.makedirs(savepath, exist_ok=True)
        savepath = os.path.join(savepath, f"state_{self.step}.pt" if self.save_checkpoints else "state.pt")
        torch.save(data, savepath)
        logger.print(f"[ utils/training ] Saved model to {savepath}")

    def load(self):
        """training.py:217-227."""
        loadpath = os.path.join(self._checkpoint_dir(), "state.pt")
        data = torch.load(loadpath, map_location="cpu", weights_only=True)
        self.step = data["step"]
        self.model.load_state_dict(data["model"])
        self.ema_model.load_state_dict(data["ema"])
        ops.params_changed()
