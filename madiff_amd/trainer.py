"""Data-parallel optimizer step for the denoiser: flat parameter/gradient buffers, bucketed gradient all-reduce
(RCCL over xGMI through torch.distributed) overlapped with backward, and one fused Adam(+EMA) launch.

Reference behaviour reproduced (diffuser/utils/training.py): Adam(lr, torch defaults) over
``diffusion.parameters()`` (:109), ``gradient_accumulate_every`` micro-batches each scaled by 1/accumulate
(:147-152), EMA every ``update_ema_every`` steps -- plain copy before ``step_start_ema`` then a lerp with
``ema_decay`` (:20-39,134-138).  The reference is single-process; the data-parallel part is new and changes no
single-process result: world_size == 1 performs no collective.

Layout: every parameter that can receive a gradient is re-pointed into ONE flat fp32 buffer (forward/registration
order) and its ``.grad`` into a second flat buffer, so the optimizer is a single kernel over 64 M elements and a
gradient bucket is a contiguous slice.  Parameters that never receive a gradient (``self_attn.3.*`` of the
reference architecture) are detected on the first backward and left out, exactly as torch.optim.Adam
skips ``grad is None``.  Buckets are cut in REVERSE registration order (the order backward produces them); each is
all-reduced on a side stream as soon as all of its gradients of the last micro-batch have been accumulated.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from . import lib, ops


class FlatAdamEMA:
    def __init__(self, module: torch.nn.Module, ema_module: Optional[torch.nn.Module] = None, lr: float = 2e-4,
                 betas=(0.9, 0.999), eps: float = 1e-8, ema_decay: float = 0.995, bucket_mb: float = 64.0,
                 process_group=None, use_hip_kernel: Optional[bool] = None, bucket_dtype: Optional[str] = None):
        self.module, self.ema_module = module, ema_module
        self.lr, self.betas, self.eps, self.ema_decay = lr, betas, eps, ema_decay
        self.step_count = 0
        self.bucket_bytes = int(bucket_mb * 2 ** 20)
        # gradient transport: "fp32" (default: the all-reduce sums the flat fp32 gradient in place) or "bf16" (each bucket is
        # rounded to bf16 into a staging buffer, all-reduced there -- half the bytes on the wire -- and written back into the
        # fp32 master gradient; Adam, its moments and the weights stay fp32).  Lossy (8 mantissa bits per addend), off by default.
        self.bucket_dtype = bucket_dtype or os.environ.get("MADIFF_BUCKET_DTYPE", "fp32")
        if self.bucket_dtype not in ("fp32", "bf16"):
            raise ValueError(f"bucket_dtype must be 'fp32' or 'bf16', not {self.bucket_dtype!r}")
        self._stage16 = None
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        # dp: the bucketed all-reduce machinery is active (MADIFF_FORCE_DP=1 keeps it on in a 1-rank group: tests)
        self.dp = self.world > 1 or (os.environ.get("MADIFF_FORCE_DP") == "1" and dist.is_available() and dist.is_initialized())
        self._graph_events = None        # bucket index -> graph-external event (while / after capturing a DP step)
        self._graph_order: List[int] = []
        self.params: List[torch.nn.Parameter] = []
        self.names: List[str] = []
        self.flat_p = self.flat_g = self.flat_m = self.flat_v = self.flat_ema = None
        self.buckets: List[dict] = []
        self._hooks = []
        self._sync_this_backward = True
        self._notified = set()
        self._comm_stream = None
        self.use_hip = use_hip_kernel
        self.grad_scale = 1.0
        # overlap: the optimizer step is issued per gradient bucket, as soon as the bucket is complete (single process:
        # on the "wgrad" auxiliary stream; data parallel: on the communication stream right behind the bucket's
        # all-reduce), so the HBM-bound Adam/EMA pass runs under the MFMA-bound rest of backward instead of after it
        self.overlap = os.environ.get("MADIFF_OVERLAP_ADAM", "1") != "0"
        self._announced = None           # (do_ema, device_state) of the step whose backward is running (begin_step)
        self._main_stream = None
        # data-parallel capture: put the bucket event nodes on the MAIN branch, i.e. rejoin the parallel branches into it
        # at every bucket (a serialisation point per bucket; default off.  Kept as a tested alternative topology, see
        # DESIGN.md section 5 "capture failures")
        self.bucket_events_on_main = os.environ.get("MADIFF_BUCKET_EVENTS_ON_MAIN") == "1"

    # ------------------------------------------------------------------ setup after the first backward
    def _materialise(self):
        """Called once, after a backward pass has shown which parameters are live."""
        named = [(n, p) for n, p in self.module.named_parameters() if p.requires_grad and p.grad is not None]
        assert named, "run one backward pass before the first optimizer step"
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]
        dev = self.params[0].device
        if self.use_hip is None:
            if dev.type != "cuda":
                # the product optimizer is the HIP kernel; the host-arithmetic branch below exists for the gloo tests of
                # the data-parallel machinery and has to be asked for by name
                raise RuntimeError("FlatAdamEMA: the module is not on a GPU.  The fused Adam/EMA step is a HIP kernel; pass "
                                   "use_hip_kernel=False explicitly to run the host-arithmetic test path")
            self.use_hip = True
        elif self.use_hip and dev.type != "cuda":
            raise RuntimeError("FlatAdamEMA(use_hip_kernel=True) needs the module on a GPU")
        # ---- flat layout.  Weight matrices of GEMM layers are stored in the GEMM's own layout [O, taps, pad32(I)]
        # (pad columns stay 0: zero gradient -> zero Adam update), members of a fused group (q/k/v, the time-MLP
        # bundle) back to back; the nn.Parameter becomes a strided view with the reference's shape.  Adam/EMA are
        # elementwise, so the layout is free for them, and the GEMMs then need no pack/unpack pass at all.
        known = ops.known_gemm_weights() if self.use_hip else {}
        by_id = {id(p): (n, p) for n, p in named}
        order, seen = [], set()
        for n, p in named:
            if id(p) in seen:
                continue
            info = known.get(id(p))
            members = [p]
            if info is not None and len(info[2]) > 1 and all(i in by_id for i in info[2]):
                members = [by_id[i][1] for i in info[2]]
            for q in members:
                if id(q) not in seen:
                    seen.add(id(q))
                    order.append(q)
        name_of = {id(p): n for n, p in named}
        self.params = order
        self.names = [name_of[id(p)] for p in order]
        offs, total, geoms = [], 0, []
        for p in self.params:
            offs.append(total)
            info = known.get(id(p))
            if info is not None:
                kind, taps, _ = info
                O, I = (p.shape[0], p.shape[1]) if kind != "convT" else (p.shape[1], p.shape[0])
                ipad = ops.pad32(I)
                geoms.append((kind, taps, O, I, ipad))
                total += O * taps * ipad
            else:
                geoms.append(None)
                total += (p.numel() + 3) // 4 * 4      # keep every slice 16-byte aligned
        self.offsets, self.total, self.geoms = offs, total, geoms

        def view_of(flat, o, p, geom):
            if geom is None:
                return flat[o:o + p.numel()].view_as(p)
            kind, taps, O, I, ipad = geom
            v = flat[o:o + O * taps * ipad].view(O, taps, ipad)[:, :, :I]
            if kind == "conv":
                return v.permute(0, 2, 1)
            if kind == "convT":
                return v.permute(2, 0, 1)
            return v[:, 0, :]

        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        ema_named: Dict[str, torch.nn.Parameter] = dict(self.ema_module.named_parameters()) if self.ema_module is not None else {}
        if self.ema_module is not None:
            self.flat_ema = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad_views = []
        with torch.no_grad():
            for n, p, o, geom in zip(self.names, self.params, offs, geoms):
                pv, gv = view_of(self.flat_p, o, p, geom), view_of(self.flat_g, o, p, geom)
                pv.copy_(p.detach())
                gv.copy_(p.grad)
                p.data = pv
                p.grad = gv
                self.grad_views.append(gv)
                if self.ema_module is not None:
                    e = ema_named[n]
                    ev = view_of(self.flat_ema, o, e, geom)
                    ev.copy_(e.detach())
                    e.data = ev
        ops.clear_pack_cache()
        ops.params_changed()
        # buckets: contiguous slices, cut from the END of the flat buffer (first gradients backward produces)
        self.buckets = []
        hi = total
        idx = len(self.params) - 1
        while idx >= 0:
            members = []
            lo = hi
            while idx >= 0 and (not members or (hi - offs[idx]) * 4 <= self.bucket_bytes):
                members.append(idx)
                lo = offs[idx]
                idx -= 1
            self.buckets.append({"lo": lo, "hi": hi, "members": members, "pending": len(members), "work": None})
            hi = lo
        self._bucket_of = {}
        for b in self.buckets:
            for i in b["members"]:
                self._bucket_of[i] = b
        self._index = {id(p): i for i, p in enumerate(self.params)}
        hooked = self.dp or (self.overlap and self.use_hip)
        if hooked:
            if self.dp and dev.type == "cuda":
                self._comm_stream = torch.cuda.Stream(device=dev)
            for i, p in enumerate(self.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
        if self.use_hip:
            # backward kernels accumulate parameter gradients straight into the flat buffer (ops.GradSink); the GEMMs
            # read / write the GEMM-layout slots of the flat buffers directly
            sink = ops.GradSink()
            sink.flat_p, sink.flat_g = self.flat_p, self.flat_g
            self.flat_p16 = self.flat_p.to(torch.bfloat16)      # bf16 shadow, kept in sync by madiff_adam_ema
            sink.flat_p16 = self.flat_p16
            for p, o, geom, gv in zip(self.params, offs, geoms, self.grad_views):
                sink.register(p, gv)
                if geom is not None:
                    kind, taps, O, I, ipad = geom
                    sink.register_matrix(p, o, O, taps * ipad, kind, taps)
                    sink.shadow_state[id(p)] = (p._version, ops._EPOCH_EXT)
            if hooked:
                hooks = [self._make_hook(i) for i in range(len(self.params))]
                sink.notify = lambda p: hooks[self._index[id(p)]](p)
            ops.SINK = sink

    def _make_hook(self, i):
        def hook(_p):
            if not self._sync_this_backward:
                return
            # a parameter can be announced twice in one backward: by the kernel that accumulated its gradient straight
            # into the flat buffer (GradSink.done) and again by autograd's post-accumulate hook; count it once
            if i in self._notified:
                return
            self._notified.add(i)
            b = self._bucket_of[i]
            b["pending"] -= 1
            if b["pending"] == 0:
                self._launch_bucket(b)
        return hook

    def begin_step(self, do_ema: int = 0, device_state: Optional[torch.Tensor] = None):
        """Announce the optimizer step that follows the coming (last micro-batch's) backward pass: with it the bucket
        hooks may issue that step slice by slice while backward is still running.  step() must then be called with the
        same arguments.  Without an announcement step() does the whole update itself, as before."""
        self._announced = (int(do_ema) if self.ema_module is not None else 0, device_state)

    def _adam_slice(self, lo: int, hi: int, do_ema: int, device_state, stream):
        """Fused Adam/EMA on flat[lo:hi) for the step in progress; never advances the device-side step counter."""
        b1, b2 = self.betas
        step = self.step_count + 1
        e = None if self.flat_ema is None else self.flat_ema.data_ptr() + 4 * lo
        # algorithmic bytes per element: read p, g, m, v (16), write p, m, v, zeroed g (16), bf16 copy (2); the EMA pass of
        # every 10th step (+8) is not counted
        ops._timed("adam_ema_kernel", 0.0, lambda: lib.call(
            "madiff_adam_ema", self.flat_p.data_ptr() + 4 * lo, self.flat_g.data_ptr() + 4 * lo,
            self.flat_m.data_ptr() + 4 * lo, self.flat_v.data_ptr() + 4 * lo, e, hi - lo, self.lr, b1, b2, self.eps,
            1 - b1 ** step, math.sqrt(1 - b2 ** step), self.grad_scale / self.world, do_ema, self.ema_decay,
            None if device_state is None else device_state.data_ptr(), self.flat_p16.data_ptr() + 2 * lo, 3,
            stream.cuda_stream), nbytes=34.0 * (hi - lo), stream=stream)

    def _order_behind_producers(self, tgt):
        """Make `tgt` wait for every stream a member of the bucket may have been produced on: the stream that is current
        now, the step's main stream and the parallel branches (ops.aux_stream)."""
        cur = torch.cuda.current_stream()
        for st in (cur, self._main_stream):
            if st is None or st.cuda_stream == tgt.cuda_stream:
                continue
            if ops.stream_is_capturing(tgt) and not ops.stream_is_capturing(st):
                continue
            ops.wait_stream(tgt, st)
        ops.join_aux_streams(tgt)

    def _reduce_begin(self, lo: int, hi: int):
        """Start the all-reduce of flat_g[lo:hi) on the current stream; returns the work handle."""
        g = self.flat_g[lo:hi]
        if self.bucket_dtype == "bf16":
            if self._stage16 is None:
                self._stage16 = torch.empty(self.total, dtype=torch.bfloat16, device=self.flat_g.device)
            st = self._stage16[lo:hi]
            st.copy_(g)                                   # fp32 -> bf16 (round to nearest even)
            return dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        return dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def _reduce_end(self, work, lo: int, hi: int):
        """Order the current stream behind the collective (no host wait on the GPU) and, for bf16 transport, bring the sum
        back into the fp32 master gradient."""
        work.wait()
        if self.bucket_dtype == "bf16":
            self.flat_g[lo:hi].copy_(self._stage16[lo:hi])

    def _launch_bucket(self, b):
        if not self.dp:
            # single process: nothing to reduce; the bucket's optimizer slice goes onto the "wgrad" branch
            if self._announced is None or not self.use_hip or ops.TIMER is not None:
                return                        # (per-launch event timing: keep the timed kernels alone on the GPU)
            cur = torch.cuda.current_stream()
            tgt = ops.aux_stream(cur.device, "wgrad") if ops.WGRAD_SIDE_STREAM else cur
            self._order_behind_producers(tgt)  # every member's data-gradient launch (it reads the weights) is behind us
            self._adam_slice(b["lo"], b["hi"], *self._announced, tgt)
            b["opt"] = True
            return
        if self._graph_events is not None and torch.cuda.is_current_stream_capturing():
            # Capturing a data-parallel step (GraphedTrainStep): no collective goes into the graph.  The bucket's
            # completion becomes a graph-external event node; after each replay the communication stream waits on it and
            # runs the all-reduce eagerly, overlapped with the rest of the replayed backward pass.
            # The event node goes onto the "wgrad" auxiliary stream (where most members are produced anyway) after that
            # stream has waited for the issuing stream and the other branches: edges then only ever run INTO the wgrad
            # branch, which rejoins the main branch once, at the end of the step.  (Joining the branches into the
            # main stream at every bucket instead made hipStreamEndCapture crash.)
            cur = torch.cuda.current_stream()
            tgt = ops.aux_stream(cur.device, "wgrad") if (ops.WGRAD_SIDE_STREAM and not self.bucket_events_on_main) else cur
            self._order_behind_producers(tgt)         # members may have been produced on a parallel branch
            bi = self.buckets.index(b)
            lib.call("madiff_event_record_external", self._graph_events[bi], tgt.cuda_stream)
            self._graph_order.append(bi)
            b["work"] = "graph"
            return
        if self._comm_stream is not None:
            self._order_behind_producers(self._comm_stream)   # members may have been produced on a side branch
            with torch.cuda.stream(self._comm_stream):
                b["work"] = self._reduce_begin(b["lo"], b["hi"])
                if self._announced is not None and self.use_hip and self.overlap and ops.TIMER is None:
                    self._reduce_end(b["work"], b["lo"], b["hi"])
                    self._adam_slice(b["lo"], b["hi"], *self._announced, self._comm_stream)
                    b["work"], b["opt"] = "done", True
        else:
            b["work"] = self._reduce_begin(b["lo"], b["hi"])

    # ------------------------------------------------------------------ per-step API
    def zero_grad(self):
        if self.flat_g is not None:
            if not getattr(self, "_grads_zeroed", False):   # the fused Adam kernel clears the gradient as it reads it
                self.flat_g.zero_()
            self._grads_zeroed = False
        else:
            for p in self.module.parameters():
                p.grad = None

    def set_sync(self, flag: bool):
        """Only the last micro-batch of an accumulation group triggers the all-reduce (no_sync equivalent)."""
        self._sync_this_backward = flag
        self._notified = set()
        # the stream the step is issued on: a bucket may be completed from code running on a parallel branch (the
        # inverse-dynamics backward runs on its own stream) while other members were produced on this one
        self._main_stream = torch.cuda.current_stream() if (self.flat_p is not None and self.flat_p.is_cuda) else None
        if flag:
            for b in self.buckets:
                b["pending"] = len(b["members"])
                b["work"] = None
                b["opt"] = False

    # -- data-parallel step replayed from a hipGraph (GraphedTrainStep) ---------------------------------------
    def begin_graph_capture(self):
        import ctypes
        assert self.dp and self._comm_stream is not None and self.flat_g is not None
        if self._graph_events is None:
            self._graph_events = {}
            for bi in range(len(self.buckets)):
                h = ctypes.c_void_p()
                lib.call("madiff_event_create", ctypes.byref(h))
                self._graph_events[bi] = h.value
        self._graph_order = []
        self.set_sync(True)

    def allreduce_after_replay(self, do_ema: int = 0):
        """Called right after graph.replay(): start every bucket's all-reduce on the communication stream as soon as
        its event node inside the running graph has fired, with the bucket's optimizer slice right behind it (both
        overlap the rest of the replayed backward); the caller's stream then waits for all of them."""
        assert self._graph_order, "no bucket completion was recorded while capturing"
        cs = self._comm_stream
        if self.ema_module is None:
            do_ema = 0
        fused = self.overlap and self.use_hip
        for b in self.buckets:
            b["opt"] = False
        with torch.cuda.stream(cs):
            for bi in self._graph_order:
                b = self.buckets[bi]
                lib.call("madiff_stream_wait_event", cs.cuda_stream, self._graph_events[bi])
                self._reduce_end(self._reduce_begin(b["lo"], b["hi"]), b["lo"], b["hi"])
                if fused:
                    self._adam_slice(b["lo"], b["hi"], do_ema, None, cs)
                    b["opt"] = True
                b["work"] = "done"
        torch.cuda.current_stream().wait_stream(cs)
        if fused:
            self._announced = (do_ema, None)

    def _finish_allreduce(self):
        if not self.dp:
            return
        if not self._hooks:            # first step: buffers were created after backward -> reduce everything now
            self._reduce_end(self._reduce_begin(0, self.total), 0, self.total)
            return
        for b in self.buckets:
            if b["work"] == "done":    # reduced by allreduce_after_replay()
                continue
            if b["work"] is None:      # a bucket whose hooks did not all fire (should not happen) -> reduce now
                self._launch_bucket(b)
            if b["work"] != "done":
                if self._comm_stream is not None:
                    with torch.cuda.stream(self._comm_stream):
                        self._reduce_end(b["work"], b["lo"], b["hi"])
                else:
                    self._reduce_end(b["work"], b["lo"], b["hi"])
        if self._comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self._comm_stream)

    def step(self, do_ema: int = 0, device_state: Optional[torch.Tensor] = None):
        """do_ema: 0 none, 1 lerp with ema_decay, 2 copy (reference: before step_start_ema).
        device_state (int32[2] on the GPU = {completed steps, do_ema}): the bias corrections and the EMA mode are taken
        from the device instead of the arguments, so the launch can live inside a replayed hipGraph (GraphedTrainStep)."""
        first = self.flat_p is None
        if first:
            self._materialise()
            self._announced = None               # (the hooks did not exist during this backward: nothing was issued early)
        if self.flat_p.is_cuda:                      # join the parallel branches of the step (ops.aux_stream)
            ops.join_aux_streams()
        self._finish_allreduce()
        if self.ema_module is None:
            do_ema = 0
        ann, self._announced = self._announced, None
        early = [] if first else [b for b in self.buckets if b.get("opt")]
        if early and ann is not None and (ann[0] != int(do_ema) or
                                          (ann[1] is None) != (device_state is None)):
            raise RuntimeError("FlatAdamEMA.step(): arguments differ from what begin_step() announced")
        b1, b2 = self.betas
        scale = self.grad_scale / self.world
        if self.use_hip:
            cur = torch.cuda.current_stream()
            if not early:                            # nothing was issued during backward: one launch over everything
                lib.call("madiff_adam_ema", self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.flat_m.data_ptr(),
                         self.flat_v.data_ptr(), None if self.flat_ema is None else self.flat_ema.data_ptr(), self.total,
                         self.lr, b1, b2, self.eps, 1 - b1 ** (self.step_count + 1),
                         math.sqrt(1 - b2 ** (self.step_count + 1)), scale, do_ema, self.ema_decay,
                         None if device_state is None else device_state.data_ptr(),
                         self.flat_p16.data_ptr(), 1, cur.cuda_stream)
            else:                                    # the remaining slices, then advance the device-side counter once
                for b in self.buckets:
                    if not b.get("opt"):
                        self._adam_slice(b["lo"], b["hi"], int(do_ema), device_state, cur)
                if device_state is not None:
                    lib.call("madiff_adam_ema", None, None, None, None, None, 0, self.lr, b1, b2, self.eps, 1.0, 1.0, scale,
                             0, self.ema_decay, device_state.data_ptr(), None, 1, cur.cuda_stream)
            for b in self.buckets:
                b["opt"] = False
            self.step_count += 1
            self._grads_zeroed = True
            ops.params_changed(by_optimizer=True)
        else:   # host arithmetic for the gloo/CPU tests of the data-parallel machinery only
            self.step_count += 1
            bc1 = 1 - b1 ** self.step_count
            bc2s = math.sqrt(1 - b2 ** self.step_count)
            with torch.no_grad():
                g = self.flat_g * scale
                self.flat_m.mul_(b1).add_(g, alpha=1 - b1)
                self.flat_v.mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = (self.flat_v.sqrt() / bc2s).add_(self.eps)
                self.flat_p.addcdiv_(self.flat_m, denom, value=-self.lr / bc1)
                if do_ema == 1:
                    self.flat_ema.mul_(self.ema_decay).add_(self.flat_p, alpha=1 - self.ema_decay)
                elif do_ema == 2:
                    self.flat_ema.copy_(self.flat_p)

    def broadcast_parameters(self, src: int = 0):
        """Make every rank start from rank `src`'s weights (the reference has one process; DP needs this once)."""
        if self.world <= 1:
            return
        for p in self.module.parameters():
            dist.broadcast(p.data, src=src, group=self.pg)
        if self.ema_module is not None:
            for p in self.ema_module.parameters():
                dist.broadcast(p.data, src=src, group=self.pg)
        ops.params_changed()


FOLD_MICRO_BATCHES = os.environ.get("MADIFF_FOLD_MICRO", "1") != "0"


def can_fold(diffusion, batches) -> bool:
    """May the micro-batches of one optimizer step run as ONE pass over their concatenation?  Needs the fused GPU loss
    (it normalises the inverse-dynamics term per micro-batch, see GaussianDiffusion.loss(micro_batches=)) and equal sizes."""
    from . import diffusion as D
    if not FOLD_MICRO_BATCHES or len(batches) < 2 or len(batches) > 64 or not D.FUSED_LOSS:
        return False
    if getattr(diffusion, "train_only_inv", False) or not (getattr(diffusion, "share_inv", False)
                                                           or getattr(diffusion, "joint_inv", False)):
        return False
    x0 = batches[0]["x"]
    if not (x0.is_cuda and x0.dtype == torch.float32):
        return False
    def same(vals):
        """Every micro-batch holds the same kind of leaf here: all None, or tensors of one shape / dtype (nested dicts
        likewise).  Anything else (a Python scalar, a ragged field) is left to the unfolded loop, which accepts it."""
        v0 = vals[0]
        if v0 is None:
            return all(v is None for v in vals)
        if isinstance(v0, dict):
            return (all(isinstance(v, dict) and set(v.keys()) == set(v0.keys()) for v in vals)
                    and all(same([v[k] for v in vals]) for k in v0))
        return torch.is_tensor(v0) and all(torch.is_tensor(v) and v.shape == v0.shape and v.dtype == v0.dtype
                                           and v.device == v0.device for v in vals)

    return same(list(batches))


def fold_batches(batches):
    """Concatenate batch dicts (nested dicts of tensors) along the batch axis."""
    def cat(vals):
        if vals[0] is None:                 # a field every micro-batch leaves out (legal_actions=None, env_ts=None)
            return None
        if isinstance(vals[0], dict):
            return {k: cat([v[k] for v in vals]) for k in vals[0]}
        return torch.cat(vals, dim=0)
    return {k: cat([b[k] for b in batches]) for k in batches[0]}


def train_step(diffusion, opt: FlatAdamEMA, batches, do_ema: int = 0, fold: Optional[bool] = None):
    """One optimizer step over `len(batches)` micro-batches (training.py:147-158).  Returns the last (loss, infos).
    fold (default: when possible): one forward/backward over the concatenated micro-batches, same gradient."""
    n = len(batches)
    if (fold is None or fold) and can_fold(diffusion, batches):
        opt.set_sync(True)
        opt.begin_step(do_ema)
        loss, infos = diffusion.loss(**fold_batches(batches), micro_batches=n)
        loss.backward()
        opt.step(do_ema)
        opt.zero_grad()
        return infos.pop("micro_loss").detach(), infos
    assert not fold, "these micro-batches cannot be folded (can_fold)"
    out = None
    for i, batch in enumerate(batches):
        opt.set_sync(i == n - 1)
        if i == n - 1:
            opt.begin_step(do_ema)
        loss, infos = diffusion.loss(**batch)
        (loss / n).backward()
        out = (loss.detach(), infos)
    opt.step(do_ema)
    opt.zero_grad()
    return out


class GraphedTrainStep:
    """One optimizer step (loss forward, backward, fused Adam/EMA) captured ONCE into a hipGraph and replayed: the
    ~300 kernel launches of a step then cost one host call, which matters because at batch 256 the GPU finishes a
    step faster than Python can issue it.  Same arithmetic as train_step: the graph holds exactly the launches an eager
    step makes.  What changes per step lives in static device buffers refreshed before each replay:
      * the batch (copied into the captured input tensors),
      * the classifier-free-guidance dropout mask, drawn like the reference on the CPU generator
        (ma_temporal.py:346-351) and copied over,
      * {completed steps, EMA mode} for the Adam kernel's bias corrections (device int32[2]).
    The diffusion timestep t and the noise are drawn inside the graph by torch's graph-safe Philox generator, in the
    reference's order.  Data-parallel jobs (world_size > 1): the graph holds loss forward + backward only and every
    gradient bucket's completion is a graph-EXTERNAL event node; after each replay the communication stream waits on
    those events and all-reduces the buckets eagerly (no collective inside the graph), overlapped with the rest of the
    replayed backward pass, then the fused Adam/EMA launch follows eagerly.
    Call after at least one eager train_step (buffers materialised, GEMM variants tuned)."""

    def __init__(self, diffusion, opt: FlatAdamEMA, example_batch, draws=None, warm_do_ema: int = 0,
                 fold: Optional[bool] = None):
        """example_batch: one batch dict, or a list of them = the micro-batches of one optimizer step (the reference's
        gradient_accumulate_every, training.py:147-152; each loss is scaled by 1/len before backward).
        draws = {"t", "noise", "dropout_mask"} (or one such dict per micro-batch) fixes the random draws of every
        replay (parity tests).  The constructor performs ONE real optimizer step on example_batch, eagerly on the
        capture stream (allocator / autograd warm-up), with EMA mode warm_do_ema; its (loss, infos) is `self.first`."""
        assert opt.flat_p is not None and opt.use_hip
        self._warm_ema = int(warm_do_ema)
        self.dp = opt.dp
        self.single = isinstance(example_batch, dict)
        examples = [example_batch] if self.single else list(example_batch)
        n = len(examples)
        if draws is not None:
            draws = [draws] * n if isinstance(draws, dict) else list(draws)
            assert len(draws) == n
        self.draws = None if draws is None else [{k: v.clone() for k, v in d.items()} for d in draws]
        self.diffusion, self.opt = diffusion, opt
        clone = lambda v: {k: clone(u) for k, u in v.items()} if isinstance(v, dict) else v.clone()  # noqa: E731
        # fold: the n micro-batches live in ONE concatenated static batch (self.big) and the step is one forward/backward
        # over it (GaussianDiffusion.loss(micro_batches=n)): same gradient as n accumulated passes, a third of the launches
        # at n = 2 -- what matters at the reference's own setting (batch 32 x accumulate 2), which is launch-latency bound
        self.fold = n > 1 and (fold is None or fold) and can_fold(diffusion, examples)
        assert not (fold and not self.fold), "these micro-batches cannot be folded (can_fold)"
        if self.fold:
            self.big = fold_batches(examples)
            Bm = examples[0]["x"].shape[0]
            view = lambda v, i: ({k: view(u, i) for k, u in v.items()} if isinstance(v, dict)      # noqa: E731
                                 else v[i * Bm:(i + 1) * Bm])
            self.batches = [view(self.big, i) for i in range(n)]
        else:
            self.batches = [clone(b) for b in examples]
        self.batch = self.batches[0]
        dev = opt.flat_p.device
        net = diffusion.model
        self.mask_dist = net.net.mask_dist if getattr(net, "returns_condition", False) else None
        B = examples[0]["x"].shape[0]
        assert all(b["x"].shape[0] == B for b in examples)
        self.masks = torch.ones(n, B, net.n_agents, 1, device=dev)
        self.mask = self.masks[0]
        if self.draws is not None:
            self.mask_dist = None
            for i, d in enumerate(self.draws):
                self.masks[i].copy_(d["dropout_mask"].reshape(self.masks[i].shape))
        # host staging for the per-step masks: a ring of pinned buffers, each guarded by an event, because the host runs
        # several replays ahead of the GPU and must not overwrite a buffer whose async copy has not executed yet
        self._ring = [(torch.ones(n, B, net.n_agents, 1).pin_memory(), torch.cuda.Event()) for _ in range(4)]
        self._ring_pos = 0
        self.state = torch.tensor([opt.step_count, 0], dtype=torch.int32, device=dev)   # {completed steps, EMA mode}
        assert ops.TIMER is None, "per-launch event timing cannot be captured"
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        self._refresh_masks()
        with torch.cuda.stream(side):          # one more eager step on the capture stream (allocator / autograd warm-up)
            self.first = self._body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        if self.dp:
            opt.begin_graph_capture()
        else:
            opt.begin_step(0, device_state=self.state)
        with ops.graph_capture(self.graph):
            self.loss, self.infos = self._body(captured=True)
        if self.dp:
            assert sorted(opt._graph_order) == list(range(len(opt.buckets))), "a gradient bucket never completed"
        # the step counter lives on the device from here on: the captured Adam launch reads it and a captured
        # one-thread kernel increments it after every replay
        self.state.copy_(torch.tensor([opt.step_count, 0], dtype=torch.int32))

    def _body(self, captured: bool = False):
        n = len(self.batches)
        if self.fold:
            self.opt.set_sync(True)
            if not captured:
                self.opt.begin_step(self._warm_ema)
            kw = dict(self.big)
            kw.setdefault("dropout_mask", self.masks.view(n * self.masks.shape[1], *self.masks.shape[2:]))
            if self.draws is not None:
                kw["t"] = torch.cat([d["t"] for d in self.draws])
                kw["noise"] = torch.cat([d["noise"] for d in self.draws])
            loss, infos = self.diffusion.loss(**kw, micro_batches=n)
            loss.backward()
            loss = infos.pop("micro_loss")           # what the reference's loop logs: the last micro-batch's loss
        for i, batch in enumerate(() if self.fold else self.batches):
            extra = {} if self.draws is None else {"t": self.draws[i]["t"], "noise": self.draws[i]["noise"]}
            self.opt.set_sync(i == n - 1)          # bucket hooks (all-reduce / optimizer slices): last micro-batch only
            if i == n - 1 and not captured:
                self.opt.begin_step(self._warm_ema)
            kw = dict(batch)
            kw.setdefault("dropout_mask", self.masks[i])     # a batch may carry its own explicit draws (parity tests)
            kw.update(extra)
            loss, infos = self.diffusion.loss(**kw)
            (loss if n == 1 else loss / n).backward()
        if captured and self.dp:
            # data parallel: the graph ends with backward; all-reduce and Adam are issued after each replay
            ops.join_aux_streams()
        elif captured:
            self.opt.step(device_state=self.state)      # launches only; host-side counters advance in step()
            self.opt.step_count -= 1
            self.opt.zero_grad()
        else:
            self.opt.step(self._warm_ema)
            self.opt.zero_grad()
        return loss.detach(), {k: (v.detach() if torch.is_tensor(v) else v) for k, v in infos.items()}

    def _refresh_masks(self):
        """New classifier-free-guidance dropout masks for the coming step: one draw of shape (B, A, 1) per micro-batch,
        in order, on the CPU generator, as the reference's forward makes them (ma_temporal.py:346-351)."""
        if self.mask_dist is None:
            return
        host, ev = self._ring[self._ring_pos]
        self._ring_pos = (self._ring_pos + 1) % len(self._ring)
        ev.synchronize()
        for i in range(host.shape[0]):
            host[i].copy_(self.mask_dist.sample(sample_shape=tuple(host.shape[1:3]) + (1,)).reshape(host.shape[1:]))
        self.masks.copy_(host, non_blocking=True)
        ev.record()

    def _copy_in(self, dst, src):
        if isinstance(dst, dict):
            for k in dst:
                self._copy_in(dst[k], src[k])
        elif dst.data_ptr() != src.data_ptr():
            dst.copy_(src, non_blocking=True)

    def step(self, batch=None, do_ema: int = 0):
        """Replays the captured step on `batch` (a dict, or the list of micro-batches; None: what the static buffers
        already hold)."""
        if batch is not None:
            given = [batch] if isinstance(batch, dict) else list(batch)
            assert len(given) == len(self.batches)
            for dst, src in zip(self.batches, given):
                self._copy_in(dst, src)
        self._refresh_masks()
        if self.opt.ema_module is None:
            do_ema = 0
        if self.dp:
            self.graph.replay()
            self.opt.allreduce_after_replay(do_ema)
            self.opt.step(do_ema)                  # eager fused Adam/EMA (host-side step counter), clears the gradient
            self.opt.zero_grad()
            return self.loss, self.infos
        self.state[1:].fill_(int(do_ema))      # the value travels as a kernel argument: no host buffer to race on
        self.graph.replay()
        self.opt.step_count += 1
        ops.params_changed(by_optimizer=True)
        return self.loss, self.infos
