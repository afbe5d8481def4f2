"""torch.autograd.Functions over the C ABI of libmadiff_hip.so.

Internal activation format ("cl"): a contiguous 2-D tensor [rows = (sequence n, time t), pad32(C)] in the
build's storage dtype (bf16 or fp32) whose pad columns are zero.  Parameters stay ordinary fp32
nn.Parameters with the reference's shapes; GEMM-layout copies are cached and refreshed when a
parameter's version counter changes.  Every op raises if the HIP library is missing (madiff_amd.lib
fails at import) - there is no PyTorch/CPU fallback on this path.
"""
from __future__ import annotations

import os
import weakref
from typing import List, Optional, Sequence, Tuple

import torch

from . import lib

# ------------------------------------------------------------------------------------------------
# precision ("build") selection
# ------------------------------------------------------------------------------------------------
_PRECISION = os.environ.get("MADIFF_PRECISION", "bf16")


def set_precision(name: str) -> None:
    """'bf16' (bf16 storage + bf16 MFMA, fp32 accumulate) or 'fp32' (fp32 storage + exact fp32 MFMA)."""
    global _PRECISION
    assert name in ("bf16", "fp32")
    if name != _PRECISION:
        global _TUNE
        _PRECISION = name
        _PACK_CACHE.clear()
        # kernel-variant choices are per build: a (variant, splits) pair tuned for the bf16 kernels names variants the
        # fp32 build does not have (its launch would be refused and the gradient never written)
        _TUNE = _TUNE_BY_PRECISION.setdefault(name, {})


def get_precision() -> str:
    return _PRECISION


def act_dtype() -> torch.dtype:
    return torch.bfloat16 if _PRECISION == "bf16" else torch.float32


def _dt() -> int:
    return lib.BF16 if _PRECISION == "bf16" else lib.F32


_PAD64 = os.environ.get("MADIFF_PAD64", "1") != "0"


def pad32(c: int) -> int:
    """Channel padding of the cl format: a multiple of 32 -- and of 64 up to 128 channels, so that the narrow layers (the
    trajectory features F = 16 .. 88 of the first convolution and its residual 1x1, the inverse-dynamics input 2F) have a
    reduction dimension made of whole 64-channel chunks and run on the LDS-DMA kernels instead of the register-staged
    generic one (6 launches of ~24 us per training step before).  Pad columns are zero; the cost is a few zero columns of
    the narrowest tensors."""
    return (c + 63) // 64 * 64 if (c <= 128 and _PAD64) else (c + 31) // 32 * 32


_SPLITK_WS: dict = {}          # device index -> (workspace, counters) registered with the library (kept alive here)
SPLITK_WS_MB = int(os.environ.get("MADIFF_SPLITK_WS_MB", "64"))
SPLITK_WS_SLOTS = 16           # = MADIFF_WS_SLOTS of csrc/gemm.hip


def _ensure_splitk_workspace() -> None:
    """Register the all-taps kernel's reduction-split workspace for the current device, once (madiff_set_splitk_workspace).
    Not under graph capture: the first eager call does it (every product path warms up eagerly before capturing)."""
    dev = torch.cuda.current_device()
    if dev in _SPLITK_WS or SPLITK_WS_MB <= 0 or torch.cuda.is_current_stream_capturing():
        return
    # SPLITK_WS_SLOTS slots of SPLITK_WS_MB each: the library hands one slot to every stream / capture that uses the split
    # (launches that may overlap never share partial tiles or tickets, csrc/gemm.hip attach_ws)
    ws = torch.empty((SPLITK_WS_MB << 20) * SPLITK_WS_SLOTS, dtype=torch.uint8, device=f"cuda:{dev}")
    cnt = torch.zeros(1024 * SPLITK_WS_SLOTS, dtype=torch.int32, device=f"cuda:{dev}")
    lib.call("madiff_set_splitk_workspace", ws.data_ptr(), ws.numel(), cnt.data_ptr(), cnt.numel())
    _SPLITK_WS[dev] = (ws, cnt)


def reset_splitk_slots() -> None:
    """Hand every slot of the reduction-split workspace of the current device back (the library never does: a slot belongs
    to one stream or one captured branch for good, and when the 16 are used up later captures run unsplit -- same
    results to fp32 rounding of the partial sums, slower small-batch layers).  For a process that has dropped its captured
    graphs, and for the test suite (every test starts with all slots, so that "graph == eager, bit for bit" does not
    depend on how many captures ran before it).  Graphs captured earlier keep using their old slots: do not replay them
    beside new work after this call."""
    dev = torch.cuda.current_device()
    if dev not in _SPLITK_WS:
        return
    torch.cuda.synchronize()
    ws, cnt = _SPLITK_WS[dev]
    cnt.zero_()
    lib.call("madiff_set_splitk_workspace", ws.data_ptr(), ws.numel(), cnt.data_ptr(), cnt.numel())
    torch.cuda.synchronize()


def _stream() -> int:
    if torch.cuda.current_device() not in _SPLITK_WS:
        _ensure_splitk_workspace()
    return torch.cuda.current_stream().cuda_stream


AUX_STREAMS: dict = {}


def aux_stream(device, name: str = "branch") -> "torch.cuda.Stream":
    """The auxiliary streams (one per device and name) on which independent branches of a step are issued: "branch" = the
    inverse-dynamics loss, "wgrad" = parameter gradients.  The data-parallel optimizer waits on all of them, besides the
    current stream, before it reduces a gradient bucket, and the optimizer step joins them."""
    key = (torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device(), name)
    st = AUX_STREAMS.get(key)
    if st is None:
        st = AUX_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


def stream_is_capturing(st) -> bool:
    with torch.cuda.stream(st):
        return torch.cuda.is_current_stream_capturing()


def wait_stream(waiter, waited) -> None:
    """waiter.wait_stream(waited), safe inside a capture.  When both streams already belong to the same capture the
    dependency is added by hand (madiff_capture_join): on this runtime side streams of a capture that wait on each
    other through events make hipStreamEndCapture recurse until the stack overflows.  Outside a
    capture, and when `waiter` is about to JOIN the capture, it is the ordinary event record + wait."""
    if waiter.cuda_stream == waited.cuda_stream:
        return
    rc = lib.call_rc("madiff_capture_join", waiter.cuda_stream, waited.cuda_stream)     # (raises on a HIP error)
    if rc == 0:
        return
    # 1 = the two are not (both, actively) in one capture: the event path -- unless that would reach the runtime with
    # something it answers by aborting the process
    s1 = lib._lib.madiff_stream_capture_status(waiter.cuda_stream)
    s2 = lib._lib.madiff_stream_capture_status(waited.cuda_stream)
    if s1 == 2 or s2 == 2 or s1 < 0 or s2 < 0:
        raise RuntimeError(f"wait_stream: the stream capture has been invalidated (status {s1}, {s2}); the step cannot be captured")
    if s1 == 1 and s2 == 0:
        raise RuntimeError("wait_stream: a capturing stream was asked to wait for a stream outside its capture")
    if s1 == 1 and s2 == 1:
        raise RuntimeError("wait_stream: the two streams belong to different captures")
    waiter.wait_stream(waited)


def join_aux_streams(target=None):
    """Make `target` (default: the current stream) wait for everything issued so far on the auxiliary streams.  While
    `target` is capturing, only the auxiliary streams that have joined the same capture are waited for: an event
    recorded on a stream outside the capture must not be waited on from inside it (the runtime crashes), and such a
    stream holds no work of the step being captured anyway."""
    cur = torch.cuda.current_stream()
    target = cur if target is None else target
    for st in AUX_STREAMS.values():
        if st.cuda_stream == target.cuda_stream:
            continue
        # asked again for every stream: a target that was outside the capture JOINS it with its first wait on a capturing
        # stream, and from then on must not wait on a stream outside it (round 3: with eight weight-gradient branches a
        # small model leaves some of them unused; the state taken once before the loop let the freshly joined target wait
        # on such a stream's event -> abort inside hipStreamWaitEvent, seen once in four full test runs)
        if stream_is_capturing(target) and not stream_is_capturing(st):
            continue
        wait_stream(target, st)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _chk_cl(x: torch.Tensor):
    assert x.is_cuda and x.dim() == 2 and x.stride(1) == 1 and x.dtype == act_dtype(), (x.shape, x.stride(), x.dtype)
    assert x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0


# ------------------------------------------------------------------------------------------------
# weight packing cache
# ------------------------------------------------------------------------------------------------
class ConvSpec:
    """Geometry of one implicit-GEMM layer.  kind: 'conv' (nn.Conv1d [O, I, k]), 'convT' (nn.ConvTranspose1d
    [I, O, k]) or 'linear' ([O, I])."""

    __slots__ = ("kind", "taps", "stride", "pad", "tr")

    def __init__(self, kind="linear", taps=1, stride=1, pad=0, tr=0):
        self.kind, self.taps, self.stride, self.pad, self.tr = kind, taps, stride, pad, tr

    # Compared BY VALUE: modules keep a spec as an attribute, and copy.deepcopy(model) -- how the reference makes its EMA
    # model (utils/training.py:57) -- or unpickling a whole module gives them a new object.  Round 3 found the identity test
    # `spec is CONV5` silently keeping every deep-copied model (i.e. every EMA sampler after training) off the fused
    # conv+GroupNorm launch and the ResBlock node.
    def _key(self):
        return (self.kind, self.taps, self.stride, self.pad, self.tr)

    def __eq__(self, other):
        return isinstance(other, ConvSpec) and self._key() == other._key()

    def __ne__(self, other):
        return not self.__eq__(other)

    def __hash__(self):
        return hash(self._key())

    def __deepcopy__(self, memo):
        return self                      # immutable

    def out_len(self, t_in: int) -> int:
        if self.tr:
            return (t_in - 1) * self.stride - 2 * self.pad + self.taps
        return (t_in + 2 * self.pad - self.taps) // self.stride + 1


CONV5 = ConvSpec("conv", 5, 1, 2, 0)
CONV1 = ConvSpec("conv", 1, 1, 0, 0)
DOWN = ConvSpec("conv", 3, 2, 1, 0)
UP = ConvSpec("convT", 4, 2, 1, 1)
LINEAR = ConvSpec("linear", 1, 1, 0, 0)


def _strides(kind: str, w: torch.Tensor):
    """(O, I, so, si, st): element (o, i, tap) of the GEMM view lives at o*so + i*si + tap*st of the parameter."""
    if kind == "conv":
        O, I, k = w.shape
        return O, I, w.stride(0), w.stride(1), w.stride(2)
    if kind == "convT":
        I, O, k = w.shape
        return O, I, w.stride(1), w.stride(0), w.stride(2)
    O, I = w.shape
    return O, I, w.stride(0), w.stride(1), 0


_PACK_CACHE: dict = {}
_EPOCH = 0        # bumped whenever parameter memory was written behind autograd's back
_EPOCH_EXT = 0    # ... by anything but the fused optimizer (which keeps its bf16 shadow weights in sync itself)


def params_changed(by_optimizer: bool = False) -> None:
    """Tell the pack cache that parameter memory was written behind autograd's back (e.g. by the fused
    Adam kernel, which does not bump torch's version counters).  by_optimizer=True: the writer was
    madiff_adam_ema, which refreshed the bf16 shadow of the flat buffer in the same pass."""
    global _EPOCH, _EPOCH_EXT
    _EPOCH += 1
    if not by_optimizer:
        _EPOCH_EXT += 1


PACK_EPB = 2048          # elements per block of madiff_pack_weights_multi (csrc/gemm.hip)
_MULTI = {"sig": None, "table": None, "n": 0, "blocks": 0}


class _PackEntry:
    __slots__ = ("vers", "buf", "refs", "jobs", "ready")     # ready: (event, stream id) of the launch that last wrote buf


def _mark_ready(entries):
    """Entries were just (re)packed on the current stream: a reader on another stream (a parallel branch of the step,
    ops.aux_stream) has to wait for that launch, which _packed() arranges through this event."""
    if not entries or not entries[0].buf.is_cuda:
        return
    ev = torch.cuda.Event()
    ev.record()
    sid = torch.cuda.current_stream().cuda_stream
    for e in entries:
        e.ready = (ev, sid)


def _entry_vers(weights):
    return (_EPOCH,) + tuple((w._version, w.data_ptr()) for w in weights)


def _pack_jobs(weights, spec: ConvSpec, backward: bool, buf):
    """[(src_ptr, dst_ptr, O, I, taps, so, si, st, dst_ld, dst_off)] for one cache entry."""
    dims = [_strides(spec.kind, w) for w in weights]
    I = dims[0][1]
    Osum = sum(d[0] for d in dims)
    es = buf.element_size()
    jobs, off = [], 0
    for w, (O, I_, so, si, st) in zip(weights, dims):
        assert I_ == I and w.dtype == torch.float32
        if backward:   # rows = i, cols = (tap, o): swap the roles of o and i
            jobs.append((w.data_ptr(), buf.data_ptr(), I, O, spec.taps, si, so, st, pad32(Osum), off))
        else:
            jobs.append((w.data_ptr(), buf.data_ptr() + off * spec.taps * pad32(I) * es, O, I, spec.taps, so, si, st,
                         pad32(I), 0))
        off += O
    return jobs


def _refresh_stale(trigger_key):
    """Repack every stale cache entry whose parameters are alive with ONE multi-tensor launch (train steps make
    all ~150 entries stale at once).  Falls back to per-entry launches for a single stale entry."""
    stale = []
    for key, ent in list(_PACK_CACHE.items()):
        ws = [r() for r in ent.refs]
        if any(w is None for w in ws):
            continue
        v = _entry_vers(ws)
        if v != ent.vers:
            stale.append((key, ent, ws, v))
    if not stale:
        return
    if len(stale) == 1:
        key, ent, ws, v = stale[0]
        for j in ent.jobs(ws):
            lib.call("madiff_pack_weight", _dt(), j[0], j[1], j[2], j[3], j[4], j[5], j[6], j[7], j[8], j[9], _stream())
        ent.vers = v
        _mark_ready([ent])
        return
    rows, blk = [], 0
    for key, ent, ws, v in stale:
        for j in ent.jobs(ws):
            n = j[2] * j[3] * j[4]
            rows.append(list(j) + [n, blk])
            blk += (n + PACK_EPB - 1) // PACK_EPB
    sig = tuple(tuple(r) for r in rows)
    if _MULTI["sig"] != sig and torch.cuda.is_current_stream_capturing():
        for key, ent, ws, v in stale:          # no host-to-device table upload inside a capture: one launch per job
            for j in ent.jobs(ws):
                lib.call("madiff_pack_weight", _dt(), j[0], j[1], j[2], j[3], j[4], j[5], j[6], j[7], j[8], j[9], _stream())
            ent.vers = v
        _mark_ready([e_[1] for e_ in stale])
        return
    if _MULTI["sig"] != sig:
        dev = stale[0][1].buf.device
        _MULTI["table"] = torch.tensor(rows, dtype=torch.int64).to(dev)
        _MULTI["sig"], _MULTI["n"], _MULTI["blocks"] = sig, len(rows), blk
    lib.call("madiff_pack_weights_multi", _dt(), _MULTI["table"].data_ptr(), _MULTI["n"], _MULTI["blocks"], _stream())
    for key, ent, ws, v in stale:
        ent.vers = v
    _mark_ready([e_[1] for e_ in stale])


def _packed(weights: Sequence[torch.Tensor], spec: ConvSpec, backward: bool) -> torch.Tensor:
    """GEMM-layout copy of a group of parameters that share the input (concatenated along the output dim).

    forward : [sum O, taps * pad32(I)]        backward (data gradient): [I, taps * pad32(sum O)]
    """
    if not backward and SINK is not None:
        m = SINK.weight_matrix(weights, spec)
        if m is not None:
            return m
    key = (tuple(id(w) for w in weights), spec.kind, spec.taps, backward, _PRECISION)
    ent = _PACK_CACHE.get(key)
    if ent is not None and not all(r() is w for r, w in zip(ent.refs, weights)):
        ent = None                      # id() of a dead parameter was reused by a new object
    if ent is None:
        dims = [_strides(spec.kind, w) for w in weights]
        I = dims[0][1]
        Osum = sum(d[0] for d in dims)
        shape = (I, spec.taps * pad32(Osum)) if backward else (Osum, spec.taps * pad32(I))
        ent = _PackEntry()
        ent.buf = torch.zeros(shape, dtype=act_dtype(), device=weights[0].device)
        ent.vers = None
        ent.ready = None
        ent.refs = tuple(weakref.ref(w, lambda _r, k=key: _PACK_CACHE.pop(k, None)) for w in weights)
        buf = ent.buf
        ent.jobs = lambda ws, spec=spec, backward=backward, buf=buf: _pack_jobs(ws, spec, backward, buf)
        _PACK_CACHE[key] = ent
    if ent.vers != _entry_vers(weights):
        _refresh_stale(key)
    rd = ent.ready
    if rd is not None and rd[1] != torch.cuda.current_stream().cuda_stream:
        torch.cuda.current_stream().wait_event(rd[0])
    return ent.buf


def clear_pack_cache() -> None:
    _PACK_CACHE.clear()


# ------------------------------------------------------------------------------------------------
# optional per-launch timing with HIP events on the launch stream (bench.py's roofline figures)
# ------------------------------------------------------------------------------------------------
class KernelTimer:
    """Collects (kernel name, algorithmic FLOPs, algorithmic bytes, start event, stop event) per launch while installed."""

    def __init__(self):
        self.records = []

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, flops, nbytes, e0, e1 in self.records:
            d = out.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += flops
            d["bytes"] += nbytes
        return out


TIMER: Optional[KernelTimer] = None


def _gemm_label(cin: int, csplit: int, out_f32: bool) -> str:
    """Kernel template madiff_gemm_fwd dispatches to (so bench.py's figures line up with rocprofv3's names; every tile
    shape / output type / weight-layout instantiation of the LDS-DMA kernel is reported under one name)."""
    if _PRECISION != "bf16":
        return "gemm_fwd_kernel<float,float>"
    if cin % 64 == 0 and csplit % 64 == 0 and os.environ.get("MADIFF_GEMM_IMPL", "1") != "0":
        return "gemm_fwd_bf16_kernel"
    return "gemm_fwd_kernel<bf16,*>"


class GraphKernelTimer:
    """Per-launch kernel durations INSIDE a replayed hipGraph: while installed (ops.GRAPH_TIMER) during a capture, every
    GEMM launch is bracketed by two timing-enabled event-record nodes on its own stream (added by hand, as the
    data-parallel bucket events are); collect() after each replay adds the elapsed times up.  So the figures are taken
    in the real step: graph-paced launches, the step's own concurrency between branches, clocks under load."""

    def __init__(self):
        self.pairs = []              # (name, flops, start handle, stop handle), one per captured launch
        self.totals = {}
        self.replays = 0

    def bracket(self, name, flops, launch, nbytes=0.0, stream=None):
        import ctypes
        h0, h1 = ctypes.c_void_p(), ctypes.c_void_p()
        st = _stream() if stream is None else stream.cuda_stream
        lib.call("madiff_event_create_timed", ctypes.byref(h0))
        lib.call("madiff_event_create_timed", ctypes.byref(h1))
        lib.call("madiff_event_record_external", h0.value, st)
        used = launch()
        lib.call("madiff_event_record_external", h1.value, st)
        if used != "":
            self.pairs.append((used if isinstance(used, str) else name, flops, nbytes, h0.value, h1.value))

    def collect(self):
        import ctypes
        torch.cuda.synchronize()
        ms = ctypes.c_float()
        for name, flops, nbytes, h0, h1 in self.pairs:
            lib.call("madiff_event_elapsed_ms", h0, h1, ctypes.byref(ms))
            d = self.totals.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["ms"] += ms.value
            d["flops"] += flops
            d["bytes"] += nbytes
        self.replays += 1

    def summary(self):
        return self.totals

    def close(self):
        for _, _, _, h0, h1 in self.pairs:
            lib.call("madiff_event_destroy", h0)
            lib.call("madiff_event_destroy", h1)
        self.pairs = []


GRAPH_TIMER: Optional[GraphKernelTimer] = None


_ABLATE = frozenset(filter(None, os.environ.get("MADIFF_ABLATE", "").split(",")))


def _timed(name: str, flops: float, launch, nbytes: float = 0.0, stream=None):
    """launch() may return the name of the kernel it actually dispatched to (overrides `name`).  `nbytes` = algorithmic
    HBM bytes of the launch (the HBM-bound families); `stream` = the stream the launch goes to when it is not the current one."""
    if _ABLATE and name in _ABLATE:          # diagnostics only (scripts/ablate.sh): what does the step cost without this family?
        return
    if TIMER is None:
        if GRAPH_TIMER is not None and torch.cuda.is_current_stream_capturing():
            GRAPH_TIMER.bracket(name, flops, launch, nbytes, stream)
            return
        launch()
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    used = launch()
    e1.record(stream)
    if used != "":
        TIMER.records.append((used if isinstance(used, str) else name, flops, nbytes, e0, e1))


def graph_capture(graph):
    """torch.cuda.graph(graph) for the train-step and sampler captures, in "thread_local" capture-error mode.

    Why: with an RCCL process group alive, torch's ProcessGroupNCCL watchdog THREAD polls the completion events of recent
    collectives with hipEventQuery (ATen/hip/HIPEvent.h query()).  While any stream captures in the default "global" mode
    this runtime answers hipEventQuery / hipStreamQuery / hipEventSynchronize / hipMalloc from ANY thread with error 900
    (hipErrorStreamCaptureUnsupported) and invalidates the capture (hipStreamEndCapture -> 901); query() passes every error
    but hipErrorNotReady to C10_HIP_CHECK, which throws inside the watchdog thread -> std::terminate -> abort() at whatever
    line the capturing thread happens to be (round 1's "abort at random lines").  In "thread_local" mode the same calls
    from other threads succeed and the capture stays valid (scripts/dbg/capture_foreign_thread.cpp,
    profiles/r02_capture_foreign_thread.txt).  So thread_local IS the fix; nothing is waited for or slept on.
    Kernels launched by the autograd thread onto the capturing stream are captured in either mode."""
    torch.cuda.synchronize()
    return torch.cuda.graph(graph, capture_error_mode="thread_local")


FIRST_TAPS_TILE = 14      # variant ids >= this are the all-taps k5 kernel (csrc/conv_taps_bf16.hip)


def _tile_label(tile):
    return "conv_taps_bf16_kernel" if isinstance(tile, int) and tile >= FIRST_TAPS_TILE else None


# ------------------------------------------------------------------------------------------------
# raw launches
# ------------------------------------------------------------------------------------------------
_TUNE_BY_PRECISION: dict = {_PRECISION: {}}
_TUNE: dict = _TUNE_BY_PRECISION[_PRECISION]
_AUTOTUNE = os.environ.get("MADIFF_AUTOTUNE", "1") != "0"
# MADIFF_TUNE_CACHE=<file>: kernel-variant choices (per build, keyed by layer shape) are read from the file at import and
# written back at exit, so a later process starts without timing launches (and a profiler sees only the steady state).
_TUNE_CACHE = os.environ.get("MADIFF_TUNE_CACHE")


def load_tune_cache(path: str) -> int:
    """Merge remembered variant choices from `path` (a file save_tune_cache wrote).  Returns the number of entries."""
    import ast
    import json
    global _TUNE
    with open(path) as f:
        data = json.load(f)
    if data.get("arch") != lib.arch() or data.get("version") != lib.version():
        return 0                       # choices of another library build do not carry over
    n = 0
    for prec, entries in data.get("choices", {}).items():
        d = _TUNE_BY_PRECISION.setdefault(prec, {})
        for k, v in entries:
            d[ast.literal_eval(k)] = tuple(v) if isinstance(v, list) else v
            n += 1
    _TUNE = _TUNE_BY_PRECISION.setdefault(_PRECISION, {})
    return n


def save_tune_cache(path: str) -> None:
    import json
    data = {"arch": lib.arch(), "version": lib.version(),
            "choices": {prec: [[repr(k), v] for k, v in d.items()] for prec, d in _TUNE_BY_PRECISION.items() if d}}
    tmp = f"{path}.{os.getpid()}.tmp"
    with open(tmp, "w") as f:
        json.dump(data, f)
    os.replace(tmp, path)


# Packaged defaults (madiff_amd/tune_gfx950.json, written by scripts/make_tune_defaults.sh from the best of several
# freshly tuned bench runs): read when no MADIFF_TUNE_CACHE is named, never written; shapes they do not hold are tuned on
# first use as usual.  MADIFF_TUNE_DEFAULTS=0 ignores them.
_TUNE_DEFAULTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tune_gfx950.json")
if not _TUNE_CACHE and os.environ.get("MADIFF_TUNE_DEFAULTS", "1") != "0" and os.path.exists(_TUNE_DEFAULTS):
    try:
        load_tune_cache(_TUNE_DEFAULTS)
    except (OSError, ValueError, SyntaxError):
        pass

if _TUNE_CACHE:
    import atexit
    if os.path.exists(_TUNE_CACHE):
        try:
            load_tune_cache(_TUNE_CACHE)
        except (OSError, ValueError, SyntaxError):
            pass                       # unreadable cache: tune again
    atexit.register(lambda: save_tune_cache(_TUNE_CACHE))


_TUNE_COLD = os.environ.get("MADIFF_TUNE_COLD", "1") != "0"
_THRASH: dict = {}


def _thrash_l2() -> None:
    """Between two timed launches of a tuning decision: overwrite 64 MB, twice the chip's L2.  In the real step a layer's
    weights arrive cold (every layer of the network runs between two uses); five back-to-back launches of ONE layer keep
    its weight tiles in L2 and flattered the larger tiles: the tuner chose the 192-row tile for the T = 12 layers of the
    sampler (208 vs 215 us alone) where the 96-row one is 189 vs 232 us inside the loop.  The fill costs every candidate
    the same."""
    if not _TUNE_COLD:
        return
    dev = torch.cuda.current_device()
    buf = _THRASH.get(dev)
    if buf is None:
        buf = _THRASH[dev] = torch.empty(64 << 20, dtype=torch.uint8, device=f"cuda:{dev}")
    buf.zero_()


def _time_calls(fn, reps: int = 5) -> float:
    """Milliseconds for `reps` calls of fn on the current stream (tuning decisions, never under capture)."""
    fn()
    best = float("inf")
    for _ in range(2):                   # best of two rounds (see _tuned_tile)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            _thrash_l2()
            fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def _tuned_tile(sig, candidates, launch) -> int:
    """Output tile of the LDS-DMA GEMM for one layer shape.  The first time a shape is seen (outside graph capture)
    every tile the library offers is timed on the real operands and the fastest is remembered; results do not depend
    on the tile (same reduction order per output element), so this is a pure speed choice.  -1 = library's model."""
    t = _TUNE.get(sig)
    if t is not None:
        return t
    if not _AUTOTUNE or _PRECISION != "bf16" or torch.cuda.is_current_stream_capturing():
        return -1
    best, best_ms = -1, float("inf")
    for tile in candidates:
        if launch(tile):                 # non-zero: the variant does not cover this layer (e.g. the all-taps k5 kernel)
            continue
        ms = float("inf")
        for _ in range(3):               # best of three rounds: one round's figure moves by several % with what ran before it
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                _thrash_l2()
                launch(tile)
            e1.record()
            e1.synchronize()
            ms = min(ms, e0.elapsed_time(e1))
        if ms < best_ms:
            best, best_ms = tile, ms
    _TUNE[sig] = best
    return best


def _gemm(x, x2, csplit, M, cin, spec: ConvSpec, t_out, t_in, tr, W, nout, npad, bias, rowbias, out, act=0,
          perm: Optional[Tuple[int, int]] = None):
    out_f32 = 1 if (out.dtype == torch.float32 and _PRECISION == "bf16") else 0
    dma = _PRECISION == "bf16" and cin % 64 == 0 and csplit % 64 == 0

    def launch(tile):
        return lib.call_rc("madiff_gemm_fwd", _dt(), out_f32, x.data_ptr(), x.stride(0), _p(x2),
                           0 if x2 is None else x2.stride(0), csplit, M, cin, spec.taps, t_out, t_in, spec.stride, spec.pad,
                           tr, W.data_ptr(), W.stride(0), nout, npad, _p(bias), _p(rowbias),
                           0 if rowbias is None else rowbias.stride(0), out.data_ptr(), out.stride(0), act,
                           perm[0] if perm else 0, perm[1] if perm else 0, tile, _stream())

    tile = _tuned_tile(("fwd", M, cin, spec.taps, npad, t_out, spec.stride, tr, out_f32, x2 is not None,
                        rowbias is not None, x.stride(0), out.stride(0)),
                       lib.tiles(), launch) if dma else -1
    if launch(tile):                     # (a tuned variant may not cover a same-shaped layer with other alignments)
        tile = -1
        if launch(-1):
            raise RuntimeError("madiff_gemm_fwd: no kernel variant covers this layer")
    return _tile_label(tile)


def _dgrad_wt(dy, M_in, npad, spec: ConvSpec, t_in, t_out, Wf, wtap, wcol0, ncols, ncols_pad, addend=None):
    """Data gradient straight from the forward-layout bf16 weights (no transposed pack).  Returns the cl gradient
    [M_in, ncols_pad], or None when the shape is not covered.  addend (cl, same shape): merged in the kernel's epilogue
    when the kernel offers it (_dgrad_wt.fused_add tells), else left to the caller."""
    # few output tiles + a long reduction (the time-MLP bundle: 1024 x 256 from K = 11520): slice the reduction
    blocks = -(-M_in // 64) * -(-ncols_pad // 128)
    nsteps = (npad // 64) * spec.taps
    ksplit = 0
    if blocks < 96 and nsteps >= 32:
        ksplit = max(1, min(nsteps // 8, 256 // blocks))
    gx = torch.empty(M_in, ncols_pad, dtype=torch.float32 if ksplit > 1 else act_dtype(), device=dy.device)
    _dgrad_wt.fused_add = False
    if addend is not None and (ksplit > 1 or addend.shape != (M_in, ncols_pad) or addend.dtype != act_dtype()
                               or addend.stride(1) != 1):
        addend = None

    def launch(tile, add=addend):
        return lib.call_rc("madiff_gemm_dgrad_wt_add", dy.data_ptr(), dy.stride(0), M_in, npad, spec.taps, t_in, t_out,
                           spec.stride, spec.pad, 1 - spec.tr, Wf.data_ptr(), Wf.stride(0), wtap, wcol0, ncols,
                           ncols_pad, gx.data_ptr(), gx.stride(0), tile, ksplit, _p(add), 0 if add is None else add.stride(0),
                           _stream())

    sig = ("wt", M_in, npad, spec.taps, ncols_pad, t_in, spec.stride, spec.tr)
    tile = _TUNE.get(sig)
    if tile == "uncovered":
        return None
    if ksplit > 1 or tile is None:
        if launch(-1) != 0:          # probe = the real launch with the library's default variant
            if addend is None or launch(-1, None) != 0:
                _TUNE[sig] = "uncovered"
                return None
            addend = None
        if ksplit <= 1:              # remember the fastest variant (no-op under capture / if disabled)
            _tuned_tile(sig, lib.tiles(wt=True), lambda t: launch(t, None))
            if _TUNE.get(sig) is not None and addend is not None and launch(-1) != 0:
                raise RuntimeError("madiff_gemm_dgrad_wt_add: the probed variant stopped covering this layer")
    elif launch(tile):
        if addend is not None and launch(tile, None) == 0:
            addend = None            # this variant has no vector epilogue here: the caller adds
        else:
            tile = -1
            if launch(-1):
                if addend is None or launch(-1, None):
                    raise RuntimeError("madiff_gemm_dgrad_wt: no kernel variant covers this layer")
                addend = None
    _dgrad_wt.fused_add = addend is not None
    _dgrad_wt.last_label = _tile_label(tile)
    return gx.to(act_dtype()) if ksplit > 1 else gx


# variants of madiff_gemm_wgrad: (O tile, C tile, all taps in one pass?); 4..7 = {64x64, 128x64} x {2, 3} LDS stages;
# 8, 9 = the 8-wave all-taps form (reduction rows of a K-step split over two wave groups, combined in LDS)
_WGRAD_TILES = ((128, 128, 0), (128, 64, 0), (64, 128, 0), (64, 64, 0), (64, 64, 1), (64, 64, 1), (128, 64, 1), (128, 64, 1),
                (64, 64, 1), (64, 64, 1))


def _wgrad(dy, x, x2, csplit, M, cin, spec: ConvSpec, t_out, t_in, dW, nout, accumulate=0, dbias=None, btab=None):
    """dbias (fp32 [nout], accumulated) / btab (device table of one destination address per output column): also produce
    the bias gradient (madiff_gemm_wgrad_bias); the tuning launches never do."""
    def launch(dst, acc, tile, splits, bias=False):
        if bias and (dbias is not None or btab is not None):
            return lib.call_rc("madiff_gemm_wgrad_bias", _dt(), dy.data_ptr(), dy.stride(0), dy.shape[1], x.data_ptr(),
                               x.stride(0), _p(x2), 0 if x2 is None else x2.stride(0), csplit, M, cin, spec.taps, t_out,
                               t_in, spec.stride, spec.pad, spec.tr, dst.data_ptr(), dst.stride(0), nout, acc, tile, splits,
                               _p(dbias), _p(btab), _stream())
        return lib.call_rc("madiff_gemm_wgrad", _dt(), dy.data_ptr(), dy.stride(0), dy.shape[1], x.data_ptr(),
                           x.stride(0), _p(x2), 0 if x2 is None else x2.stride(0), csplit, M, cin, spec.taps, t_out, t_in,
                           spec.stride, spec.pad, spec.tr, dst.data_ptr(), dst.stride(0), nout, acc, tile, splits,
                           _stream())

    sig = ("wgrad", M, cin, spec.taps, nout, t_out, spec.stride, spec.tr, x2 is not None, accumulate)
    cfg = _TUNE.get(sig)
    if cfg is None and _AUTOTUNE and _PRECISION == "bf16" and not torch.cuda.is_current_stream_capturing():
        # time every (variant, reduction split) candidate on the real operands into a scratch matrix
        scratch = torch.empty(nout, dW.shape[1], dtype=torch.float32, device=dW.device)
        best, best_ms = (-1, 0), float("inf")
        for tile, (bo, bc, all_taps) in enumerate(_WGRAD_TILES):
            steps = (M + 63) // 64
            tiles = -(-nout // bo) * -(-cin // bc) * (1 if all_taps else spec.taps)
            for splits in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
                if splits > 1 and (splits > steps // 4 or tiles * splits > 2048):
                    continue
                if launch(scratch, 1, tile, splits) != 0:
                    break                      # variant not applicable to this layer
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    _thrash_l2()
                    launch(scratch, 1, tile, splits)
                e1.record()
                e1.synchronize()
                ms = e0.elapsed_time(e1)
                if ms < best_ms:
                    best, best_ms = (tile, splits), ms
        cfg = _TUNE[sig] = best
    if cfg is None:
        cfg = (-1, 0)
    if launch(dW, accumulate, cfg[0], cfg[1], True) != 0:
        # (a remembered variant that does not cover this layer after all: the library's own choice must)
        if cfg == (-1, 0) or launch(dW, accumulate, -1, 0, True) != 0:
            raise RuntimeError("madiff_gemm_wgrad: no kernel variant covers this layer")
        cfg = _TUNE[sig] = (-1, 0)
    return "gemm_wgrad_taps_bf16_kernel" if (cfg[0] >= 4 and _PRECISION == "bf16") else None


def pack_traj(x: torch.Tensor, fpad: int, noise=None, t=None, sqrt_ac=None, sqrt_1mac=None, cond=None, mask=None,
              x_noisy=None) -> torch.Tensor:
    """[B,T,A,F] fp32 -> cl [(b a) t, fpad] (optionally fused add_noise + inpainting).  x may be a column slice of a wider
    contiguous [B,T,A,W] tensor (the batch's observation columns, diffusion.py:463): it is read in place."""
    B, T, A, F = x.shape
    assert x.dtype == torch.float32
    W = x.stride(2)
    if not (x.stride(3) == 1 and W >= F and x.stride(1) == A * W and x.stride(0) == T * A * W):
        x = x.contiguous()
        W = F
    out = torch.empty(B * A * T, fpad, dtype=act_dtype(), device=x.device)
    lib.call("madiff_pack_input", _dt(), x.data_ptr(), _p(noise), _p(t), _p(sqrt_ac), _p(sqrt_1mac), _p(cond), _p(mask),
             out.data_ptr(), _p(x_noisy), B, T, A, F, fpad, W, _stream())
    return out


def unpack_traj(xcl: torch.Tensor, B, T, A, F) -> torch.Tensor:
    out = torch.empty(B, T, A, F, dtype=torch.float32, device=xcl.device)
    lib.call("madiff_unpack_output", _dt(), xcl.data_ptr(), xcl.stride(0), out.data_ptr(), B, T, A, F, _stream())
    return out


# ------------------------------------------------------------------------------------------------
# autograd Functions
# ------------------------------------------------------------------------------------------------
# ------------------------------------------------------------------------------------------------
# gradient sinks: let backward kernels accumulate straight into the optimizer's flat gradient buffer
# ------------------------------------------------------------------------------------------------
class GradSink:
    """Maps a parameter to a live fp32 gradient view (same shape).  When installed (ops.SINK), the backward kernels
    ACCUMULATE parameter gradients into these views and return None to autograd, which removes one temporary, one
    zero-fill and one `grad += new` pass per parameter.  `notify(param)` is called after the write (the
    data-parallel optimizer counts bucket readiness there).

    GEMM-layout slots (the flat optimizer keeps every weight MATRIX as [O, taps, pad32(I)] inside its flat buffers):
    `register_matrix` records the slot, after which
      * weight_matrix(group) is the GEMM's forward weight operand itself -- the fp32 flat parameter buffer in the
        fp32 build, its bf16 shadow (maintained by madiff_adam_ema) in the bf16 build: no pack pass;
      * grad_matrix(group) is the [sum O, taps*pad32(I)] slice of the flat gradient the weight-gradient GEMM
        accumulates into directly: no unpack pass.
    A group (layers sharing their input, e.g. q/k/v) is served when its members sit back to back in the flat buffer.
    """

    def __init__(self):
        self.views = {}
        self.notify = None
        self.deferred = None
        self.flat_p = self.flat_g = self.flat_p16 = None
        self.slots = {}          # id(param) -> (weakref, offset, O, K, kind, taps)
        self.shadow_state = {}   # id(param) -> (param._version, _EPOCH_EXT) at which the bf16 shadow was in sync
        self._groups = {}
        self._bias_tables = {}

    def register(self, param, view):
        assert view.dtype == torch.float32 and view.shape == param.shape
        self.views[id(param)] = (weakref.ref(param), view)

    def register_matrix(self, param, offset: int, O: int, K: int, kind: str, taps: int):
        self.slots[id(param)] = (weakref.ref(param), offset, O, K, kind, taps)

    def get(self, param):
        ent = self.views.get(id(param))
        if ent is None or ent[0]() is not param:
            return None
        return ent[1]

    def done(self, param):
        if self.deferred is not None:        # issued on a side stream: announce once the issuing code is back on its stream
            self.deferred.append(param)
        elif self.notify is not None:
            self.notify(param)

    def _group(self, weights, spec):
        """(offset, sum O, K) when `weights` occupy consecutive GEMM-layout slots matching `spec`, else None."""
        key = (tuple(id(w) for w in weights), spec.kind, spec.taps)
        g = self._groups.get(key)
        if g is not None and all(self.slots[id(w)][0]() is w for w in weights):
            return g
        off0 = nxt = K0 = None
        osum = 0
        for w in weights:
            sl = self.slots.get(id(w))
            if sl is None or sl[0]() is not w or sl[4] != spec.kind or sl[5] != spec.taps:
                return None
            _, off, O, K, _, _ = sl
            if off0 is None:
                off0, K0 = off, K
            elif off != nxt or K != K0:
                return None
            nxt = off + O * K
            osum += O
        g = (off0, osum, K0)
        self._groups[key] = g
        return g

    def weight_matrix(self, weights, spec):
        if self.flat_p is None:
            return None
        g = self._group(weights, spec)
        if g is None:
            return None
        off, osum, K = g
        if _PRECISION == "fp32":
            return self.flat_p[off:off + osum * K].view(osum, K)      # the parameter memory itself
        if self.flat_p16 is None:
            return None
        cur_ext = _EPOCH_EXT
        for w in weights:
            if self.shadow_state.get(id(w)) != (w._version, cur_ext):
                self._refresh_shadow(w)
        return self.flat_p16[off:off + osum * K].view(osum, K)

    def _refresh_shadow(self, w):
        """The parameter was written by something other than the fused optimizer (load_state_dict, broadcast, ...)."""
        _, off, O, K, kind, taps = self.slots[id(w)]
        O_, I, so, si, st = _strides(kind, w)
        lib.call("madiff_pack_weight", lib.BF16, w.data_ptr(), self.flat_p16.data_ptr() + 2 * off, O, I, taps, so, si, st,
                 K // taps, 0, _stream())
        self.shadow_state[id(w)] = (w._version, _EPOCH_EXT)

    def bias_table(self, biases, device):
        """Device array of one gradient-element address per output column of a bias group (built on first use, outside
        graph capture: it needs a host-to-device copy); None when it cannot be built now."""
        key = tuple(id(b) for b in biases)
        ent = self._bias_tables.get(key)
        if ent is not None and all(r() is b for r, b in zip(ent[0], biases)):
            return ent[1]
        if torch.cuda.is_current_stream_capturing():
            return None
        addrs = []
        for b in biases:
            v = self.get(b)
            assert v.is_contiguous()
            addrs += [v.data_ptr() + 4 * i for i in range(v.numel())]
        tab = torch.tensor(addrs, dtype=torch.int64).to(device)
        self._bias_tables[key] = (tuple(weakref.ref(b) for b in biases), tab)
        return tab

    def grad_matrix(self, weights, spec):
        if self.flat_g is None:
            return None
        g = self._group(weights, spec)
        if g is None:
            return None
        off, osum, K = g
        return self.flat_g[off:off + osum * K].view(osum, K)


def known_gemm_weights():
    """{id(param): (kind, taps, group ids)} for every parameter the GEMM layers have used so far (from the pack cache);
    the flat optimizer lays its buffers out from this after the first forward/backward."""
    out = {}
    for (ids, kind, taps, backward, _prec), ent in _PACK_CACHE.items():
        ws = [r() for r in ent.refs]
        if any(w is None for w in ws):
            continue
        for w in ws:
            prev = out.get(id(w))
            if prev is None or len(ids) > len(prev[2]):
                out[id(w)] = (kind, taps, ids)
    return out


SINK: Optional[GradSink] = None


def _sink_view(param):
    return None if SINK is None else SINK.get(param)


class _Geom:
    pass


def _conv_forward(x, x2, rowbias, spec: ConvSpec, t_in: int, act: int, perm, weights, biases):
    _chk_cl(x)
    dims = [_strides(spec.kind, w) for w in weights]
    I = dims[0][1]
    osum = sum(d[0] for d in dims)
    c1 = x.shape[1]
    cin = c1 + (x2.shape[1] if x2 is not None else 0)
    assert cin == pad32(I) or (x2 is not None and cin == I), (cin, I)
    M_in = x.shape[0]
    n_seq = M_in // t_in
    t_out = spec.out_len(t_in)
    M = n_seq * t_out
    W = _packed(weights, spec, False)
    rv = rowvec_of(rowbias)
    if rv is not None:        # per-step vector beside the row bias: it takes the bias operand (whoever set it folded the
        bias = rv             # layer's own biases into the row-bias matrix, see SharedConvAttentionDeconv.proj_static)
        assert rv.dtype == torch.float32 and rv.is_contiguous() and rv.numel() >= osum
    elif len(biases) == 0:
        bias = None
    elif len(biases) == 1:
        bias = biases[0]
    else:
        bias = torch.cat([b.detach() for b in biases])
    out_f32 = perm == "f32"
    if out_f32:
        perm = None
        npad = pad32(osum)
        out = torch.empty(M, npad, dtype=torch.float32, device=x.device)
    elif perm is not None:
        out = torch.empty(M, osum, dtype=torch.float32, device=x.device)
        npad = osum
    else:
        npad = pad32(osum)
        out = torch.empty(M, npad, dtype=act_dtype(), device=x.device)
    if x2 is not None:
        _chk_cl(x2)
    g = _Geom()
    # algorithmic FLOPs of this layer = 2 * dense MACs (the count SURVEY.md section 8d uses); the data- and
    # weight-gradient launches of the same layer do the same number of MACs
    g.flops = 2.0 * (M_in if spec.tr else M) * osum * I * spec.taps
    _timed(_gemm_label(cin, c1, out.dtype == torch.float32), g.flops,
           lambda: _gemm(x, x2, c1, M, cin, spec, t_out, t_in, spec.tr, W, osum, npad, bias, rowbias, out, act, perm))
    # (_gemm returns the dispatched kernel's name when it is not the default LDS-DMA kernel)
    g.spec, g.t_in, g.t_out, g.act, g.perm = spec, t_in, t_out, act, perm
    g.dims, g.osum, g.cin, g.c1, g.n_seq, g.I = dims, osum, cin, c1, n_seq, I
    return out, g


def _conv_backward(g, x, x2, weights, biases, dy, need_dx, need_dx2, need_w, bias_mode, need_rb, addends=(None, None)):
    """dy: cl [M_out, pad32(osum)] in the storage dtype.  bias_mode: 'colsum' | 'none' (bias gradient produced
    elsewhere, e.g. by the fused GroupNorm backward).  addends = gradients of (x, x2) from another consumer, added to
    dx / dx2 (in the data-gradient kernel's epilogue where it offers that).  Returns (dx, dx2, drowbias, [gw], [gb])."""
    spec, t_in, t_out, osum, cin, c1, n_seq, I = g.spec, g.t_in, g.t_out, g.osum, g.cin, g.c1, g.n_seq, g.I
    M, M_in, npad = n_seq * t_out, n_seq * t_in, pad32(osum)
    dx = dx2 = drb = None
    if need_dx or (x2 is not None and need_dx2):
        # bf16 build: the data gradient reads the FORWARD weight matrix through transposing LDS reads
        Wf = _packed(weights, spec, False) if (_PRECISION == "bf16" and npad % 64 == 0 and osum == npad) else None
        Wb = None
        parts = [(0, c1 if x2 is not None else I, need_dx)]
        if x2 is not None:
            parts.append((c1, I - c1, need_dx2))
        grads = []
        for (row0, rows, need), addend in zip(parts, addends):
            if not need:
                grads.append(None)
                continue
            res = [None]
            fused_add = False

            def run_wt():
                _dgrad_wt.last_label = None
                res[0] = _dgrad_wt(dy, M_in, npad, spec, t_in, t_out, Wf, cin, row0, rows, pad32(rows), addend)
                return _dgrad_wt.last_label

            if Wf is not None:
                _timed("gemm_fwd_bf16_kernel", g.flops * rows / I, run_wt)
                fused_add = res[0] is not None and _dgrad_wt.fused_add
            gx = res[0]
            if gx is None:
                gx = torch.empty(M_in, pad32(rows), dtype=act_dtype(), device=dy.device)
                if Wb is None:
                    Wb = _packed(weights, spec, True)          # [I, taps * npad]
                es = Wb.element_size()
                _timed(_gemm_label(npad, npad, False), g.flops * rows / I, lambda: lib.call(
                    "madiff_gemm_fwd", _dt(), 0, dy.data_ptr(), dy.stride(0), None, 0, npad, M_in, npad, spec.taps,
                    t_in, t_out, spec.stride, spec.pad, 1 - spec.tr, Wb.data_ptr() + row0 * Wb.stride(0) * es,
                    Wb.stride(0), rows, pad32(rows), None, None, 0, gx.data_ptr(), gx.stride(0), 0, 0, 0, -1, _stream()))
            if addend is not None and not fused_add:
                gx = gx + addend
            grads.append(gx)
        dx = grads[0]
        if x2 is not None:
            dx2 = grads[1]
    gw: List[Optional[torch.Tensor]] = [None] * len(weights)
    gw, gb = _param_grads_side(g, x, x2, weights, biases, dy, need_w, bias_mode, gw)
    if need_rb:
        drb = torch.empty(n_seq, osum, dtype=torch.float32, device=dy.device)
        lib.call("madiff_seqsum", _dt(), dy.data_ptr(), dy.stride(0), drb.data_ptr(), drb.stride(0), n_seq, t_out,
                 osum, _stream())
    return dx, dx2, drb, gw, gb


WGRAD_SIDE_STREAM = os.environ.get("MADIFF_WGRAD_SIDE_STREAM", "1") != "0"
FUSE_BIAS_WGRAD = os.environ.get("MADIFF_FUSE_BIAS_WGRAD", "1") != "0"
# parameter-gradient launches go round-robin over this many auxiliary streams (parallel branches of the captured step)
WGRAD_STREAMS = int(os.environ.get("MADIFF_WGRAD_STREAMS", "4"))


def wgrad_branch_name(sink, weights) -> str:
    """The auxiliary stream a layer's parameter-gradient launch goes to.  The branch belongs to the PARAMETER (assigned in
    turn the first time the layer's weight is seen, then remembered in the sink), not to the launch: the launch is a
    non-atomic read-modify-write of the flat gradient, so two launches for one parameter -- the micro-batches of an
    unfolded accumulation step -- must be ordered, which the same stream (and, in a captured step, the chain of nodes it
    becomes) guarantees.  A global launch counter put them on different branches with no edge between them."""
    if WGRAD_STREAMS <= 1:
        return "wgrad"
    table = sink.__dict__.setdefault("_wgrad_branch", {})
    key = id(weights[0])
    ent = table.get(key)
    if ent is None or ent[0]() is not weights[0]:
        ent = table[key] = (weakref.ref(weights[0]), len(table) % WGRAD_STREAMS)
    return "wgrad%d" % ent[1]


def _param_grads_side(g, x, x2, weights, biases, dy, need_w, bias_mode, gw):
    """Parameter gradients of a layer are needed by the optimizer only, the data gradient by the rest of backward: when
    every result lands in the flat gradient (gradient sink installed), the weight-gradient GEMM and the bias column
    sums are issued on the "wgrad" auxiliary stream, i.e. as a parallel branch of the (captured) step, so that they fill
    the compute units the small layers' data-gradient launches leave idle.  Bucket notifications are delivered after
    returning to the issuing stream (the optimizer's hook then orders the bucket behind every auxiliary stream)."""
    sink = SINK
    if not (WGRAD_SIDE_STREAM and dy.is_cuda and sink is not None and sink.deferred is None and TIMER is None
            and any(need_w) and all(need_w) and sink.grad_matrix(weights, g.spec) is not None
            and (bias_mode != "colsum" or all(_sink_view(b) is not None for b in biases))):
        return _param_grads(g, x, x2, weights, biases, dy, need_w, bias_mode, gw)
    cur = torch.cuda.current_stream()
    side = aux_stream(dy.device, wgrad_branch_name(sink, weights))
    wait_stream(side, cur)
    sink.deferred = []
    try:
        with torch.cuda.stream(side):
            out = _param_grads(g, x, x2, weights, biases, dy, need_w, bias_mode, gw)
    finally:
        pending, sink.deferred = sink.deferred, None
    for t in (dy, x, x2):
        if t is not None:
            t.record_stream(side)
    for p_ in pending:
        sink.done(p_)
    return out


def _param_grads(g, x, x2, weights, biases, dy, need_w, bias_mode, gw):
    spec, t_in, t_out, osum, cin, c1, n_seq, I = g.spec, g.t_in, g.t_out, g.osum, g.cin, g.c1, g.n_seq, g.I
    M = n_seq * t_out
    bias_done = False
    if any(need_w):
        wlabel = "gemm_wgrad_bf16_kernel" if _PRECISION == "bf16" else "gemm_wgrad_kernel<float>"
        gm = SINK.grad_matrix(weights, spec) if (SINK is not None and all(need_w)) else None
        if gm is not None:       # accumulate straight into the flat gradient (kept in this GEMM layout)
            assert gm.shape == (osum, spec.taps * cin)
            # the bias gradient rides along (column sums of dy inside the weight-gradient kernel) when every bias of the
            # group has a live slot in the flat gradient
            dbias = btab = None
            if FUSE_BIAS_WGRAD and biases and bias_mode == "colsum" and all(_sink_view(b) is not None for b in biases):
                if len(biases) == 1:
                    dbias = _sink_view(biases[0])
                else:
                    btab = SINK.bias_table(biases, dy.device)
            _timed(wlabel, g.flops, lambda: _wgrad(dy, x, x2, c1, M, cin, spec, t_out, t_in, gm, osum, 1, dbias, btab))
            for w in weights:
                SINK.done(w)
            if dbias is not None or btab is not None:
                bias_done = True
                for b in biases:
                    SINK.done(b)
        else:
            dW = torch.empty(osum, spec.taps * cin, dtype=torch.float32, device=dy.device)
            _timed(wlabel, g.flops, lambda: _wgrad(dy, x, x2, c1, M, cin, spec, t_out, t_in, dW, osum, 0))
            off = 0
            for i, (w, (O, I_, so, si, st)) in enumerate(zip(weights, g.dims)):
                if need_w[i]:
                    gv = dW[off:off + O].view(O, spec.taps, cin)[:, :, :I_]     # GEMM layout -> the parameter's shape
                    gv = gv.permute(0, 2, 1) if spec.kind == "conv" else (gv.permute(2, 0, 1) if spec.kind == "convT"
                                                                          else gv[:, 0, :])
                    view = _sink_view(w)
                    if view is not None:
                        view.add_(gv)
                        SINK.done(w)
                    else:
                        gw[i] = gv
                off += O
    gb: List[Optional[torch.Tensor]] = [None] * len(biases)
    if biases and bias_mode == "colsum" and not bias_done:
        if len(biases) == 1 and _sink_view(biases[0]) is not None:
            view = _sink_view(biases[0])       # colsum accumulates atomically: add straight into the live gradient
            lib.call("madiff_colsum", _dt(), dy.data_ptr(), dy.stride(0), view.data_ptr(), M, osum, _stream())
            SINK.done(biases[0])
        elif all(_sink_view(b) is not None for b in biases):
            # a group (q/k/v, the time-MLP bundle): one column-slice reduction per member, each straight into its live
            # gradient (no temporary, no `grad += new` pass per member)
            tab = SINK.bias_table(biases, dy.device)
            if tab is None or lib.call_rc("madiff_colsum_scatter", _dt(), dy.data_ptr(), dy.stride(0), tab.data_ptr(), M,
                                          osum, _stream()):
                off, es = 0, dy.element_size()
                for b, (O, *_r) in zip(biases, g.dims):
                    lib.call("madiff_colsum", _dt(), dy.data_ptr() + off * es, dy.stride(0), _sink_view(b).data_ptr(), M,
                             O, _stream())
                    off += O
            for b in biases:
                SINK.done(b)
        else:
            db = torch.zeros(osum, dtype=torch.float32, device=dy.device)
            lib.call("madiff_colsum", _dt(), dy.data_ptr(), dy.stride(0), db.data_ptr(), M, osum, _stream())
            off = 0
            for i, (O, *_r) in enumerate(g.dims):
                gb[i] = db[off:off + O]
                off += O
    return gw, gb


class ConvGemm(torch.autograd.Function):
    """y = conv/linear(x [, x2]) + bias (+ rowbias[n]) for a GROUP of layers sharing the input.

    apply(x, x2, rowbias, spec, t_in, act, perm, n_groups, *weights, *biases) -> cl [M_out, pad32(sum O)]
    perm = (T, A): fp32 [B*T*A, O] in the reference row order; perm = "f32": fp32 [M_out, pad32(sum O)].
    """

    @staticmethod
    def forward(ctx, x, x2, rowbias, spec: ConvSpec, t_in: int, act: int, perm, n_groups: int, *params):
        weights, biases = params[:n_groups], params[n_groups:]
        out, g = _conv_forward(x, x2, rowbias, spec, t_in, act, perm, weights, biases)
        ctx.save_for_backward(x, x2, out if act == 1 else None, *params)
        ctx.g, ctx.n_groups, ctx.has_rowbias = g, n_groups, rowbias is not None
        return out

    @staticmethod
    def backward(ctx, dy):
        x, x2, y_act = ctx.saved_tensors[:3]
        params = ctx.saved_tensors[3:]
        g, ng = ctx.g, ctx.n_groups
        weights, biases = params[:ng], params[ng:]
        npad = pad32(g.osum)
        if g.perm is not None:       # dy arrives as fp32 [B*T*A, O] in the reference row order
            Tn, A = g.perm
            dy = pack_traj(dy.reshape(g.n_seq // A, Tn, A, g.osum), npad)
        else:
            dy = dy.contiguous()
            if dy.dtype != act_dtype():
                dy = dy.to(act_dtype())
        if g.act == 1:
            d2 = torch.empty_like(dy)
            lib.call("madiff_act_bwd", _dt(), 1, y_act.data_ptr(), dy.data_ptr(), d2.data_ptr(), dy.numel(), _stream())
            dy = d2
        needs = ctx.needs_input_grad
        dx, dx2, drb, gw, gb = _conv_backward(g, x, x2, weights, biases, dy, needs[0], needs[1],
                                              [needs[8 + i] for i in range(ng)], "colsum",
                                              ctx.has_rowbias and needs[2])
        return (dx, dx2, drb, None, None, None, None, None, *gw, *gb)


def conv_gemm(x, weights, biases, spec: ConvSpec, t_in: int, x2=None, rowbias=None, act=0, perm=None):
    weights = list(weights)
    biases = [] if biases is None else list(biases)
    if x2 is not None and x.shape[1] % 128 != 0:   # the split-source wgrad tiles need 128-aligned halves
        x, x2 = torch.cat((x, x2), dim=1), None
    return ConvGemm.apply(x, x2, rowbias, spec, t_in, act, perm, len(weights), *weights, *biases)


def _gn_forward(y, gamma, beta, rowbias, res, n_seq, t, C, G):
    _chk_cl(y)
    rv = rowvec_of(rowbias)
    if rv is not None:                         # the standalone kernels take one row bias: form it (rare: unfused layers)
        rowbias = rowbias + rv[: rowbias.shape[1]]
    cpad = y.shape[1]
    out = torch.empty_like(y)
    stats = torch.empty(2, n_seq * G, dtype=torch.float32, device=y.device)
    mode = 1 if rowbias is not None else (2 if res is not None else 0)
    if rowbias is not None:
        assert rowbias.dtype == torch.float32 and rowbias.stride(1) == 1
    if res is not None:
        _chk_cl(res)
    es = y.element_size()                      # algorithmic bytes: read y (+ residual), write out
    _timed("gn_mish_fwd_rb_kernel", 0.0, lambda: lib.call(
        "madiff_gn_mish_fwd", _dt(), y.data_ptr(), y.stride(0), out.data_ptr(), out.stride(0), gamma.data_ptr(),
        beta.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), n_seq, t, C, cpad, G, 1e-5, mode, _p(rowbias),
        0 if rowbias is None else rowbias.stride(0), _p(res), 0 if res is None else res.stride(0), _stream()),
        nbytes=float(n_seq * t * C * es * (3 if mode == 2 else 2)))
    return out, stats, (n_seq, t, C, cpad, G, mode)


def _gn_param_bufs(gamma, beta, conv_bias, C, device):
    """Destinations of dgamma / dbeta / dconv_bias: the live slots of the flat gradient when a sink is installed, else a
    zeroed temporary.  -> (plist, views, bufs)"""
    plist = [gamma, beta] + ([conv_bias] if conv_bias is not None else [])
    views = [_sink_view(p_) for p_ in plist]
    tmp = None
    if any(v is None for v in views):
        tmp = torch.zeros(len(plist), C, dtype=torch.float32, device=device)
    bufs = [v if v is not None else tmp[i] for i, v in enumerate(views)]
    return plist, views, bufs


def _gn_param_outs(plist, views, bufs, conv_bias):
    outs = []
    for p_, v, b in zip(plist, views, bufs):
        if v is not None:
            SINK.done(p_)
            outs.append(None)
        else:
            outs.append(b)
    if conv_bias is None:
        outs.append(None)
    return outs


def _gn_backward(y, gamma, beta, stats, geom, dout, conv_bias=None, drb_slot=None):
    """-> (dy, dgamma, dbeta, dconv_bias, drowbias); parameter gradients go to the sink when one is installed.
    drb_slot: grad_slot() of the row-bias tensor: its gradient is then written straight into the shared buffer."""
    n_seq, t, C, cpad, G, mode = geom
    dout = dout.contiguous()
    dy = torch.empty_like(y)
    plist, views, bufs = _gn_param_bufs(gamma, beta, conv_bias, C, y.device)
    drb = None
    if mode == 1:
        drb = slot_view(drb_slot, n_seq, C)
        if drb is None:
            drb = torch.empty(n_seq, C, dtype=torch.float32, device=y.device)
    _timed("gn_mish_bwd_rb_kernel", 0.0, lambda: lib.call(           # algorithmic bytes: read y and dout, write dy
        "madiff_gn_mish_bwd", _dt(), y.data_ptr(), y.stride(0), dout.data_ptr(), dout.stride(0), dy.data_ptr(),
        dy.stride(0), gamma.data_ptr(), beta.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(),
        bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr() if conv_bias is not None else None, _p(drb),
        C if drb is None else drb.stride(0), n_seq, t, C, cpad, G, _stream()),
        nbytes=float(3 * n_seq * t * C * y.element_size()))
    outs = _gn_param_outs(plist, views, bufs, conv_bias)
    return dy, outs[0], outs[1], outs[2], drb, dout


FUSE_DGRAD_GN_BWD = os.environ.get("MADIFF_FUSE_DGRAD_GN_BWD", "1") != "0"


FUSE_GNB_CHAIN = os.environ.get("MADIFF_GNB_CHAIN", "1") != "0"


class ChainTag:
    """What two CONSECUTIVE ResidualTemporalBlocks share when the first one's output feeds the second one only (the model
    knows that; it hands the same tag to both, ResBlock chain_out / chain_in).  The first block's forward leaves what its
    blocks[1] GroupNorm backward needs; the second block's backward then runs its blocks[0] data gradient with that
    GroupNorm backward as the epilogue (madiff_conv_dgrad_gn_bwd_chain) and leaves the results for the first block's
    backward, which skips its standalone GroupNorm-backward launch."""
    __slots__ = ("y1", "st1", "geom1", "g1", "be1", "b1", "result", "dh")

    def __init__(self):
        self.y1 = self.st1 = self.geom1 = self.g1 = self.be1 = self.b1 = self.result = self.dh = None


def _dgrad_gn_bwd_fused(g1, w1, dy1, y0, stats0, geom0, gamma0, beta0, bias0, drb_slot=None, addend=None, want_dh=False):
    """ResidualTemporalBlock backward, middle part, as ONE launch (madiff_conv_dgrad_gn_bwd): the data gradient of
    blocks[1]'s convolution with blocks[0]'s GroupNorm + Mish backward as its epilogue; dh never reaches memory.
    -> (dy0, dgamma0, dbeta0, dbias0, drowbias) or None when the layer is not covered (bf16 build, k5 "same" convolution,
    channels a multiple of 128, whole groups inside 128 channels, T <= 32).
    want_dh (two consecutive blocks, ChainTag): the convolution is the NEXT block's blocks[0], dh = data gradient + addend
    is stored as well (madiff_conv_dgrad_gn_bwd_chain) and returned as a sixth value."""
    if not FUSE_DGRAD_GN_BWD or _PRECISION != "bf16" or not dy1.is_cuda or g1.spec != CONV5:
        return None
    n_seq, t, C, cpad, G, mode = geom0
    if want_dh and (not FUSE_GNB_CHAIN or mode == 1 or g1.n_seq != n_seq or g1.t_out != t or g1.c1 != g1.cin):
        return None
    if addend is not None and (not want_dh or addend.shape != (n_seq * t, C) or addend.dtype != act_dtype()
                               or addend.stride(1) != 1 or addend.stride(0) % 8 or addend.data_ptr() % 16):
        return None
    npad = pad32(g1.osum)
    if cpad != C or C % 128 or g1.osum != npad or npad % 64 or g1.cin != C or t > 32 or y0 is None:
        return None
    Cg = C // G
    if C % G or Cg % 8 or 128 % Cg or (Cg // 8) & (Cg // 8 - 1):
        return None
    sig = ("dgrad_gnb", n_seq, t, C, npad, G, mode) + (("chain", addend is not None) if want_dh else ())
    if _TUNE.get(sig) == "uncovered":
        return None
    Wf = _packed([w1], g1.spec, False)
    dy0 = torch.empty(n_seq * t, C, dtype=act_dtype(), device=dy1.device)
    dh_out = torch.empty(n_seq * t, C, dtype=act_dtype(), device=dy1.device) if want_dh else None
    drb = None
    if mode == 1:
        drb = slot_view(drb_slot, n_seq, C)
        if drb is None:
            drb = torch.empty(n_seq, C, dtype=torch.float32, device=dy1.device)

    def launch(variant, dst):
        if want_dh:
            return lib.call_rc("madiff_conv_dgrad_gn_bwd_chain", dy1.data_ptr(), dy1.stride(0), n_seq * t, npad, t,
                               Wf.data_ptr(), Wf.stride(0), g1.cin, C, y0.data_ptr(), y0.stride(0), gamma0.data_ptr(),
                               beta0.data_ptr(), stats0[0].data_ptr(), stats0[1].data_ptr(), G, dy0.data_ptr(),
                               dy0.stride(0), dst[0].data_ptr(), dst[1].data_ptr(),
                               dst[2].data_ptr() if bias0 is not None else None, _p(addend),
                               0 if addend is None else addend.stride(0), dh_out.data_ptr(), dh_out.stride(0), variant,
                               _stream())
        return lib.call_rc("madiff_conv_dgrad_gn_bwd", dy1.data_ptr(), dy1.stride(0), n_seq * t, npad, t, Wf.data_ptr(),
                           Wf.stride(0), g1.cin, C, y0.data_ptr(), y0.stride(0), gamma0.data_ptr(), beta0.data_ptr(),
                           stats0[0].data_ptr(), stats0[1].data_ptr(), G, dy0.data_ptr(), dy0.stride(0), dst[0].data_ptr(),
                           dst[1].data_ptr(), dst[2].data_ptr() if bias0 is not None else None, _p(drb),
                           0 if drb is None else drb.stride(0), variant, _stream())

    variant = _TUNE.get(sig)
    if _AUTOTUNE and _TUNE.get(sig + ("use",)) is False:      # measured: data gradient + standalone GroupNorm backward is faster here
        return None
    if variant is None and _AUTOTUNE and not torch.cuda.is_current_stream_capturing():
        scratch = torch.zeros(3, C, dtype=torch.float32, device=dy1.device)      # the timing launches accumulate here
        variant = _tuned_tile(sig, (1, 2), lambda v: launch(v, scratch))
        if variant >= 0:

            def unfused():
                dh = _dgrad_wt(dy1, n_seq * t, npad, g1.spec, t, t, Wf, g1.cin, 0, C, C, addend=addend)
                if dh is not None and addend is not None and not _dgrad_wt.fused_add:
                    dh = dh + addend
                if dh is not None:
                    lib.call("madiff_gn_mish_bwd", _dt(), y0.data_ptr(), y0.stride(0), dh.data_ptr(), dh.stride(0),
                             dy0.data_ptr(), dy0.stride(0), gamma0.data_ptr(), beta0.data_ptr(), stats0[0].data_ptr(),
                             stats0[1].data_ptr(), scratch[0].data_ptr(), scratch[1].data_ptr(),
                             scratch[2].data_ptr() if bias0 is not None else None, _p(drb),
                             C if drb is None else drb.stride(0), n_seq, t, C, cpad, G, _stream())

            unfused()
            t_fused, t_plain = _time_calls(lambda: launch(variant, scratch), 20), _time_calls(unfused, 20)
            _TUNE[sig + ("use",)] = bool(t_fused <= t_plain)
            if t_fused > t_plain:
                return None
    if variant is None or variant < 0:
        if variant is not None and _TUNE.get(sig) == -1:
            _TUNE[sig] = "uncovered"
            return None
        # not tuned (graph capture of an unseen shape / tuning disabled): fill the chip, as the fused forward does
        variant = 2 if (npad % 128 == 0 and (-(-n_seq // (192 // t))) * (C // 128) >= 200) else 1
    plist, views, bufs = _gn_param_bufs(gamma0, beta0, bias0, C, dy1.device)
    rc = [0]

    def run():
        rc[0] = launch(variant, bufs)
        return "conv_taps_bf16_kernel<GNB>" if rc[0] == 0 else ""

    # its own family in the timing records (rocprofv3 shows it as the GNB = true instantiation): the launch carries the
    # data gradient's FLOPs plus the GroupNorm backward's traffic (read y, write dy; dh stays on chip)
    _timed("conv_taps_bf16_kernel<GNB>", g1.flops, run,
           nbytes=float((2 + int(want_dh) + int(addend is not None)) * n_seq * t * C * dy1.element_size()))
    if rc[0]:
        _TUNE[sig] = "uncovered"
        return None
    outs = _gn_param_outs(plist, views, bufs, bias0)
    if want_dh:
        return dy0, outs[0], outs[1], outs[2], drb, dh_out
    return dy0, outs[0], outs[1], outs[2], drb


FUSE_CONV_GN = os.environ.get("MADIFF_FUSE_CONV_GN", "1") != "0"


def _conv_block_fused(x, x2, rowbias, res, spec: ConvSpec, t_in: int, G: int, w, b, gamma, beta, need_y: bool):
    """Conv1dBlock forward as ONE launch (madiff_conv_gn_mish_fwd: the all-taps k5 kernel with the GroupNorm + Mish
    epilogue).  Returns (out, y | None, stats | None, g, geom) or None when the layer is not covered."""
    if not FUSE_CONV_GN or _PRECISION != "bf16" or spec != CONV5 or not x.is_cuda:
        return None
    O, I, so, si, st = _strides(spec.kind, w)
    c1 = x.shape[1]
    cin = c1 + (x2.shape[1] if x2 is not None else 0)
    if O % 128 or O % G or cin % 64 or c1 % 64 or t_in > 32 or not (cin == pad32(I) or (x2 is not None and cin == I)):
        return None
    Cg = O // G
    if Cg % 8 or 128 % Cg or (Cg // 8) & (Cg // 8 - 1):
        return None
    _chk_cl(x)
    if x2 is not None:
        _chk_cl(x2)
    M = x.shape[0]
    n_seq = M // t_in
    W = _packed([w], spec, False)
    out = torch.empty(M, O, dtype=act_dtype(), device=x.device)
    y = torch.empty(M, O, dtype=act_dtype(), device=x.device) if need_y else None
    stats = torch.empty(2, n_seq * G, dtype=torch.float32, device=x.device) if need_y else None
    mode = 1 if rowbias is not None else (2 if res is not None else 0)
    if rowbias is not None:
        assert rowbias.dtype == torch.float32 and rowbias.stride(1) == 1
    if res is not None:
        _chk_cl(res)

    rv = rowvec_of(rowbias)
    if rv is not None:
        assert rv.dtype == torch.float32 and rv.is_contiguous() and rv.numel() >= O and not need_y

    def launch(variant):
        return lib.call_rc("madiff_conv_gn_mish_fwd_rv", x.data_ptr(), x.stride(0), _p(x2), 0 if x2 is None else x2.stride(0),
                           c1, M, cin, t_in, W.data_ptr(), W.stride(0), O, _p(b), _p(y), O, out.data_ptr(), O,
                           gamma.data_ptr(), beta.data_ptr(), None if stats is None else stats[0].data_ptr(),
                           None if stats is None else stats[1].data_ptr(), G, 1e-5, mode, _p(rowbias),
                           0 if rowbias is None else rowbias.stride(0), _p(res), 0 if res is None else res.stride(0), variant,
                           _p(rv), _stream())

    sig = ("convgn", M, cin, O, t_in, G, mode, need_y, x2 is not None)
    if _AUTOTUNE and _TUNE.get(sig + ("use",)) is False:      # measured: convolution + standalone GroupNorm is faster for this shape
        return None
    variant = _tuned_tile(sig, (1, 2), launch)
    if variant < 0:
        if _TUNE.get(sig) == -1:   # tuned before: no fused variant covers this layer
            return None
        # not tuned (graph capture of an unseen shape / tuning disabled): fill the chip
        variant = 2 if (-(-n_seq // (192 // t_in))) * (O // 128) >= 200 else 1
    elif _TUNE.get(sig + ("use",)) is None and _AUTOTUNE and not torch.cuda.is_current_stream_capturing():
        # fused or not?  With few sequences (small batches) the epilogue runs on a handful of workgroups and the
        # convolution's best tile is a small one: time the fused launch against convolution + standalone GroupNorm once
        # (bare library calls on preallocated outputs: the host must stay ahead of ~40 us kernels for the comparison to
        # be about the GPU)
        yy, oo = torch.empty(M, O, dtype=act_dtype(), device=x.device), torch.empty(M, O, dtype=act_dtype(), device=x.device)
        st = torch.empty(2, n_seq * G, dtype=torch.float32, device=x.device)

        def unfused():
            _gemm(x, x2, c1, M, cin, spec, t_in, t_in, spec.tr, W, O, O, b, None, yy, 0, None)
            lib.call("madiff_gn_mish_fwd", _dt(), yy.data_ptr(), yy.stride(0), oo.data_ptr(), oo.stride(0), gamma.data_ptr(),
                     beta.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), n_seq, t_in, O, O, G, 1e-5, mode, _p(rowbias),
                     0 if rowbias is None else rowbias.stride(0), _p(res), 0 if res is None else res.stride(0), _stream())

        unfused()                                # (tunes the plain convolution's tile on first use)
        t_fused, t_plain = _time_calls(lambda: launch(variant), 20), _time_calls(unfused, 20)
        _TUNE[sig + ("use",)] = bool(t_fused <= t_plain)
        if t_fused > t_plain:
            return None
    g = _Geom()
    g.flops = 2.0 * M * O * I * spec.taps
    rc = [0]

    def run():
        rc[0] = launch(variant)
        return "conv_taps_bf16_kernel" if rc[0] == 0 else ""      # "" = nothing was dispatched: no timing record

    _timed("conv_taps_bf16_kernel", g.flops, run)
    if rc[0]:
        return None
    g.spec, g.t_in, g.t_out, g.act, g.perm = spec, t_in, t_in, 0, None
    g.dims, g.osum, g.cin, g.c1, g.n_seq, g.I = [(O, I, so, si, st)], O, cin, c1, n_seq, I
    return out, y, stats, g, (n_seq, t_in, O, O, G, mode)


class ConvBlock(torch.autograd.Function):
    """Conv1dBlock as ONE autograd node: implicit-GEMM conv -> GroupNorm -> Mish (+ rowbias | + residual).
    Backward: fused GN/Mish backward (which also yields the conv-bias gradient), then data + weight gradient GEMMs.

    apply(x, x2, rowbias, res, spec, t_in, G, w, b, gamma, beta)"""

    @staticmethod
    def forward(ctx, x, x2, rowbias, res, spec: ConvSpec, t_in: int, G: int, w, b, gamma, beta, grad_on=True):
        # grad_on = torch.is_grad_enabled() at the call (inside forward() autograd is always off, and needs_input_grad
        # follows requires_grad whatever the grad mode): under no_grad -- the sampler -- the fused kernel must not write the
        # pre-norm output and the statistics, which only backward reads
        fused = _conv_block_fused(x, x2, rowbias, res, spec, t_in, G, w, b, gamma, beta,
                                  bool(grad_on) and any(ctx.needs_input_grad))
        if fused is not None:
            out, y, stats, g, geom = fused
        else:
            y, g = _conv_forward(x, x2, None, spec, t_in, 0, None, [w], [b])
            out, stats, geom = _gn_forward(y, gamma, beta, rowbias, res, g.n_seq, g.t_out, g.osum, G)
        ctx.save_for_backward(x, x2, y, stats, w, b, gamma, beta)
        ctx.g, ctx.geom = g, geom
        return out

    @staticmethod
    def backward(ctx, dout):
        x, x2, y, stats, w, b, gamma, beta = ctx.saved_tensors
        needs = ctx.needs_input_grad
        dy, dgamma, dbeta, dbias, drb, dres = _gn_backward(y, gamma, beta, stats, ctx.geom, dout, conv_bias=b)
        dx, dx2, _, gw, _ = _conv_backward(ctx.g, x, x2, [w], [b], dy, needs[0], needs[1], [needs[7]], "none", False)
        return (dx, dx2, drb, dres if ctx.geom[5] == 2 else None, None, None, None, gw[0], dbias, dgamma, dbeta, None)


class ResBlock(torch.autograd.Function):
    """ResidualTemporalBlock (temporal.py:161-200) as ONE autograd node:
        h = Conv1dBlock0(x) + temb[n];  out = Conv1dBlock1(h) + residual_conv(x)
    so that backward owns the order of its own launches: the input has two consumers (blocks[0] and residual_conv), and
    instead of torch adding the two contributions in a separate pass, the residual one is handed to blocks[0]'s
    data-gradient kernel as an addend.

    apply(x, x2, temb, t_in, G, w0, b0, g0, be0, w1, b1, g1, be1, rw, rb, chain_out, chain_in)
    (rw / rb None: identity residual; chain_out / chain_in: ChainTag | None, see there)"""

    @staticmethod
    def forward(ctx, x, x2, temb, t_in: int, G: int, w0, b0, g0, be0, w1, b1, g1, be1, rw, rb, chain_out=None, chain_in=None,
                grad_on=True):
        need = bool(grad_on) and any(ctx.needs_input_grad)      # (grad_on: see ConvBlock.forward)

        def block(xa, xb, rowbias, res, w, b, gamma, beta):
            fused = _conv_block_fused(xa, xb, rowbias, res, CONV5, t_in, G, w, b, gamma, beta, need)
            if fused is not None:
                return fused
            y, g = _conv_forward(xa, xb, None, CONV5, t_in, 0, None, [w], [b])
            out, stats, geom = _gn_forward(y, gamma, beta, rowbias, res, g.n_seq, g.t_out, g.osum, G)
            return out, y, stats, g, geom

        h, y0, st0, cg0, geom0 = block(x, x2, temb, None, w0, b0, g0, be0)
        if rw is not None:
            res, cgr = _conv_forward(x, x2, None, CONV1, t_in, 0, None, [rw], [rb])
        else:
            assert x2 is None
            res, cgr = x, None
        out, y1, st1, cg1, geom1 = block(h, None, None, res, w1, b1, g1, be1)
        ctx.save_for_backward(x, x2, y0, st0, h, y1, st1, w0, b0, g0, be0, w1, b1, g1, be1, rw, rb)
        ctx.geo = (cg0, geom0, cgr, cg1, geom1)
        ctx.temb_slot = grad_slot(temb)
        ctx.chain_out = ctx.chain_in = None
        if need and chain_out is not None and y1 is not None:
            chain_out.y1, chain_out.st1, chain_out.geom1 = y1, st1, geom1
            chain_out.g1, chain_out.be1, chain_out.b1 = g1, be1, b1
            ctx.chain_out = chain_out
        if need and chain_in is not None and chain_in.y1 is not None and x2 is None:
            ctx.chain_in = chain_in
        return out

    @staticmethod
    def backward(ctx, dout):
        x, x2, y0, st0, h, y1, st1, w0, b0, g0, be0, w1, b1, g1, be1, rw, rb = ctx.saved_tensors
        cg0, geom0, cgr, cg1, geom1 = ctx.geo
        needs = ctx.needs_input_grad
        dout = dout.contiguous()
        if dout.dtype != act_dtype():
            dout = dout.to(act_dtype())
        tag = ctx.chain_out
        if tag is not None and tag.result is not None:
            # the next block's backward already took dout (it produced it) through this block's blocks[1] GroupNorm backward
            if tag.dh is None or tag.dh.data_ptr() != dout.data_ptr():
                raise RuntimeError("chained residual blocks: the first block's output has another consumer")
            (dy1, dgamma1, dbeta1, dbias1), tag.result, tag.dh = tag.result, None, None
        else:
            dy1, dgamma1, dbeta1, dbias1, _, _ = _gn_backward(y1, g1, be1, st1, geom1, dout, conv_bias=b1)
        # blocks[1]'s data gradient with blocks[0]'s GroupNorm backward in its epilogue (one launch, no dh), else separately
        fused = _dgrad_gn_bwd_fused(cg1, w1, dy1, y0, st0, geom0, g0, be0, b0, ctx.temb_slot)
        if fused is not None:
            _, _, _, gw1, _ = _conv_backward(cg1, h, None, [w1], [b1], dy1, False, False, [needs[9]], "none", False)
        else:
            dh, _, _, gw1, _ = _conv_backward(cg1, h, None, [w1], [b1], dy1, True, False, [needs[9]], "none", False)
        gwr, gbr = [None], [None]
        if rw is not None:
            dxr, dx2r, _, gwr, gbr = _conv_backward(cgr, x, x2, [rw], [rb], dout, needs[0], needs[1], [needs[13]], "colsum",
                                                   False)
        else:
            dxr, dx2r = dout, None
        if fused is not None:
            dy0, dgamma0, dbeta0, dbias0, drb = fused
        else:
            dy0, dgamma0, dbeta0, dbias0, drb, _ = _gn_backward(y0, g0, be0, st0, geom0, dh, conv_bias=b0,
                                                                drb_slot=ctx.temb_slot)
        chained = None
        tin = ctx.chain_in
        if tin is not None and needs[0] and x2 is None:
            # this block's input is the previous block's output and nobody else's: blocks[0]'s data gradient (+ the residual
            # branch's gradient) IS the previous block's dout -- run that block's blocks[1] GroupNorm backward in the epilogue
            chained = _dgrad_gn_bwd_fused(cg0, w0, dy0, tin.y1, tin.st1, tin.geom1, tin.g1, tin.be1, tin.b1, None,
                                          addend=dxr, want_dh=True)
        if chained is not None:
            tin.result, tin.dh = (chained[0], chained[1], chained[2], chained[3]), chained[5]
            dx, dx2 = chained[5], None
            _, _, _, gw0, _ = _conv_backward(cg0, x, x2, [w0], [b0], dy0, False, False, [needs[5]], "none", False)
        else:
            dx, dx2, _, gw0, _ = _conv_backward(cg0, x, x2, [w0], [b0], dy0, needs[0], needs[1], [needs[5]], "none", False,
                                                addends=(dxr, dx2r))
        return (dx, dx2, drb if needs[2] else None, None, None, gw0[0], dbias0, dgamma0, dbeta0, gw1[0], dbias1, dgamma1,
                dbeta1, gwr[0], gbr[0], None, None, None)


def res_block(x, x2, temb, t_in: int, G: int, blk0, blk1, rw=None, rb=None, chain_out=None, chain_in=None):
    """blk0 / blk1 = (weight, bias, gamma, beta) of the two Conv1dBlocks; chain_out / chain_in: ChainTag | None."""
    if x2 is not None and x.shape[1] % 128 != 0:   # the split-source kernels need 128-aligned halves
        x, x2 = torch.cat((x, x2), dim=1), None
    return ResBlock.apply(x, x2, temb, t_in, G, *blk0, *blk1, rw, rb, chain_out, chain_in, torch.is_grad_enabled())


def conv_block(x, w, b, gamma, beta, spec: ConvSpec, t_in: int, G: int, x2=None, rowbias=None, res=None):
    if x2 is not None and x.shape[1] % 128 != 0:
        x, x2 = torch.cat((x, x2), dim=1), None
    return ConvBlock.apply(x, x2, rowbias, res, spec, t_in, G, w, b, gamma, beta, torch.is_grad_enabled())


class GNMish(torch.autograd.Function):
    """GroupNorm(8) + Mish (+ rowbias[n] | + residual) on a cl tensor."""

    @staticmethod
    def forward(ctx, y, gamma, beta, rowbias, res, n_seq: int, t: int, C: int, G: int):
        out, stats, geom = _gn_forward(y, gamma, beta, rowbias, res, n_seq, t, C, G)
        ctx.save_for_backward(y, gamma, beta, stats)
        ctx.geom = geom
        return out

    @staticmethod
    def backward(ctx, dout):
        y, gamma, beta, stats = ctx.saved_tensors
        dy, dgamma, dbeta, _, drb, dres = _gn_backward(y, gamma, beta, stats, ctx.geom, dout)
        return dy, dgamma, dbeta, drb, (dres if ctx.geom[5] == 2 else None), None, None, None, None


def _attn_label(kind: str, A, T, H, Cq, C) -> str:
    """Name of the kernel madiff_attn_{fwd,bwd} dispatches to (bench.py's per-family records follow rocprofv3's names)."""
    mfma = _PRECISION == "bf16" and lib._lib.madiff_attn_mfma_covers(lib.BF16, A, T, H, Cq, C) == 1
    return f"attn_{kind}_mfma_kernel" if mfma else f"attn_{kind}_v2_kernel"


class AttnCore(torch.autograd.Function):
    """softmax over agents of q.k, times v, then x + gamma*out; qkv is the fused projection [M, 2Cq + C]."""

    @staticmethod
    def forward(ctx, qkv, x, gamma, B, A, T, H, Cq, C, residual: bool):
        _chk_cl(qkv)
        _chk_cl(x)
        out = torch.empty_like(x)
        if out.shape[1] > C:
            out[:, C:].zero_()
        _timed(_attn_label("fwd", A, T, H, Cq, C), 0.0, lambda: lib.call(   # bytes: read qkv and x, write out
            "madiff_attn_fwd", _dt(), qkv.data_ptr(), qkv.stride(0), x.data_ptr(), x.stride(0), out.data_ptr(),
            out.stride(0), _p(gamma), B, A, T, H, Cq, C, int(residual), _stream()),
            nbytes=float(B * A * T * (2 * Cq + 3 * C) * x.element_size()))
        ctx.save_for_backward(qkv, gamma)
        ctx.geom = (B, A, T, H, Cq, C, residual)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, gamma = ctx.saved_tensors
        B, A, T, H, Cq, C, residual = ctx.geom
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        if dqkv.shape[1] > 2 * Cq + C:
            dqkv[:, 2 * Cq + C:].zero_()
        view = _sink_view(gamma) if residual else None
        dgamma = view if view is not None else (torch.zeros(1, dtype=torch.float32, device=qkv.device) if residual else None)
        _timed(_attn_label("bwd", A, T, H, Cq, C), 0.0, lambda: lib.call(   # bytes: read qkv and dout, write dqkv
            "madiff_attn_bwd", _dt(), qkv.data_ptr(), qkv.stride(0), dout.data_ptr(), dout.stride(0),
            dqkv.data_ptr(), dqkv.stride(0), _p(gamma), _p(dgamma), B, A, T, H, Cq, C, int(residual), _stream()),
            nbytes=float(B * A * T * (2 * (2 * Cq + C) + C) * qkv.element_size()))
        if view is not None:
            SINK.done(gamma)
            dgamma = None
        return dqkv, (dout if residual else None), dgamma, None, None, None, None, None, None, None


class AttnBlock(torch.autograd.Function):
    """TemporalSelfAttention / SelfAttention (temporal.py:57-124, helpers.py:77-124) as ONE autograd node: the fused
    Q/K/V projection (+ per-sequence time bias) and the attention core.  Backward then (a) takes the time-bias gradient
    from the attention-backward kernel itself (madiff_attn_bwd_rb: no pass over dqkv to sum over t), and (b) hands the
    residual path's gradient (out = x + gamma * attn: dx gets dout) to the projection's data-gradient kernel as an
    epilogue addend instead of a separate add.

    apply(x, time_proj | None, gamma | None, B, A, T, H, Cq, C, residual, wq, wk, wv, bq, bk, bv)"""

    @staticmethod
    def forward(ctx, x, time_proj, gamma, B, A, T, H, Cq, C, residual: bool, wq, wk, wv, bq, bk, bv):
        weights, biases = [wq, wk, wv], [bq, bk, bv]
        qkv, g = _conv_forward(x, None, time_proj, CONV1, T, 0, None, weights, biases)
        _chk_cl(qkv)
        out = torch.empty_like(x)
        if out.shape[1] > C:
            out[:, C:].zero_()
        _timed(_attn_label("fwd", A, T, H, Cq, C), 0.0, lambda: lib.call(
            "madiff_attn_fwd", _dt(), qkv.data_ptr(), qkv.stride(0), x.data_ptr(), x.stride(0), out.data_ptr(),
            out.stride(0), _p(gamma), B, A, T, H, Cq, C, int(residual), _stream()),
            nbytes=float(B * A * T * (2 * Cq + 3 * C) * x.element_size()))
        ctx.save_for_backward(x, qkv, gamma, *weights, *biases)
        ctx.g, ctx.geom, ctx.has_rb = g, (B, A, T, H, Cq, C, residual), time_proj is not None
        ctx.rb_slot = grad_slot(time_proj) if time_proj is not None else None
        return out

    @staticmethod
    def backward(ctx, dout):
        x, qkv, gamma = ctx.saved_tensors[:3]
        weights, biases = list(ctx.saved_tensors[3:6]), list(ctx.saved_tensors[6:9])
        B, A, T, H, Cq, C, residual = ctx.geom
        needs = ctx.needs_input_grad
        dout = dout.contiguous()
        if dout.dtype != act_dtype():
            dout = dout.to(act_dtype())
        dqkv = torch.empty_like(qkv)
        ncol = 2 * Cq + C
        if dqkv.shape[1] > ncol:
            dqkv[:, ncol:].zero_()
        view = _sink_view(gamma) if residual else None
        dgamma = view if view is not None else (torch.zeros(1, dtype=torch.float32, device=qkv.device) if residual else None)
        drb = None
        if ctx.has_rb and needs[1]:
            drb = slot_view(ctx.rb_slot, B * A, ncol)
            if drb is None:
                drb = torch.empty(B * A, ncol, dtype=torch.float32, device=qkv.device)
        _timed(_attn_label("bwd", A, T, H, Cq, C), 0.0, lambda: lib.call(
            "madiff_attn_bwd_rb", _dt(), qkv.data_ptr(), qkv.stride(0), dout.data_ptr(), dout.stride(0), dqkv.data_ptr(),
            dqkv.stride(0), _p(gamma), _p(dgamma), _p(drb), 0 if drb is None else drb.stride(0), B, A, T, H, Cq, C,
            int(residual), _stream()), nbytes=float(B * A * T * (2 * ncol + C) * qkv.element_size()))
        if view is not None:
            SINK.done(gamma)
            dgamma = None
        dx, _, _, gw, gb = _conv_backward(ctx.g, x, None, weights, biases, dqkv, needs[0], False,
                                          [needs[10], needs[11], needs[12]], "colsum", False,
                                          addends=(dout if residual else None, None))
        return (dx, drb, dgamma, None, None, None, None, None, None, None, *gw, *gb)


class Act(torch.autograd.Function):
    """Mish (kind 2) / ReLU (kind 1) on a cl tensor (pad columns stay zero: both map 0 -> 0)."""

    @staticmethod
    def forward(ctx, x, kind: int):
        x = x.contiguous()
        y = torch.empty_like(x)
        lib.call("madiff_act_fwd", _dt(), kind, x.data_ptr(), y.data_ptr(), x.numel(), _stream())
        ctx.save_for_backward(x)
        ctx.kind = kind
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        dx = torch.empty_like(x)
        lib.call("madiff_act_bwd", _dt(), ctx.kind, x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), _stream())
        return dx, None


def mish(x):
    return Act.apply(x, 2)


class PackTraj(torch.autograd.Function):
    """[B,T,A,F] fp32 (reference layout) -> cl [(b a) t, pad32(F)], differentiable."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = tuple(x.shape)
        return pack_traj(x, pad32(x.shape[-1]))

    @staticmethod
    def backward(ctx, d):
        B, T, A, F = ctx.shape
        d = d.contiguous()
        if d.dtype != act_dtype():
            d = d.to(act_dtype())
        return unpack_traj(d, B, T, A, F)


class _GradHolder:
    """Gradient buffer of a SplitCols input, shared with the consumers of its column blocks: a consumer whose backward
    kernel can write its block's gradient with a row stride (the row-bias gradients of GroupNorm / attention backward)
    asks for its slice and writes there, and SplitCols.backward then hands the buffer on without a concatenation."""

    def __init__(self, n, total, dtype, device):
        self.n, self.total, self.dtype, self.device = n, total, dtype, device
        self.buf = None

    def view(self, off, width):
        if self.buf is None:          # not zero-filled: SplitCols.backward defines every column nobody wrote
            self.buf = torch.empty(self.n, self.total, dtype=self.dtype, device=self.device)
        return self.buf[:, off:off + width]


def rowvec_of(t):
    """A per-channel fp32 vector that goes with a row-bias tensor (set by the sampler's per-step time projection,
    modules.SharedConvAttentionDeconv.forward_cl(proj=...)): consumers add rowbias[n, c] + rowvec[c].  None otherwise."""
    return None if t is None else getattr(t, "_madiff_rowvec", None)


def grad_slot(t):
    """(holder, offset, width) when `t` is a column block made by split_cols whose gradient may be written in place."""
    return getattr(t, "_madiff_slot", None)


def slot_view(slot, n_rows, width):
    """The destination view of a slot if it fits what the kernel will write ([n_rows, width] fp32, 16-byte aligned), else None."""
    if slot is None:
        return None
    holder, off, w = slot
    if w != width or holder.n != n_rows or holder.dtype != torch.float32 or off % 4 or holder.total % 4:
        return None
    return holder.view(off, width)


class SplitCols(torch.autograd.Function):
    """Views of column blocks of one fp32 matrix (the fused time-MLP output); backward hands on the shared gradient buffer
    the consumers wrote their blocks into (_GradHolder), or concatenates what it was given."""

    @staticmethod
    def forward(ctx, big, sizes):
        ctx.sizes = sizes
        ctx.meta = (big.shape[0], big.dtype, big.device)
        outs, off = [], 0
        for s in sizes:
            outs.append(big[:, off:off + s])
            off += s
        ctx.total = big.shape[1]
        ctx.holder = _GradHolder(big.shape[0], big.shape[1], big.dtype, big.device)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        n, dt, dev = ctx.meta
        holder, buf = ctx.holder, ctx.holder.buf
        holder.buf = None                      # one buffer per backward pass
        if buf is not None:
            off, es = 0, buf.element_size()
            for g, s in zip(grads, ctx.sizes):
                if g is None:
                    buf[:, off:off + s].zero_()          # a block without a consumer
                elif not (g.data_ptr() == buf.data_ptr() + off * es and g.shape == (n, s) and g.stride() == (ctx.total, 1)):
                    buf[:, off:off + s].copy_(g)         # a consumer that did not use its slot
                off += s
            if ctx.total > off:
                buf[:, off:].zero_()                     # pad columns
            return buf, None
        parts = [g if g is not None else torch.zeros(n, s, dtype=dt, device=dev) for g, s in zip(grads, ctx.sizes)]
        used = sum(ctx.sizes)
        if ctx.total > used:
            parts.append(torch.zeros(n, ctx.total - used, dtype=dt, device=dev))
        return torch.cat(parts, dim=1), None


def split_cols(big, sizes):
    """SplitCols.apply + tags on the outputs that let their consumers write gradients in place (training only)."""
    outs = list(SplitCols.apply(big, tuple(sizes)))
    node = outs[0].grad_fn if outs else None
    holder = getattr(node, "holder", None)
    if holder is not None:
        off = 0
        for o, s_ in zip(outs, sizes):
            o._madiff_slot = (holder, off, s_)
            off += s_
    return outs


def sinusoid(ts: torch.Tensor, freq: torch.Tensor, n_agents: int) -> torch.Tensor:
    """SinusoidalPosEmb for ts[B] (int64), repeated over agents: cl [B*A, 2*half]."""
    B, half = ts.shape[0], freq.shape[0]
    assert ts.dtype == torch.int64 and ts.is_contiguous()
    out = torch.empty(B * n_agents, pad32(2 * half), dtype=act_dtype(), device=ts.device)
    lib.call("madiff_sinusoid", _dt(), ts.data_ptr(), freq.data_ptr(), out.data_ptr(), out.stride(0), B, n_agents, half,
             _stream())
    return out


class MaskedStateLoss(torch.autograd.Function):
    """WeightedStateL2 + loss-mask reduction (helpers.py:243-250, diffusion.py:328-330).

    forward(pred, targ, w[T,A,F], mask[B,T,A] fp32) -> (loss float64 scalar, a0_loss fp32 scalar)
    """

    @staticmethod
    def forward(ctx, pred, targ, w, mask):
        B, T, A, F = pred.shape
        pred, targ = pred.contiguous(), targ.contiguous()
        sums = torch.empty(B, 3, dtype=torch.float32, device=pred.device)
        lib.call("madiff_mse_fwd", pred.data_ptr(), targ.data_ptr(), w.data_ptr(), mask.data_ptr(), sums.data_ptr(), B,
                 T * A, F, _stream())
        ctx.save_for_backward(pred, targ, w, mask, sums)
        s = sums.double()
        loss = (s[:, 0] / s[:, 1]).sum() / (B * F)
        a0 = sums[:, 2].sum() / (B * T * A * F)
        ctx.mark_non_differentiable(a0)
        return loss, a0

    @staticmethod
    def backward(ctx, gl, _ga):
        pred, targ, w, mask, sums = ctx.saved_tensors
        B, T, A, F = pred.shape
        go = gl.to(torch.float32).reshape(1).contiguous()
        dp = torch.empty_like(pred)
        lib.call("madiff_mse_bwd", pred.data_ptr(), targ.data_ptr(), w.data_ptr(), mask.data_ptr(), sums.data_ptr(),
                 go.data_ptr(), dp.data_ptr(), B, T * A, F, _stream())
        return dp, None, None, None


def inv_pairs(x: torch.Tensor, act: int, joint: bool) -> torch.Tensor:
    """Inputs of the inverse-dynamics MLP, [o_t | o_{t+1}] for t < T-1 (diffusion.py:381-384), straight from the batch's
    x [B,T,A,act+F] into the GEMM's activation layout: cl [B(T-1)A, pad32(2F)], or [B(T-1), pad32(A 2F)] for joint_inv."""
    B, T, A, W = x.shape
    F = W - act
    assert x.dtype == torch.float32 and x.is_contiguous()
    rows, ld = (B * (T - 1), pad32(A * 2 * F)) if joint else (B * (T - 1) * A, pad32(2 * F))
    out = torch.empty(rows, ld, dtype=act_dtype(), device=x.device)
    lib.call("madiff_inv_pairs", _dt(), x.data_ptr(), out.data_ptr(), ld, B, T, A, act, F, int(joint), _stream())
    return out


class FusedLoss(torch.autograd.Function):
    """GaussianDiffusion.loss after the two networks have run (diffusion.py:311,328-330,369-436,492): masked weighted
    state loss + inverse-dynamics loss (shared / joint MLP; MSE or legal-action cross-entropy + inv_acc) + 0.5 * sum, two
    launches forward and one backward.

    apply(eps [B,T,A,F], noise, w [T,A,F], loss_masks [B,T,A(,1)] fp64|fp32, pred fp32 [rows, ld], x [B,T,A,act+F],
          legal [B,T,A,nout] | None, (act, nout, discrete, joint[, segments])) -> (loss, a0_loss, inv_loss, inv_acc, last_loss)
    loss / inv_loss / inv_acc take the dtype of loss_masks (the reference's promotion), a0_loss is fp32.
    segments = S > 1: the batch holds S micro-batches of one optimizer step; `loss` is the mean of their S losses (its
    gradient = the accumulated gradient of training.py:147-152) while a0_loss / inv_loss / inv_acc / last_loss are the
    LAST micro-batch's values, i.e. what the reference's loop has in hand when it logs."""

    @staticmethod
    def forward(ctx, eps, noise, w, masks, pred, x, legal, geom):
        act, nout, discrete, joint = geom[:4]
        S = int(geom[4]) if len(geom) > 4 else 1
        B, T, A, F = eps.shape
        eps, noise, w, pred, x = eps.contiguous(), noise.contiguous(), w.contiguous(), pred.contiguous(), x.contiguous()
        masks = masks.contiguous()
        assert masks.dtype in (torch.float64, torch.float32) and masks.numel() == B * T * A
        assert eps.dtype == noise.dtype == pred.dtype == x.dtype == torch.float32 and x.shape == (B, T, A, act + F)
        assert S >= 1 and B % S == 0
        if legal is not None:
            legal = legal.to(torch.float32).contiguous()
            assert legal.shape == (B, T, A, nout)
        ld = pred.shape[1]
        if joint:
            assert pred.shape[0] == B * (T - 1) and ld >= A * nout
            rs, as_ = ld, nout
        else:
            assert pred.shape[0] == B * (T - 1) * A and ld >= nout
            rs, as_ = A * ld, ld
        dev = eps.device
        sums = torch.empty(B, 3, dtype=torch.float32, device=dev)
        ws = torch.empty(S * lib._lib.madiff_loss_ws_doubles(), dtype=torch.float64, device=dev)
        out = torch.empty(6 * (1 + S), dtype=torch.float64, device=dev)
        out_f = torch.empty(6 * (1 + S), dtype=torch.float32, device=dev)
        f64 = masks.dtype == torch.float64
        lib.call("madiff_loss_fwd_seg", eps.data_ptr(), noise.data_ptr(), w.data_ptr(), masks.data_ptr(), int(f64),
                 pred.data_ptr(), rs, as_, x.data_ptr(), _p(legal), B, T, A, F, act, nout, int(discrete), S, sums.data_ptr(),
                 ws.data_ptr(), out.data_ptr(), out_f.data_ptr(), _stream())
        ctx.save_for_backward(eps, noise, w, masks, pred, x, legal, sums, out)
        ctx.geom = (B, T, A, F, act, nout, int(discrete), rs, as_, int(f64), S)
        o = out if f64 else out_f
        k = 6 * S if S > 1 else 0               # the last micro-batch's six (the only six when S == 1)
        res = (o[0], out_f[k + 1], o[k + 2], o[k + 3], o[k])
        ctx.mark_non_differentiable(*res[1:])
        return res

    @staticmethod
    def backward(ctx, gl, *_unused):
        eps, noise, w, masks, pred, x, legal, sums, out = ctx.saved_tensors
        B, T, A, F, act, nout, discrete, rs, as_, f64, S = ctx.geom
        if gl.dtype not in (torch.float32, torch.float64):
            gl = gl.to(torch.float32)
        gl = gl.contiguous()
        d_eps = torch.empty_like(eps)
        d_pred = torch.empty_like(pred)
        lib.call("madiff_loss_bwd_seg", eps.data_ptr(), noise.data_ptr(), w.data_ptr(), masks.data_ptr(), f64, pred.data_ptr(),
                 rs, as_, x.data_ptr(), _p(legal), B, T, A, F, act, nout, discrete, S, sums.data_ptr(), out.data_ptr(),
                 gl.data_ptr(), int(gl.dtype == torch.float64), d_eps.data_ptr(), d_pred.data_ptr(), _stream())
        return d_eps, None, None, None, d_pred, None, None, None


def posterior_step(x, eps_c, eps_u, w, z, cond, mask_u8, coef, out, step_ptr=None, step_val=0):
    lib.call("madiff_posterior_step", x.data_ptr(), eps_c.data_ptr(), _p(eps_u), float(w), _p(z), _p(cond), _p(mask_u8),
             coef.data_ptr(), _p(step_ptr), int(step_val), out.data_ptr(), x.numel(), _stream())
    return out


def posterior_step_rng(x, eps_c, eps_u, w, z_all, rng, cond, mask_u8, coef, out, step_ptr=None, step_val=0):
    """posterior_step with the noise taken on the device: row `step` of z_all [n_steps, numel] (explicit draws) or, when
    z_all is None, Philox4x32-10 normals keyed by rng (int64[2] on the device = {seed, stream id})."""
    lib.call("madiff_posterior_step_rng", x.data_ptr(), eps_c.data_ptr(), _p(eps_u), float(w), _p(z_all), _p(rng), _p(cond),
             _p(mask_u8), coef.data_ptr(), _p(step_ptr), int(step_val), out.data_ptr(), x.numel(), _stream())
    return out


def inpaint_(x, cond, mask_u8):
    lib.call("madiff_inpaint", x.data_ptr(), cond.data_ptr(), mask_u8.data_ptr(), x.numel(), _stream())
    return x
