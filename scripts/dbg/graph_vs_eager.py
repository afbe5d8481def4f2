"""Per-parameter difference between eager train_step and GraphedTrainStep twins (fp32 build, small model)."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import diffuser.models as DM
from madiff_amd import ops
from madiff_amd.host import batch_to_device
from madiff_amd.trainer import FlatAdamEMA, GraphedTrainStep, train_step
from oracle import madiff_oracle as O
ops.set_precision("fp32")
dev = torch.device("cuda:0")
torch.manual_seed(0)
A, T, Fd, B = 3, 24, 16, 4
net = DM.SharedConvAttentionDeconv(horizon=T, transition_dim=Fd, dim=32, dim_mults=(1, 4, 8), n_agents=A,
                                   returns_condition=True, condition_dropout=0.25)
dif = DM.GaussianDiffusion(net, n_agents=A, horizon=T, history_horizon=0, observation_dim=Fd, action_dim=5,
                           n_timesteps=50, returns_condition=True, hidden_dim=64).to(dev)
for a in net.self_attn:
    a.gamma.data.fill_(0.3)
batch = batch_to_device(O.synthetic_batch(B, T, A, Fd, 5, 0, seed=1), dev)
g = torch.Generator().manual_seed(5)
draws = dict(t=torch.randint(0, 50, (B,), generator=g).to(dev), noise=torch.randn(B, T, A, Fd, generator=g).to(dev),
             dropout_mask=(torch.rand(B, A, 1, generator=g) < 0.75).float().to(dev))
res = {}
STAMP = None
if os.environ.get("STAMP") == "1":
    import ctypes
    _st = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libstamp.so"))
    _st.dbg_stamp.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    ST_CNT = torch.zeros(1, dtype=torch.int32, device=dev)
    ST_SLOTS = torch.full((8192,), -1, dtype=torch.int64, device=dev)
    ST_LOG = []
    def STAMP(tag):
        if not torch.cuda.is_current_stream_capturing():
            return
        i = len(ST_LOG)
        st = torch.cuda.current_stream().cuda_stream
        ST_LOG.append((i, hex(st), tag))
        _st.dbg_stamp(ST_CNT.data_ptr(), ST_SLOTS.data_ptr(), i, st)
if os.environ.get("NOSLOT") == "1":
    ops.slot_view = lambda slot, n_rows, width: None
KEEP = os.environ.get("KEEP", "")
KEPT = []
if os.environ.get("POISON") == "1":
    _empty, _empty_like = torch.empty, torch.empty_like
    ALLOCS = []
    import traceback
    def _post(t):
        if t.is_floating_point() and t.is_cuda:
            if os.environ.get("SEQFILL") == "1":
                seq = len(ALLOCS)
                fr = [f"{f.filename.split('/')[-1]}:{f.lineno}:{f.name}" for f in traceback.extract_stack(limit=8)[:-2]
                      if "madiff" in f.filename]
                ALLOCS.append((seq, tuple(t.shape), str(t.dtype), hex(t.data_ptr()), t.numel() * t.element_size(),
                               hex(torch.cuda.current_stream().cuda_stream), torch.cuda.is_current_stream_capturing(), fr))
                t.fill_(float(100000 + seq))
            else:
                t.fill_(float("nan"))
            if STAMP is not None:
                STAMP("alloc %s %s %s" % (tuple(t.shape), hex(t.data_ptr()), ";".join(
                    f"{f.filename.split('/')[-1]}:{f.lineno}" for f in traceback.extract_stack(limit=7)[:-2] if "madiff" in f.filename)))
            if KEEP == "all" or (KEEP and t.numel() == int(KEEP)):
                KEPT.append(t)
        return t
    def empty(*a, **k):
        return _post(_empty(*a, **k))
    def empty_like(x, **k):
        return _post(_empty_like(x, **k))
    torch.empty, torch.empty_like = empty, empty_like
DBG = {}
SNAP = {k: torch.zeros(12, 256, device=dev) for k in ("after_kernel", "at_split_entry")}
CNT = {"n": 0}
_gb = ops._gn_backward
def gnb(y, gamma, beta, stats, geom, dout, conv_bias=None, drb_slot=None):
    out = _gb(y, gamma, beta, stats, geom, dout, conv_bias=conv_bias, drb_slot=drb_slot)
    if geom[2] == 256 and geom[5] == 1:
        if CNT["n"] == 0:
            SNAP["after_kernel"].copy_(out[4])
            DBG["drb_ptr"] = out[4].data_ptr()
        CNT["n"] += 1
    return out
ops._gn_backward = gnb
_sb = ops.SplitCols.backward
def sb(ctx, *grads):
    SNAP["at_split_entry"].copy_(grads[7])
    DBG["g7_ptr"] = grads[7].data_ptr()
    CNT["n"] = 0
    if STAMP is not None:
        STAMP("SplitCols.backward entry")
    out = _sb(ctx, *grads)
    if STAMP is not None:
        STAMP("SplitCols.backward exit")
    key = "buf"
    if key not in DBG:
        DBG[key] = torch.zeros_like(out[0])
    DBG[key].copy_(out[0])
    DBG["sizes"] = ctx.sizes
    return out
ops.SplitCols.backward = staticmethod(sb)
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for mode in ("eager", "graph"):
    m = copy.deepcopy(dif)
    opt = FlatAdamEMA(m, copy.deepcopy(m), lr=1e-3, ema_decay=0.9, bucket_mb=0.25)
    train_step(m, opt, [{**batch, **draws}], do_ema=2)
    log = []
    real = opt._adam_slice
    def spy(lo, hi, do_ema, device_state, stream, real=real, log=log):
        log.append((lo, hi, torch.cuda.is_current_stream_capturing(), device_state is not None, opt.step_count))
        return real(lo, hi, do_ema, device_state, stream)
    opt._adam_slice = spy
    if mode == "eager":
        for i in range(1 + nsteps):
            loss, _ = train_step(m, opt, [{**batch, **draws}], do_ema=0)
    else:
        if os.environ.get("MEMHIST"):
            torch.cuda.memory._record_memory_history(max_entries=400000, context="all", stacks="python")
        if os.environ.get("DOT"):
            _G = torch.cuda.CUDAGraph
            g_made = []
            def mk(*a, **k):
                o = _G(*a, keep_graph=True, **k)
                g_made.append(o)
                return o
            torch.cuda.CUDAGraph = mk
        gs = GraphedTrainStep(m, opt, batch, draws=draws)
        if os.environ.get("DOT"):
            import ctypes
            torch.cuda.CUDAGraph = _G
            hip = ctypes.CDLL("libamdhip64.so")
            hip.hipGraphDebugDotPrint.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint]
            raw = gs.graph.raw_cuda_graph()
            rc = hip.hipGraphDebugDotPrint(ctypes.c_void_p(raw), os.environ["DOT"].encode(), 1)   # 1 = verbose
            print("hipGraphDebugDotPrint rc", rc, "SNAP ptrs", {k: hex(v.data_ptr()) for k, v in SNAP.items()})
        for i in range(nsteps):
            loss, _ = gs.step(batch, do_ema=0)
    torch.cuda.synchronize()
    if mode == "graph" and STAMP is not None:
        sl = ST_SLOTS.cpu().view(-1, 2)
        by_stream = {}
        for i, st, tag in ST_LOG:
            by_stream.setdefault(st, []).append((i, int(sl[i, 0]), int(sl[i, 1]), tag))
        with open("gpurun_out/r3_stamps.txt", "w") as fh:
            for st, rows in by_stream.items():
                fh.write(f"stream {st}: {len(rows)} stamps\n")
                prev = -1
                for i, tk, clk, tag in rows:
                    flag = "   <<<< OUT OF ORDER" if tk < prev else ""
                    fh.write(f"  #{i:4d} ticket {tk:5d} clk {clk}  {tag}{flag}\n")
                    prev = max(prev, tk)
        print("stamps written:", len(ST_LOG))
    if mode == "graph" and os.environ.get("MEMHIST"):
        snap = torch.cuda.memory._snapshot()
        torch.cuda.memory._record_memory_history(enabled=None)
        ptr = DBG.get("g7_ptr")
        print(f"events touching [{ptr:#x}, +12288):")
        for ev in snap["device_traces"][0]:
            a, sz = ev.get("addr", 0), ev.get("size", 0)
            if a < ptr + 12288 and a + sz > ptr and ev["action"] in ("alloc", "free_requested", "free_completed", "free"):
                fr = [f"{f['filename'].split('/')[-1]}:{f['line']}:{f['name']}" for f in ev.get("frames", []) if "madiff" in f["filename"] or "graph_vs" in f["filename"]][:4]
                print(f"   {ev['action']:15s} addr {a:#x} size {sz:7d} stream {ev.get('stream')}  {fr}")
    res[mode] = (float(loss), {n: p.detach().clone() for n, p in m.named_parameters()}, opt)
    res[mode] += (opt.flat_m.clone(), opt.flat_g.clone(), DBG["buf"].clone())

    DBG.pop("buf")
    print(mode, "mid2 drb: NaN right after its kernel", int(torch.isnan(SNAP["after_kernel"]).sum()), "| NaN in grads[7] at SplitCols.backward entry",
          int(torch.isnan(SNAP["at_split_entry"]).sum()), "| same tensor:", DBG.get("drb_ptr") == DBG.get("g7_ptr"), hex(DBG.get("drb_ptr", 0)), hex(DBG.get("g7_ptr", 0)))
    ops.SINK = None
print("loss", res["eager"][0], res["graph"][0])
rows = []
for n, p in res["eager"][1].items():
    q = res["graph"][1][n]
    d = float((p - q).abs().max())
    rows.append((d, n, tuple(p.shape)))
rows.sort(reverse=True)
for d, n, shp in rows[:30]:
    print(f"{d:.3e}  {n}  {shp}")
opt = res["eager"][2]
me, mg = res["eager"][3], res["graph"][3]
print("first-moment differences per parameter slot (relative to the slot's max):")
rows = []
for n, o, o2 in zip(opt.names, opt.offsets, list(opt.offsets[1:]) + [opt.total]):
    a, b = me[o:o2], mg[o:o2]
    rows.append((float((a - b).abs().max()) / max(float(a.abs().max()), 1e-30), n, o, o2 - o))
rows.sort(reverse=True)
for d, n, o, ln in rows[:25]:
    print(f"{d:.3e}  {n}  off {o} len {ln}")
be, bg = res["eager"][5], res["graph"][5]
off = 0
print("d proj_all per column block (eager vs graph):")
for i, sz in enumerate(DBG["sizes"]):
    a, b = be[:, off:off + sz], bg[:, off:off + sz]
    print(f"  block {i:2d} off {off:5d} width {sz:4d}  max|eager| {float(a.abs().max()):.3e}  max|diff| {float((a - b).abs().max()):.3e}  nan eager {int(torch.isnan(a).sum())} graph {int(torch.isnan(b).sum())}")
    if i == 7 and os.environ.get("SEQFILL") == "1":
        bad = (a - b).abs() > 1e-3
        vals = torch.unique(b[bad]).tolist()
        print("   block 7: rows with garbage", torch.nonzero(bad.any(dim=1)).flatten().tolist(), "cols", int(bad.any(dim=0).sum()), "distinct garbage values", vals[:20])
        for v in vals[:20]:
            k = int(round(v)) - 100000
            if 0 <= k < len(ALLOCS):
                print("     ", ALLOCS[k])
        n0 = [x for x in ALLOCS if x[6]]
        print("   capture-time allocations:", len(n0), "first seq", n0[0][0] if n0 else None)
        print("   streams: current", hex(torch.cuda.current_stream().cuda_stream), {k[1]: hex(v.cuda_stream) for k, v in ops.AUX_STREAMS.items()})
        with open("gpurun_out/r3_allocs.txt", "w") as fh:
            for x in n0:
                fh.write(repr(x) + "\n")
    off += sz
