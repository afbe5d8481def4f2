"""Micro-benchmark of the implicit-GEMM kernels on the layer shapes of the bench workload (B=256, A=4 -> N=1024):
forward for every output tile the library offers, data gradient from the forward weight layout (3 tiles), weight
gradient.  Prints TFLOP/s per variant and the per-step totals with the best tile of each layer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madiff_amd import ops
ops.set_precision("bf16")
dev = torch.device("cuda:0")
N = int(os.environ.get("NSEQ", 1024))
# (name, T, Cin, Cout, spec, count per forward)
shapes = [("T2 1024->1024 k5", 6, 1024, 1024, ops.CONV5, 7), ("T2 2048->512 k5", 6, 2048, 512, ops.CONV5, 1),
          ("T2 512->1024 k5", 6, 512, 1024, ops.CONV5, 1), ("T2 512->512 k5", 6, 512, 512, ops.CONV5, 3),
          ("T1 512->512 k5", 12, 512, 512, ops.CONV5, 3), ("T1 128->512 k5", 12, 128, 512, ops.CONV5, 1),
          ("T1 1024->128 k5", 12, 1024, 128, ops.CONV5, 1), ("T1 128->128 k5", 12, 128, 128, ops.CONV5, 3),
          ("T0 128->128 k5", 24, 128, 128, ops.CONV5, 4),
          ("T2 1024->1536 1x1", 6, 1024, 1536, ops.CONV1, 2), ("T1 512->768 1x1", 12, 512, 768, ops.CONV1, 1),
          ("T2 512->1024 1x1", 6, 512, 1024, ops.CONV1, 1), ("T2 2048->512 1x1", 6, 2048, 512, ops.CONV1, 1),
          ("T1 128->512 1x1", 12, 128, 512, ops.CONV1, 1), ("T1 1024->128 1x1", 12, 1024, 128, ops.CONV1, 1)]
tot = [0.0, 0.0, 0.0]
TILES = ["%dx%d/%d" % ops.lib.tile_info(t)[:3] for t in range(ops.lib.tile_count())]
WT_TILES = ops.lib.tiles(wt=True)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, T, ci, co, spec, cnt in shapes:
    M = N * T
    x = torch.randn(M, ci, device=dev).to(torch.bfloat16)
    w = (torch.randn(co, ci, spec.taps, device=dev) / (ci * spec.taps) ** 0.5)
    Wf = ops._packed([w], spec, False)
    y = torch.empty(M, co, dtype=torch.bfloat16, device=dev)
    gy = torch.randn(M, co, device=dev).to(torch.bfloat16)
    dx = torch.empty(M, ci, dtype=torch.bfloat16, device=dev)
    flops = 2.0 * M * ci * co * spec.taps
    st = ops._stream()
    fw = []
    for tile in range(len(TILES)):
        f = lambda: ops.lib.call_rc("madiff_gemm_fwd", 1, 0, x.data_ptr(), ci, None, 0, ci, M, ci, spec.taps, T, T,
                                    1, spec.pad, 0, Wf.data_ptr(), Wf.stride(0), co, co, None, None, 0,
                                    y.data_ptr(), co, 0, 0, 0, tile, st)
        fw.append(timeit(f) if f() == 0 else float("inf"))      # rc 1: variant does not cover this layer
    dg = []
    for tile in WT_TILES:
        f = lambda: ops.lib.call_rc("madiff_gemm_dgrad_wt", gy.data_ptr(), co, M, co, spec.taps, T, T, 1,
                                    spec.pad, 1, Wf.data_ptr(), Wf.stride(0), ci, 0, ci, ci, dx.data_ptr(),
                                    ci, tile, 0, st)
        dg.append(timeit(f) if f() == 0 else float("inf"))
    dW = torch.zeros(co, spec.taps * ci, dtype=torch.float32, device=dev)
    wg = {}
    steps = (M + 63) // 64
    for tile, (bo, bc, all_taps) in enumerate(ops._WGRAD_TILES):
        for splits in (1, 2, 3, 4, 6, 8, 16):
            if splits > 1 and (splits > steps // 4 or (co // bo) * (ci // bc) * (1 if all_taps else spec.taps) * splits > 2048):
                continue
            if ops.lib.call_rc("madiff_gemm_wgrad", 1, gy.data_ptr(), co, co, x.data_ptr(), ci, None, 0, ci, M, ci, spec.taps,
                               T, T, 1, spec.pad, 0, dW.data_ptr(), dW.stride(0), co, 1, tile, splits, st) != 0:
                break
            wg[(tile, splits)] = timeit(lambda: ops.lib.call_rc(
                "madiff_gemm_wgrad", 1, gy.data_ptr(), co, co, x.data_ptr(), ci, None, 0, ci, M, ci, spec.taps, T, T, 1,
                spec.pad, 0, dW.data_ptr(), dW.stride(0), co, 1, tile, splits, st), n=10)
    kbest = min(wg, key=wg.get)
    tw = wg[kbest]
    wdesc = " ".join(f"{t}/{sp}:{flops / v / 1e9:.0f}" for (t, sp), v in wg.items())
    tot[0] += cnt * min(fw)
    tot[1] += cnt * min(dg)
    tot[2] += cnt * tw
    f = " ".join(f"{flops / t / 1e9:6.0f}" for t in fw)
    d = " ".join(f"{flops / t / 1e9:6.0f}" for t in dg)
    print(f"{name:20s} x{cnt} fwd TF [{f}] best {TILES[fw.index(min(fw))]:>9s} {min(fw) * 1e3:6.1f} us | dgrad_wt TF [{d}] best {TILES[WT_TILES[dg.index(min(dg))]]:>9s} "
          f"{min(dg) * 1e3:6.1f} us | wgrad best tile/splits {kbest} {tw * 1e3:6.1f} us {flops / tw / 1e9:6.0f} TF", flush=True)
    print("      wgrad TF by tile/splits:", wdesc, flush=True)
print("tiles:", TILES)
print(f"weighted totals per step (ms): fwd {tot[0]:.3f} dgrad {tot[1]:.3f} wgrad {tot[2]:.3f}")
