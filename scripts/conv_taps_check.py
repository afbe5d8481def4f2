"""Correctness + speed of the all-taps k5 kernel (tile ids 14 = 192 rows, 15 = 96 rows) against the per-tap LDS-DMA
kernel (tile 0..13) and a torch fp32 conv1d, forward and data gradient, on the bench shapes plus ragged ones."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as Fn
from madiff_amd import ops
ops.set_precision("bf16")
dev = torch.device("cuda:0")
spec = ops.CONV5
QUICK = os.environ.get("QUICK", "0") == "1"


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


def fwd(x, Wf, y, M, T, ci, co, tile, bias=None, rowbias=None):
    return ops.lib.call_rc("madiff_gemm_fwd", 1, 0, x.data_ptr(), x.stride(0), None, 0, ci, M, ci, 5, T, T, 1, 2, 0,
                           Wf.data_ptr(), Wf.stride(0), co, co, ops._p(bias), ops._p(rowbias),
                           0 if rowbias is None else rowbias.stride(0), y.data_ptr(), y.stride(0), 0, 0, 0, tile, ops._stream())


def dgrad(gy, Wf, dx, M, T, ci, co, tile):
    return ops.lib.call_rc("madiff_gemm_dgrad_wt", gy.data_ptr(), gy.stride(0), M, co, 5, T, T, 1, 2, 1, Wf.data_ptr(),
                           Wf.stride(0), ci, 0, ci, ci, dx.data_ptr(), dx.stride(0), tile, 0, ops._stream())


# (N, T, Cin, Cout, count per forward)
shapes = [(1024, 6, 1024, 1024, 7), (1024, 6, 2048, 512, 1), (1024, 6, 512, 1024, 1), (1024, 6, 512, 512, 3),
          (1024, 12, 512, 512, 3), (1024, 12, 128, 512, 1), (1024, 12, 1024, 128, 1), (1024, 12, 128, 128, 3),
          (1024, 24, 128, 128, 4),
          (37, 7, 128, 192, 0), (61, 14, 192, 128, 0), (19, 28, 64, 64, 0), (5, 24, 128, 128, 0), (1, 6, 64, 128, 0)]
if QUICK:
    shapes = shapes[:1] + shapes[-5:]
ok = True
tot_old = [0.0, 0.0]
tot_new = [0.0, 0.0]
for N, T, ci, co, cnt in shapes:
    M = N * T
    torch.manual_seed(M + ci + co)
    x = torch.randn(M, ci, device=dev).to(torch.bfloat16)
    w = torch.randn(co, ci, 5, device=dev) / (ci * 5) ** 0.5
    bias = torch.randn(co, device=dev)
    rowbias = torch.randn(N, co, device=dev)
    Wf = ops._packed([w], spec, False)
    gy = torch.randn(M, co, device=dev).to(torch.bfloat16)
    flops = 2.0 * M * ci * co * 5
    # torch fp32 references on the bf16-rounded operands
    xr = x.float().view(N, T, ci).permute(0, 2, 1)
    wr = w.to(torch.bfloat16).float()
    yref = (Fn.conv1d(xr, wr, bias, padding=2) + rowbias[:, :, None]).permute(0, 2, 1).reshape(M, co)
    gr = gy.float().view(N, T, co).permute(0, 2, 1)
    dxref = Fn.conv_transpose1d(gr, wr, padding=2).permute(0, 2, 1).reshape(M, ci)
    line = f"N={N:4d} T={T:2d} {ci:4d}->{co:4d}:"
    best_old = [1e9, 1e9]
    for tile in list(range(14)) + [14, 15, 16]:
        y = torch.full((M, co), 7.0, dtype=torch.bfloat16, device=dev)
        rc = fwd(x, Wf, y, M, T, ci, co, tile, bias, rowbias)
        if rc != 0:
            continue
        err = (y.float() - yref).abs().max().item() / yref.abs().max().item()
        tf = timeit(lambda: fwd(x, Wf, y, M, T, ci, co, tile, bias, rowbias)) if cnt else 0.0
        dx = torch.full((M, ci), 7.0, dtype=torch.bfloat16, device=dev)
        td, derr = 0.0, 0.0
        if ops.lib.tile_info(tile)[3] and co % 64 == 0:
            rc = dgrad(gy, Wf, dx, M, T, ci, co, tile)
            if rc == 0:
                derr = (dx.float() - dxref).abs().max().item() / dxref.abs().max().item()
                td = timeit(lambda: dgrad(gy, Wf, dx, M, T, ci, co, tile)) if cnt else 0.0
        if tile >= 14:
            bad = err > 1e-2 or derr > 1e-2
            ok = ok and not bad
            line += f" | tile {tile}: fwd err {err:.1e} {flops / max(tf, 1e-9) / 1e9:5.0f} TF {tf * 1e3:6.1f} us, dgrad err {derr:.1e} {flops / max(td, 1e-9) / 1e9:5.0f} TF {td * 1e3:6.1f} us{' BAD' if bad else ''}"
            if cnt:
                tot_new[0] = tot_new[0]
        else:
            assert err < 1e-2 and derr < 1e-2, (tile, err, derr)
            if cnt:
                best_old[0] = min(best_old[0], tf)
                if td > 0:
                    best_old[1] = min(best_old[1], td)
        if tile >= 14 and cnt:
            key = "new%d" % tile
            globals().setdefault(key, [0.0, 0.0])
            globals()[key][0] += cnt * tf
            globals()[key][1] += cnt * td
    if cnt:
        tot_old[0] += cnt * best_old[0]
        tot_old[1] += cnt * best_old[1]
        line += f" | best per-tap: fwd {flops / best_old[0] / 1e9:5.0f} TF {best_old[0] * 1e3:6.1f} us, dgrad {flops / best_old[1] / 1e9:5.0f} TF {best_old[1] * 1e3:6.1f} us"
    print(line, flush=True)
print(f"per-step totals (ms): per-tap best fwd {tot_old[0]:.3f} dgrad {tot_old[1]:.3f}", {k: v for k, v in globals().items() if k.startswith('new1')})
print("ALL OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
