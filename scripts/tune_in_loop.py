"""In-loop check of the per-tap GEMM tile choices of the sampler: for the heaviest 'fwd' signatures of the sampling loop
(1x1 / strided / transposed layers at the sampler's row counts) every tile variant is tried INSIDE the captured denoising
loop (a 10-step graph replayed 4 times) and the fastest is kept.  Isolated timing cannot see what the loop sees (the
tuner's five back-to-back launches of one layer against a real neighbourhood of kernels).
Prints the changes; with MADIFF_TUNE_CACHE set, the final choices are written there at exit.
usage: MADIFF_SAMPLE_GRAPH_STEPS=10 python scripts/tune_in_loop.py [episodes=1024] [n_signatures=6]"""
import os, sys, time
os.environ.setdefault("MADIFF_SAMPLE_GRAPH_STEPS", "10")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from madiff_amd import diffusion, lib, ops

eps = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nsig = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
bench.select_workload("spread", 256)
torch.manual_seed(1)
dif = bench.build(dev)
dif.n_timesteps = 40
sb = bench.synthetic_batch(eps, dev, seed=1000)
cond = {"x": sb["cond"]["x"].contiguous(), "masks": sb["cond"]["masks"].contiguous()}
ret = sb["returns"].contiguous()


def run():
    diffusion._SAMPLE_CACHE.pop(dif, None)                 # the choices are baked into the captured graph
    dif.conditional_sample(cond, returns=ret, verbose=False)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        dif.conditional_sample(cond, returns=ret, verbose=False)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


base = run()
n_steps = len(dif._sched_tables(dif.noise_scheduler, dev)["timesteps"])
print(f"baseline: {1e3 * base / n_steps:.3f} ms per denoising step ({n_steps} steps)", flush=True)
sigs = [k for k in ops._TUNE if isinstance(k, tuple) and k and k[0] == "fwd" and k[1] >= 30000 and k[3] != 5
        and isinstance(ops._TUNE[k], int)]
sigs.sort(key=lambda k: -(k[1] * k[2] * k[3] * k[4]))
for sig in sigs[:nsig]:
    cur = ops._TUNE[sig]
    best_tile, best_t = cur, run()
    trace = []
    for tile in range(14):                                 # per-tap variants
        if tile == cur:
            continue
        ops._TUNE[sig] = tile
        try:
            t = run()
        except Exception as e:                             # noqa: BLE001  (a variant that does not cover the layer)
            trace.append((tile, None))
            continue
        trace.append((tile, round(1e3 * t / n_steps, 3)))
        if t < best_t * 0.998:
            best_tile, best_t = tile, t
    ops._TUNE[sig] = best_tile
    print(f"{sig}: tile {cur} -> {best_tile}  ({1e3 * best_t / n_steps:.3f} ms/step)  tried {trace}", flush=True)
final = run()
print(f"final: {1e3 * final / n_steps:.3f} ms per denoising step (baseline {1e3 * base / n_steps:.3f})", flush=True)
