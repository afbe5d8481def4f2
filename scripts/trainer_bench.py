"""Throughput of diffuser.utils.Trainer.train (data loader + pinned prefetch + H2D + graphed optimizer step) on a
synthetic dataset of a named workload: python scripts/trainer_bench.py <workload> <batch> <accumulate> <steps>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from madiff_amd import ops
from diffuser.utils import Trainer

wl, B, ACC, STEPS = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
bench.select_workload(wl, B)
ops.set_precision("bf16")
dev = torch.device("cuda:0")
torch.manual_seed(0)


class Windows(torch.utils.data.Dataset):
    def __init__(self, n):
        self.b = bench.synthetic_batch(n, torch.device("cpu"), 0)
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return {k: ({kk: vv[i] for kk, vv in v.items()} if isinstance(v, dict) else v[i]) for k, v in self.b.items()}


dif = bench.build(dev)
for graph in (True, False):
    tr = Trainer(dif, Windows(4096), None, train_batch_size=B, train_lr=2e-4, gradient_accumulate_every=ACC,
                 step_start_ema=2000, update_ema_every=10, log_freq=10 ** 9, sample_freq=0, save_freq=10 ** 9, eval_freq=0,
                 train_device="cuda:0", use_graph=graph)
    tr.log_freq = 10 ** 9
    tr.save = lambda: None
    tr.train(6)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train(STEPS)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{wl} batch {B} x accumulate {ACC}, {'graph' if graph else 'eager'}: {1e3 * dt / STEPS:.3f} ms per optimizer step, "
          f"{B * ACC * STEPS / dt:.0f} samples/s (data loader + H2D included)", flush=True)
    ops.SINK = None
