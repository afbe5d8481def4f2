"""Does training through diffuser.utils.Trainer actually learn, and does the bf16 build track the fp32 build?
Synthetic but structured data (per-agent sinusoid trajectories with random phase/frequency, actions = finite differences
of the first observation dims), tag shape, batch 256, identical seeds for both builds.
usage: python scripts/convergence_probe.py <steps> [precisions...]"""
import os, sys, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from madiff_amd import ops
from diffuser.utils import Trainer

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
PRECS = sys.argv[2:] or ["bf16", "fp32"]
bench.select_workload("tag", 256)
A, T, F, ACT = bench.A, bench.T, bench.F_OBS, bench.ACT
dev = torch.device("cuda:0")


class Waves(torch.utils.data.Dataset):
    def __init__(self, n, seed=0):
        g = torch.Generator().manual_seed(seed)
        t = torch.arange(T + 1).view(1, T + 1, 1, 1).float()
        freq = 0.15 + 0.35 * torch.rand(n, 1, A, 1, generator=g)
        phase = 2 * math.pi * torch.rand(n, 1, A, F, generator=g)
        amp = 0.3 + 0.6 * torch.rand(n, 1, A, 1, generator=g)
        obs = amp * torch.sin(freq * t + phase)                        # [n, T+1, A, F] in [-0.9, 0.9]
        act = (obs[:, 1:, :, :ACT] - obs[:, :-1, :, :ACT]) * 2.0        # what inverse dynamics has to recover
        self.x = torch.cat([act, obs[:, :-1]], dim=-1).contiguous()     # [n, T, A, ACT + F]
        self.ret = amp.mean(dim=(1, 3)).view(n, 1, A).contiguous()
        self.masks = torch.zeros(T, A, F, dtype=torch.bool)
        self.masks[:1] = True
        self.lm = torch.ones(T, A, 1, dtype=torch.float64)
        self.lm[0] = 0.0
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return {"x": self.x[i], "cond": {"x": self.x[i][..., ACT:].clone(), "masks": self.masks}, "loss_masks": self.lm,
                "attention_masks": torch.ones(T, A, 1, dtype=torch.float64), "returns": self.ret[i]}


data = Waves(4096)
for prec in PRECS:
    ops.set_precision(prec)
    torch.manual_seed(0)
    dif = bench.build(dev)
    tr = Trainer(dif, data, None, train_batch_size=256, train_lr=2e-4, gradient_accumulate_every=1, step_start_ema=200,
                 update_ema_every=10, log_freq=10 ** 9, sample_freq=0, save_freq=10 ** 9, eval_freq=0, train_device="cuda:0",
                 seed=0)
    tr.save = lambda: None
    torch.manual_seed(1)
    hist, t0 = [], time.perf_counter()
    losses = []
    real = tr._optimizer_step
    def wrapped(b, m):
        out = real(b, m)
        losses.append(out)
        return out
    tr._optimizer_step = wrapped
    chunk = max(1, STEPS // 10)
    for c in range(10):
        tr.train(chunk)
        l = torch.stack([x[0].float() for x in losses[-chunk:]]).mean().item()
        inv = torch.stack([x[1]["inv_loss"].float() for x in losses[-chunk:]]).mean().item()
        a0 = torch.stack([x[1]["a0_loss"].float() for x in losses[-chunk:]]).mean().item()
        hist.append((tr.step, l, a0, inv))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"[{prec}] {STEPS} steps in {dt:.1f} s ({256 * STEPS / dt:.0f} samples/s incl. data loading); "
          f"memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB peak", flush=True)
    for s, l, a0, inv in hist:
        print(f"[{prec}] step {s:5d}  loss {l:.4f}  denoise {a0:.4f}  inv_dyn {inv:.4f}", flush=True)
    # quality of the EMA model: sample a few trajectories conditioned on the first observation and compare the spectrum
    cond = {"x": data.x[:64, :, :, ACT:].to(dev).contiguous(), "masks": data.masks.unsqueeze(0).expand(64, -1, -1, -1).contiguous().to(dev)}
    xs = tr.ema_model.conditional_sample(cond, returns=data.ret[:64].to(dev), verbose=False)
    err = (xs - cond["x"]).abs().mean().item()
    base = cond["x"].abs().mean().item()
    print(f"[{prec}] EMA sample vs held trajectory: mean |error| {err:.4f} (mean |x| {base:.4f}); finite {bool(torch.isfinite(xs).all())}", flush=True)
    ops.SINK = None
    del tr, dif
    torch.cuda.empty_cache()
