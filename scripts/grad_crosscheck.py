"""Full-size gradient cross-check between the two builds: same weights, batch and draws; per-parameter relative L2
difference and cosine between the bf16-build and the fp32-build gradients (expected: a few percent / > 0.99)."""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from madiff_amd import ops
wl = sys.argv[1] if len(sys.argv) > 1 else "tag"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
bench.select_workload(wl, B)
dev = torch.device("cuda:0")
torch.manual_seed(0)
ops.set_precision("bf16")
dif = bench.build(dev)
batch = bench.synthetic_batch(B, dev, 0)
g = torch.Generator().manual_seed(3)
draws = dict(t=torch.randint(0, 200, (B,), generator=g).to(dev), noise=torch.randn(B, bench.T, bench.A, bench.F_OBS, generator=g).to(dev),
             dropout_mask=(torch.rand(B, bench.A, 1, generator=g) < 0.75).float().to(dev))
grads = {}
for prec in ("bf16", "fp32"):
    ops.set_precision(prec)
    dif.zero_grad(set_to_none=True)
    loss, info = dif.loss(**batch, **draws)
    loss.backward()
    torch.cuda.synchronize()
    grads[prec] = {k: p.grad.detach().double().clone() for k, p in dif.named_parameters() if p.grad is not None}
    print(prec, "loss", float(loss), {k: float(v) for k, v in info.items()}, flush=True)
rows = []
for k, a in grads["bf16"].items():
    b = grads["fp32"][k]
    nb = float(b.norm())
    rel = float((a - b).norm()) / max(nb, 1e-30)
    cos = float((a * b).sum()) / max(float(a.norm()) * nb, 1e-30)
    rows.append((rel, cos, nb, k))
rows.sort(reverse=True)
print("worst 25 parameters by relative L2 difference (rel, cosine, |g_fp32|, name):")
for r in rows[:25]:
    print("  %.3e  %.5f  %.3e  %s" % r)
import statistics
print("median rel", statistics.median(r[0] for r in rows), "min cosine", min(r[1] for r in rows if r[2] > 1e-6))
