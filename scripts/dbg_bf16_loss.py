import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from madiff_amd import ops
from madiff_amd.host import batch_to_device
from oracle import madiff_oracle as O
from test_gpu_parity import build, T, SHAPES
from conftest import load_golden
g = load_golden("g4_loss"); dev = torch.device("cuda:0")
for tag, name, discrete, share in (("smac", "smac", True, True), ("spread_indep", "spread", False, False)):
    A, Tt, Fd, hh = SHAPES[name]
    grads = {}
    for prec in ("fp32", "bf16"):
        ops.set_precision(prec)
        dif, act = build(name, 16, 200, discrete, share)
        batch = batch_to_device(O.synthetic_batch(3, Tt, A, Fd, act, hh, discrete, 14 if discrete else 0, seed=11, visible=g[tag + ".visible"]), dev)
        loss, info = dif.loss(**batch, t=T(g[tag + ".t"], dev), noise=T(g[tag + ".noise"], dev), dropout_mask=T(g[tag + ".dropout_mask"], dev))
        loss.backward()
        print(tag, prec, "loss", loss.item(), "ref", float(g[tag + ".loss"]), {k: float(v) for k, v in info.items()})
        grads[prec] = {k: p.grad.double().flatten() for k, p in dif.named_parameters() if p.grad is not None}
    rows = []
    for k in grads["fp32"]:
        a, b = grads["bf16"][k], grads["fp32"][k]
        rows.append((float(abs(a.norm() - b.norm()) / (b.norm() + 1e-30)), float((a - b).norm() / (b.norm() + 1e-30)), float(b.norm()), k))
    rows.sort(reverse=True)
    for r_ in rows[:8]: print("   dnorm %.3f relL2 %.3f |ref| %.3e %s" % r_)
