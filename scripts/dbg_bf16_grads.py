import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from madiff_amd import ops
from test_gpu_parity import build, T, SHAPES
from conftest import load_golden
g = load_golden("g2_denoiser_w16"); dev = torch.device("cuda:0")
for name in ("spread", "smac"):
    grads = {}
    for prec in ("fp32", "bf16"):
        ops.set_precision(prec)
        dif, _ = build(name, 16); net = dif.model
        x, t, r = T(g[name + ".x"], dev), T(g[name + ".t"], dev), T(g[name + ".returns"], dev)
        xg = x.clone().requires_grad_(True)
        y = net(xg, t, returns=r, use_dropout=True, dropout_mask=T(g[name + ".dropout_mask"], dev))
        y.backward(T(g[name + ".gy"], dev))
        grads[prec] = {k: p.grad.double().flatten() for k, p in net.named_parameters() if p.grad is not None}
        grads[prec]["__x"] = xg.grad.double().flatten()
    rows = []
    for k in grads["fp32"]:
        a, b = grads["bf16"][k], grads["fp32"][k]
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-30)); rel = float((a - b).norm() / (b.norm() + 1e-30))
        rows.append((rel, cos, float(b.norm()), k))
    rows.sort(reverse=True)
    print(name, "worst 12 by relative L2 error (rel, cos, |ref|, name):")
    for r_ in rows[:12]: print("   %.3f %.4f %.3e %s" % r_)
    big = [r_ for r_ in rows if r_[2] > 1e-3]
    print("   median rel over significant params: %.4f ; min cos: %.4f" % (np.median([r_[0] for r_ in big]), min(r_[1] for r_ in big)))
