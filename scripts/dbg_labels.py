"""debug: per (label, flops) launch counts and average in-graph durations of one captured train step"""
import os, sys, copy, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from madiff_amd import ops
from madiff_amd.trainer import FlatAdamEMA, GraphedTrainStep, train_step
ops.set_precision("bf16")
dev = torch.device("cuda:0")
torch.manual_seed(0)
dif = bench.build(dev)
ema = copy.deepcopy(dif)
opt = FlatAdamEMA(dif, ema, lr=2e-4, ema_decay=0.995, bucket_mb=32.0)
batch = bench.synthetic_batch(256, dev, 0)
for i in range(3):
    train_step(dif, opt, [batch], do_ema=1)
gt = ops.GraphKernelTimer()
ops.GRAPH_TIMER = gt
g = GraphedTrainStep(dif, opt, batch)
ops.GRAPH_TIMER = None
for i in range(5):
    g.step(batch); gt.collect()
import ctypes
agg = collections.OrderedDict()
ms = ctypes.c_float()
from madiff_amd import lib
for name, flops, h0, h1 in gt.pairs:
    lib.call("madiff_event_elapsed_ms", h0, h1, ctypes.byref(ms))
    a = agg.setdefault((name, round(flops / 1e9, 2)), [0, 0.0])
    a[0] += 1; a[1] += ms.value
for (name, gf), (n, t) in agg.items():
    print(f"{name:32s} {gf:8.2f} GF x{n:3d}  avg {1e3 * t / n:7.1f} us  {gf / (t / n) :7.0f} TF")
