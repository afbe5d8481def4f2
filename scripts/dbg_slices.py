"""debug: are the optimizer slices issued during backward (eager and captured)?"""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from madiff_amd import ops
from madiff_amd.trainer import FlatAdamEMA, GraphedTrainStep, train_step
ops.set_precision("bf16")
dev = torch.device("cuda:0")
dif = bench.build(dev)
ema = copy.deepcopy(dif)
opt = FlatAdamEMA(dif, ema, lr=2e-4, ema_decay=0.995, bucket_mb=32.0)
batch = bench.synthetic_batch(256, dev, 0)
n = [0]
real = opt._adam_slice
def counted(*a, **k):
    n[0] += 1
    return real(*a, **k)
opt._adam_slice = counted
real_lb = opt._launch_bucket
def lb(b):
    print("   launch_bucket: announced", opt._announced is not None, "use_hip", opt.use_hip, "TIMER", ops.TIMER is not None, "dp", opt.dp, flush=True)
    return real_lb(b)
for i in range(3):
    n[0] = 0
    if i == 1:
        opt._launch_bucket = lb
    train_step(dif, opt, [batch], do_ema=1)
    opt._launch_bucket = real_lb
    print("eager step", i, "slices", n[0], "hooks", len(opt._hooks), "buckets", len(opt.buckets), "overlap", opt.overlap, "notify", ops.SINK.notify is not None, flush=True)
n[0] = 0
g = GraphedTrainStep(dif, opt, batch)
print("graphed ctor (eager + capture): slices", n[0], flush=True)
