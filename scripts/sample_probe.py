"""One conditional_sample call of the bench workload (B episodes, K=200) for rocprofv3 kernel traces."""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from madiff_amd import ops
ops.set_precision("bf16")
dev = torch.device("cuda:0")
B = int(os.environ.get("SB", 256))
torch.manual_seed(0)
dif = bench.build(dev)
batch = bench.synthetic_batch(B, dev, 0)
cond = {"x": batch["cond"]["x"].contiguous(), "masks": batch["cond"]["masks"].contiguous()}
ret = batch["returns"].contiguous()
import time
for i in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    xs = dif.conditional_sample(cond, returns=ret, verbose=False)
    torch.cuda.synchronize(); print("call", i, "%.3f s" % (time.perf_counter() - t0), "->", B / (time.perf_counter() - t0), "traj/s")
