"""The sampling leg of bench.py alone (spread shape, 1024 episodes, K=200, 2B-batch CFG), for a kernel trace:
  rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 scripts/sample_prof.py [episodes]
One warm-up call (allocator, pack cache, graph capture), one timed call; prints trajectories/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from madiff_amd.sharding import sample_sharded

eps = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
wl = sys.argv[2] if len(sys.argv) > 2 else "spread"
dev = torch.device("cuda:0")
bench.select_workload(wl, 256)
torch.manual_seed(1)
sampler = bench.build(dev)
sb = bench.synthetic_batch(eps, dev, seed=1000)
cond = {"x": sb["cond"]["x"].contiguous(), "masks": sb["cond"]["masks"].contiguous()}
ret = sb["returns"].contiguous()
sample_sharded(sampler, cond, ret, dst=0)
torch.cuda.synchronize()
t0 = time.perf_counter()
xs = sample_sharded(sampler, cond, ret, dst=0)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{wl}: {eps} episodes, K={bench.K_DIFF}: {dt:.3f} s = {eps / dt:.1f} trajectories/s", flush=True)
