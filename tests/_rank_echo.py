"""Helper for test_bench_launch.py: what one rank started by bench.launch_ranks sees.  Rank 0 prints one JSON line."""
import json
import os

import torch
import torch.distributed as dist

world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.ones(1)
dist.all_reduce(t)
if rank == 0:
    print(json.dumps({"n_gpus": world, "sum": int(t.item()), "local_rank": int(os.environ["LOCAL_RANK"]),
                      "master": os.environ["MASTER_ADDR"]}), flush=True)
dist.destroy_process_group()
