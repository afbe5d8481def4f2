"""Latency of one planning step of the evaluator (DevicePlanner.act: conditions -> K-step guided sample -> inverse
dynamics -> actions) on a named workload: python scripts/planner_bench.py <workload> <num_envs> [ddim_steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from madiff_amd import ops
from madiff_amd.planner import DevicePlanner

wl, B = sys.argv[1], int(sys.argv[2])
ddim = int(sys.argv[3]) if len(sys.argv) > 3 else 0
bench.select_workload(wl, B)
ops.set_precision("bf16")
dev = torch.device("cuda:0")
torch.manual_seed(0)
dif = bench.build(dev)
if ddim:
    dif.set_ddim_scheduler(ddim)
A, HH, F = bench.A, bench.HH, bench.F_OBS
for dec in (False, True):
    pl = DevicePlanner(dif, decentralized_execution=dec)
    obs = torch.rand(B, HH + 1, A, F, device=dev) * 2 - 1
    ret = torch.rand(B, 1, A, device=dev)
    legal = (torch.rand(B, A, bench.NUM_ACTIONS, device=dev) < 0.7).float() if bench.DISCRETE else None
    for i in range(2):
        pl.act(obs, ret, None, legal)
    torch.cuda.synchronize()
    n = 3
    t0 = time.perf_counter()
    for i in range(n):
        a, s = pl.act(obs, ret, None, legal)
        a_host = a.cpu()                                   # what the simulator needs
    dt = (time.perf_counter() - t0) / n
    print(f"{wl} num_envs {B} {'decentralized' if dec else 'centralized'} {'DDIM ' + str(ddim) if ddim else 'DDPM K=200'}: "
          f"{1e3 * dt:.1f} ms per environment step ({B / dt:.1f} env-steps/s)", flush=True)
