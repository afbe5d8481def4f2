"""debug: per-step parameter movement, 2-rank vs single-process (tests/test_gpu_dp2.py sequence)"""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.multiprocessing as mp
import test_gpu_dp2 as D


def steps(tag, dif, batch, draws):
    from madiff_amd import ops
    from madiff_amd.trainer import FlatAdamEMA, GraphedTrainStep, train_step
    ema = copy.deepcopy(dif)
    opt = FlatAdamEMA(dif, ema, lr=1e-3, ema_decay=0.9, bucket_mb=0.25)
    opt.broadcast_parameters(0)
    prev = [None]
    def rep(what):
        torch.cuda.synchronize()
        p = opt.flat_p.double().clone()
        mv = float((p - prev[0]).abs().mean()) if prev[0] is not None else float("nan")
        prev[0] = p
        print(f"[{tag}] {what}: step_count {opt.step_count} mean|dp| {mv:.6e} sum p {float(p.sum()):.9f} |g| {float(opt.flat_g.abs().sum()):.3e}", flush=True)
    train_step(dif, opt, [{**batch, **draws}], do_ema=2); rep("eager1")
    train_step(dif, opt, [{**batch, **draws}], do_ema=1); rep("eager2")
    gs = GraphedTrainStep(dif, opt, batch, draws=draws); rep("graph-init")
    for i in range(2):
        gs.step(batch, do_ema=i % 2); rep(f"replay{i}")
    ops.SINK = None


def run(rank, world, port):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    D._host_staged_collectives(dist)
    dif, batch, draws = D._setup()
    half = slice(0, D.B // 2) if rank == 0 else slice(D.B // 2, D.B)
    steps(f"dp{rank}", dif, D._slice(batch, half), D._slice(draws, half))
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(run, args=(2, 29744), nprocs=2, join=True)
    dif, batch, draws = D._setup()
    steps("single", dif, batch, draws)
