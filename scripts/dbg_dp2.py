import os, sys, time, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.multiprocessing as mp
import faulthandler
def log(rank, *a):
    print(f"[{time.strftime('%H:%M:%S')}] rank {rank}:", *a, flush=True)
def run(rank, world, port):
    faulthandler.dump_traceback_later(60, repeat=False, file=sys.stderr)
    import torch.distributed as dist
    import test_gpu_dp2 as Tm
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    Tm._host_staged_collectives(dist)
    log(rank, "pg up")
    dif, batch, draws = Tm._setup()
    half = slice(0, Tm.B // 2) if rank == 0 else slice(Tm.B // 2, Tm.B)
    batch, draws = Tm._slice(batch, half), Tm._slice(draws, half)
    from madiff_amd import ops
    from madiff_amd.trainer import FlatAdamEMA, GraphedTrainStep, train_step
    ema = copy.deepcopy(dif)
    opt = FlatAdamEMA(dif, ema, lr=1e-3, ema_decay=0.9, bucket_mb=0.25)
    opt.broadcast_parameters(0); log(rank, "broadcast done", float(sum(p.double().sum() for p in dif.parameters())))
    train_step(dif, opt, [{**batch, **draws}], do_ema=2); torch.cuda.synchronize(); log(rank, "step0 done", float(opt.flat_p.double().sum()))
    import collections
    cnt = collections.Counter()
    orig_bucket_of = opt._bucket_of
    sink = ops.SINK
    orig_notify = sink.notify
    def notify(p):
        cnt[opt.names[opt._index[id(p)]]] += 1
        orig_notify(p)
    sink.notify = notify
    launched = []
    orig_launch = opt._launch_bucket
    def launch(b):
        launched.append((opt.buckets.index(b), sum(cnt.values())))
        orig_launch(b)
    opt._launch_bucket = launch
    opt._hook_log = []
    train_step(dif, opt, [{**batch, **draws}], do_ema=1); torch.cuda.synchronize()
    if rank == 0:
        c2 = collections.Counter(i for i, _ in opt._hook_log)
        log(rank, "hook calls != 1:", [(opt.names[i], [s_ for j, s_ in opt._hook_log if j == i]) for i, v in c2.items() if v != 1][:12])
    if rank == 0:
        bad = {k: v for k, v in cnt.items() if v != 1}
        missing = [n for n in opt.names if n not in cnt]
        log(rank, "notify != 1:", bad, "never notified:", missing[:10], len(missing), "launch order:", launched, "n params", len(opt.names), "buckets", [(len(b["members"])) for b in opt.buckets])
    sink.notify = orig_notify; opt._launch_bucket = orig_launch; log(rank, "step1 done (hooks)", float(opt.flat_p.double().sum()))
    gs = GraphedTrainStep(dif, opt, batch, draws=draws); torch.cuda.synchronize(); log(rank, "captured", float(opt.flat_p.double().sum()))
    for i in range(2):
        gs.step(batch, do_ema=i % 2); torch.cuda.synchronize(); log(rank, "replay", i, float(opt.flat_p.double().sum()))
    log(rank, "checksum", float(opt.flat_p.double().sum()))
    dist.destroy_process_group()
if __name__ == "__main__":
    mp.spawn(run, args=(2, 29650), nprocs=2, join=True)
