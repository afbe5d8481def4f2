"""debug: which parameters differ between the 2-rank run and the single-process run of tests/test_gpu_dp2.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.multiprocessing as mp
import test_gpu_dp2 as D

if __name__ == "__main__":
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(D._run, args=(2, 29733, out), nprocs=2, join=True)
    p0, e0 = out[0]
    from madiff_amd import ops
    dif, batch, draws = D._setup()
    opt, gs = D._steps(dif, batch, draws)
    ps = opt.flat_p.cpu()
    scale = float(ps.abs().max())
    rows = []
    for n, o, p in zip(opt.names, opt.offsets, opt.params):
        idx = opt.names.index(n)
        hi = opt.offsets[idx + 1] if idx + 1 < len(opt.offsets) else opt.total
        d = float((p0[o:hi] - ps[o:hi]).abs().max())
        rows.append((d / scale, n, o, hi - o))
    rows.sort(reverse=True)
    for r in rows[:25]:
        print("%.3e  %-60s off %9d n %8d" % r)
    print("buckets:", [(b["lo"], b["hi"], len(b["members"])) for b in opt.buckets][:50])
