"""Fused Adam/EMA kernel alone on the bench model's parameter count: host-side vs device-side bias corrections."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madiff_amd import ops
lib = ops.lib
dev = torch.device("cuda:0")
n = 64_400_000 // 4 * 4
p, g, m, v, e = [torch.randn(n, device=dev) for _ in range(5)]
v.abs_()
sh = torch.empty(n, dtype=torch.bfloat16, device=dev)
state = torch.tensor([10, 0, 0, 0], dtype=torch.int32, device=dev)
x = torch.randn(4096, 4096, device=dev)
for _ in range(200):
    x @ x


def run(st, slices, flags=3):
    per = n // slices // 4 * 4
    for i in range(slices):
        lo = i * per
        lib.call("madiff_adam_ema", p.data_ptr() + 4 * lo, g.data_ptr() + 4 * lo, m.data_ptr() + 4 * lo, v.data_ptr() + 4 * lo,
                 e.data_ptr() + 4 * lo, per, 2e-4, 0.9, 0.999, 1e-8, 0.5, 0.5, 1.0, 0, 0.995,
                 None if st is None else st.data_ptr(), sh.data_ptr() + 2 * lo, flags, torch.cuda.current_stream().cuda_stream)


for name, st in (("host bias corrections", None), ("device step counter", state)):
    for slices in (1, 3, 12):
        run(st, slices)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(5):
                run(st, slices)
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 5
        print(f"{name:24s} {slices:2d} slices: {t * 1e3:7.1f} us per full pass, {34.0 * n / t / 1e6:6.0f} GB/s", flush=True)
