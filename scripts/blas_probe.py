"""probe: hipBLASLt (through torch.matmul) on the plain 1x1 GEMM shapes of the step, graph-timed, vs this library's numbers"""
import os
import torch
dev = torch.device("cuda:0")
NSEQ = int(os.environ.get("NSEQ", 1024))     # sequences (1024 = the training step, 6144 = the 1024-episode sampler)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, M, K, N in (("T2 qkv 1024->1536", 6 * NSEQ, 1024, 1536), ("T1 qkv 512->768", 12 * NSEQ, 512, 768),
                      ("T2 res 2048->512", 6 * NSEQ, 2048, 512), ("T2 res 512->1024", 6 * NSEQ, 512, 1024),
                      ("T1 res 1024->128", 12 * NSEQ, 1024, 128)):
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    tf = timeit(lambda: torch.matmul(x, w.t()))
    td = timeit(lambda: torch.matmul(dy, w))
    tw = timeit(lambda: torch.matmul(dy.t(), x))
    gf = 2.0 * M * K * N / 1e9
    print(f"{name:20s} fwd {tf:6.1f} us {gf / tf / 1e3 * 1e3:6.0f} TF | dgrad {td:6.1f} us {gf / td:6.0f} TF | wgrad {tw:6.1f} us {gf / tw:6.0f} TF", flush=True)
