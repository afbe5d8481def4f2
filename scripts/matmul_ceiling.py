"""Same-box ceiling for the k5 convolution GEMMs: torch.matmul (hipBLASLt / rocBLAS) in bf16 on the im2col-EQUIVALENT dense
problem [M, 5*Cin] x [5*Cin, Cout] of each level of the headline workload (the library does not do the im2col, the epilogue
or the GroupNorm; this is only the MFMA-side ceiling a tuned library reaches for that shape on this box)."""
import sys, torch
dev = torch.device("cuda:0")
shapes = [("T=6  1024->1024", 6144, 5120, 1024), ("T=6  512->1024", 6144, 2560, 1024), ("T=12 512->512", 12288, 2560, 512),
          ("T=12 128->512", 12288, 640, 512), ("T=24 128->128", 24576, 640, 128), ("T=6 2048->512 (up, concat)", 6144, 10240, 512)]
for name, M, K, N in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        c = a @ b
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(20):
            c = a @ b
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 100
    print(f"{name:28s} M {M:6d} K {K:6d} N {N:5d}  {us:7.1f} us  {2.0 * M * K * N / us / 1e6:7.1f} TFLOP/s", flush=True)
print("-- sampler shapes (2 x 1024 episodes x 3 agents = 6144 sequences)")
for name, M, K, N in [("T=6  1024->1024", 36864, 5120, 1024), ("T=12 512->512", 73728, 2560, 512), ("T=24 128->128", 147456, 640, 128)]:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        c = a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        c = a @ b
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"{name:28s} M {M:6d} K {K:6d} N {N:5d}  {us:7.1f} us  {2.0 * M * K * N / us / 1e6:7.1f} TFLOP/s", flush=True)
