"""Are D2D memcpy nodes (contiguous Tensor.copy_ -> hipMemcpyAsync) ordered against kernel nodes of the same captured stream?"""
import torch
dev = torch.device("cuda:0")
n = 1 << 22
x = torch.zeros(n, device=dev); y = torch.zeros(n, device=dev); acc = torch.zeros(n, device=dev)
big = torch.full((12, 2624), float("nan"), device=dev); g = torch.ones(12, 256, device=dev); snap = torch.zeros(12, 2624, device=dev)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(2):
        x.fill_(1.0); y.copy_(x); acc.add_(y)
torch.cuda.synchronize()
acc.zero_()
graph = torch.cuda.CUDAGraph()
K = 50
with torch.cuda.graph(graph, capture_error_mode="thread_local"):
    for k in range(K):
        x.fill_(float(k + 1))          # kernel
        y.copy_(x)                     # contiguous D2D: memcpy node
        acc.add_(y)                    # kernel
    big.fill_(float("nan"))            # kernel
    big[:, 1088:1344].copy_(g)         # strided copy: kernel
    snap.copy_(big)                    # contiguous D2D: memcpy node
for rep in range(3):
    acc.zero_()
    graph.replay()
    torch.cuda.synchronize()
    want = K * (K + 1) / 2
    bad = int((acc != want).sum())
    print(f"replay {rep}: kernel->memcpy->kernel chain wrong elements {bad} of {n} (want {want}, got min {float(acc.min())} max {float(acc.max())});"
          f" strided-copy kernel -> memcpy snapshot NaNs in block: {int(torch.isnan(snap[:, 1088:1344]).sum())}")
