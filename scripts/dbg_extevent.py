import os, sys, ctypes, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madiff_amd import lib
dev = torch.device("cuda:0")
a = torch.randn(8192, 8192, device=dev)
b = torch.randn(8192, 8192, device=dev)
out1 = torch.empty_like(a); out2 = torch.empty_like(a); flag = torch.zeros(1, device=dev)
res = torch.zeros(1, device=dev)
evp = ctypes.c_void_p(); lib.call("madiff_event_create", ctypes.byref(evp)); ev = evp.value
side = torch.cuda.Stream()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        torch.mm(a, b, out=out1); torch.mm(out1, b, out=out2)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    torch.mm(a, b, out=out1)
    flag.fill_(1.0)
    lib.call("madiff_event_record_external", ev, torch.cuda.current_stream().cuda_stream)
    for _ in range(6):
        torch.mm(out1, b, out=out2)
torch.cuda.synchronize()
for it in range(3):
    flag.zero_(); res.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.replay()
    with torch.cuda.stream(side):
        lib.call("madiff_stream_wait_event", side.cuda_stream, ev)
        res.copy_(flag)          # must see 1.0 (runs after the event node), and should finish BEFORE the graph does
        e_side = torch.cuda.Event(enable_timing=True); e_side.record()
    e_main = torch.cuda.Event(enable_timing=True); e_main.record()
    e_side.synchronize(); t_side = time.perf_counter() - t0
    e_main.synchronize(); t_main = time.perf_counter() - t0
    print(f"iter {it}: flag seen {res.item()}  side done at {t_side*1e3:.2f} ms, graph done at {t_main*1e3:.2f} ms")
