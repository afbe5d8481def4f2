"""Diagnostic (not product code): what a hipEventQuery from ANOTHER thread does to a stream capture in each capture error
mode.  torch's ProcessGroupNCCL watchdog thread polls finished collectives with exactly that call (ATen/hip/HIPEvent.h
query(): hipEventQuery, and C10_HIP_CHECK on any error but hipErrorNotReady -> an uncaught exception in that thread ->
std::terminate -> abort()).  Here the query is made from a Python thread, where the error is a catchable RuntimeError."""
import sys
import threading

import torch

dev = torch.device("cuda:0")
x = torch.zeros(1024, device=dev)
for mode in ("thread_local", "relaxed", "global"):
    ev = torch.cuda.Event()
    ev.record()
    torch.cuda.synchronize()
    res = {}
    go, done = threading.Event(), threading.Event()

    def other():
        go.wait()
        try:
            res["query"] = ev.query()
        except Exception as e:          # noqa: BLE001
            res["query"] = "RAISED " + repr(e)[:300]
        done.set()

    th = threading.Thread(target=other)
    th.start()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    err = None
    try:
        with torch.cuda.stream(s):
            g.capture_begin(capture_error_mode=mode)
            x.add_(1)
            go.set()
            done.wait()
            x.add_(1)
            g.capture_end()
    except Exception as e:              # noqa: BLE001
        err = repr(e)[:300]
    th.join()
    print(f"mode={mode}: query from another thread -> {res.get('query')}; capture -> {'ok' if err is None else 'FAILED ' + err}",
          flush=True)
    torch.cuda.synchronize()
    if err is not None:
        try:
            del g
        except Exception:               # noqa: BLE001
            pass
sys.exit(0)
