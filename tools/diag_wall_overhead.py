#!/usr/bin/env python3
"""What bench.py's wall clock carries beyond the kernels at the driver's K = 20: the timed loop as bench.py runs it (two HIP events inside the wall
window, busy-wait on the second), with the end event only, and with torch.cuda.synchronize() only.  Measured: 23.8-24.4 / 23.4-24.0 / 23.5-23.8 us per
step against 23.0-23.4 us of kernel time -- the events cost nothing measurable; what is left (~0.5-1 us per step) is the launch from an idle queue and
the detection of the end."""
import sys, time
sys.path.insert(0, "pi-quant_amd")
import torch, piquant
from piquant import DataType, RoundMode
from piquant._bootstrap import C_LIB
N, SETS, K = 27_264_000, 24, 20
ctx = piquant.Context(); s = torch.cuda.Stream(); ctx.set_stream(s.cuda_stream); ctx.set_blocking(False); ctx.assume_device_pointers(True)
xs = [torch.empty(N, device="cuda").uniform_(-1, 1) for _ in range(SETS)]
qs = [torch.empty(N, dtype=torch.uint8, device="cuda") for _ in range(SETS)]
args = [(ctx._ctx, xs[k].data_ptr(), 0, qs[k].data_ptr(), 4, N, 0.0078431377, 128, 0) for k in range(SETS)]
f = C_LIB.piquant_quantize
def step(i): f(*args[i % SETS])
with torch.cuda.stream(s):
    for i in range(2000): step(i)
    torch.cuda.synchronize()
    def with_events():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); t0 = time.perf_counter(); e0.record(s)
        for i in range(K): step(i)
        e1.record(s)
        while not e1.query(): pass
        t1 = time.perf_counter(); torch.cuda.synchronize()
        return (t1 - t0) * 1e6 / K, e0.elapsed_time(e1) * 1e3 / K
    def end_event_only():
        e1 = torch.cuda.Event()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(K): step(i)
        e1.record(s)
        while not e1.query(): pass
        t1 = time.perf_counter(); torch.cuda.synchronize()
        return (t1 - t0) * 1e6 / K
    def plain_sync():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(K): step(i)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        return (t1 - t0) * 1e6 / K
    for rep in range(4):
        a = with_events(); b = end_event_only(); c = plain_sync()
        print(f"with events: wall {a[0]:.3f} us/step, events {a[1]:.3f};  end event only: {b:.3f};  synchronize only: {c:.3f}")
