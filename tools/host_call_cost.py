#!/usr/bin/env python3
"""Host cost of one stream-ordered piquant_quantize call (what bounds a small shard: at 8 GPUs the headline tensor is 3.4 M elements
per GPU, a ~4.5 us kernel): Context.quantize_ptr, the raw ctypes call with prebuilt arguments, the same launches replayed from a
hipGraph, and (numel 4096) the piquant.torch functions on device tensors.  Prints one JSON line."""
import ctypes
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "pi-quant_amd"))
import torch  # noqa: E402

import piquant  # noqa: E402
from piquant import DataType, RoundMode  # noqa: E402
from piquant._bootstrap import C_LIB as C  # noqa: E402

res = {}
ctx = piquant.Context()
s = torch.cuda.Stream()
ctx.set_stream(s.cuda_stream)
ctx.set_blocking(False)
for n in (4096, 3_408_000):
    sets = 192   # 3.3 GB in rotation at 3 408 000 elements, as bench.py --gpus 8 has: inputs AND outputs come from / go to HBM
    xs = [torch.empty(n, device="cuda").uniform_(-1, 1) for _ in range(sets)]
    qs = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(sets)]
    pin, pout = [t.data_ptr() for t in xs], [t.data_ptr() for t in qs]
    calls = 20000

    def timed(fn):
        for i in range(200):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(s)
        for i in range(calls):
            fn(i)
        t_host = time.perf_counter() - t0
        e1.record(s)
        torch.cuda.synchronize()
        return {"host_us_per_call": round(t_host / calls * 1e6, 3), "gpu_us_per_call": round(e0.elapsed_time(e1) * 1e3 / calls, 3)}

    with torch.cuda.stream(s):
        r = {}
        r["Context.quantize_ptr"] = timed(lambda i: ctx.quantize_ptr(pin[i % sets], DataType.F32, pout[i % sets], DataType.UINT8, n, 0.0078431377, 128, RoundMode.NEAREST, _device_ptrs=True))
        args = [(ctx._ctx, pin[k], 0, pout[k], 4, n, 0.0078431377, 128, 0) for k in range(sets)]
        fn = C.piquant_quantize
        r["ctypes piquant_quantize, prebuilt args"] = timed(lambda i: fn(*args[i % sets]))
        if n == 4096:   # the Python surface of the reference on device tensors: piquant.torch.* (native binding) and the additive dynamic call
            import piquant.torch as pt
            rec = torch.empty(16, dtype=torch.uint8, device="cuda")
            r["piquant.torch.quantize(out=)"] = timed(lambda i: pt.quantize(xs[i % sets], scale=0.0078431377, zero_point=128, dtype=torch.uint8, ctx=ctx, out=qs[i % sets]))
            r["piquant.torch.dequantize(out=)"] = timed(lambda i: pt.dequantize(qs[i % sets], scale=0.0078431377, zero_point=128, dtype=torch.float32, ctx=ctx, out=xs[i % sets]))
            r["piquant.torch.quantize_dynamic(out=, params=)"] = timed(lambda i: pt.quantize_dynamic(xs[i % sets], dtype=torch.uint8, ctx=ctx, out=qs[i % sets], params=rec))
            ctx.set_stream(s.cuda_stream)
            ctx.set_blocking(False)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for k in range(sets):
                fn(*args[k])
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(100):
            g.replay()
        e1.record(s)
        torch.cuda.synchronize()
        r[f"hipGraph of {sets} calls, replayed"] = {"gpu_us_per_call": round(e0.elapsed_time(e1) * 1e3 / (100 * sets), 3)}
    res[f"numel_{n}"] = r
print(json.dumps(res))
