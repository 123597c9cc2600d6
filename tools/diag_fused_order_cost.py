#!/usr/bin/env python3
"""What the per-device ordering of grid-barrier launches costs (context.cpp FusedLaunchOrder): quantize_dynamic fp32 -> uint8 at numel 27 264 000, cold
rotation, HIP events, (a) a process that has only ever used one stream, (b) after one fused launch on a second stream (from then on every fused launch
records an event behind itself and waits for the previous one's)."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "pi-quant_amd"))
import torch  # noqa: E402

import piquant  # noqa: E402
from piquant import DataType, RoundMode  # noqa: E402

N, SETS = 27_264_000, 24
ctx = piquant.Context()
s = torch.cuda.Stream()
ctx.set_stream(s.cuda_stream)
ctx.set_blocking(False)
xs = [torch.empty(N, device="cuda").uniform_(-1, 1) for _ in range(SETS)]
qs = [torch.empty(N, dtype=torch.uint8, device="cuda") for _ in range(SETS)]
rec = torch.empty(16, dtype=torch.uint8, device="cuda")
pi, po, pr = [t.data_ptr() for t in xs], [t.data_ptr() for t in qs], rec.data_ptr()
torch.cuda.synchronize()


def timed(reps=300):
    with torch.cuda.stream(s):
        for i in range(30):
            ctx.quantize_dynamic_ptr(pi[i % SETS], DataType.F32, po[i % SETS], DataType.UINT8, N, pr, RoundMode.NEAREST, _device_ptrs=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for i in range(reps):
            ctx.quantize_dynamic_ptr(pi[i % SETS], DataType.F32, po[i % SETS], DataType.UINT8, N, pr, RoundMode.NEAREST, _device_ptrs=True)
        e1.record(s)
        torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 1e3 / reps, 3)


out = {"one_stream_us": [timed() for _ in range(3)]}
other = piquant.Context()
s2 = torch.cuda.Stream()
other.set_stream(s2.cuda_stream)
other.set_blocking(False)
other.quantize_dynamic_ptr(pi[0], DataType.F32, po[0], DataType.UINT8, N, pr, RoundMode.NEAREST, _device_ptrs=True)
torch.cuda.synchronize()
out["after_a_second_stream_us"] = [timed() for _ in range(3)]
print(json.dumps(out))
