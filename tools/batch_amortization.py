#!/usr/bin/env python3
"""What the batch entry points buy for BASELINE config 3 (68 MB per tensor, where one launch's fixed cost keeps a single call at 64-73 % of the
HBM peak): dequantize_dp_batch (uint4 -> bf16, device records) and quantize_dynamic_batch (bf16 -> uint4, parameters + quantize) over 1, 2,
4 and 8 tensors of numel 27 264 000 per launch, cold rotation (>= 1.6 GB).  One JSON document.

  python tools/batch_amortization.py > profiles/rNN_batch_amortization.json
"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "pi-quant_amd"))
import torch  # noqa: E402

import piquant  # noqa: E402
import piquant.torch as pt  # noqa: E402

N, SETS = 27_264_000, 24
dev = torch.device("cuda")
ctx = piquant.Context()
s = torch.cuda.Stream()
xb = [torch.empty(N, device=dev).uniform_(-1, 1).to(torch.bfloat16) for _ in range(SETS)]
q4 = [torch.empty((N + 1) // 2, dtype=torch.uint8, device=dev) for _ in range(SETS)]
recs = [torch.empty(16, dtype=torch.uint8, device=dev) for _ in range(SETS)]
with torch.cuda.stream(s):
    for k in range(SETS):
        pt.quantize_dynamic(xb[k], dtype=torch.quint4x2, ctx=ctx, out=q4[k], params=recs[k])
torch.cuda.synchronize()


def timed(fn, reps=60):
    with torch.cuda.stream(s):
        for i in range(6):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for i in range(reps):
            fn(i)
        e1.record(s)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


out = {"device": torch.cuda.get_device_name(0), "numel_per_tensor": N, "peak_GB/s": 8000, "rows": []}
for count in (1, 2, 4, 8):
    idx = lambda i: [(i * count + j) % SETS for j in range(count)]   # noqa: E731
    us = timed(lambda i: pt.dequantize_dynamic_batch([q4[k] for k in idx(i)], [recs[k] for k in idx(i)], dtype=torch.bfloat16, ctx=ctx, outs=[xb[k] for k in idx(i)],
                                                     quant_dtype=torch.quint4x2, shapes=[(N,)] * count))
    row = {"tensors_per_launch": count, "dequantize_dp_batch_u4_bf16": {"us_per_launch": round(us, 2), "us_per_tensor": round(us / count, 2),
                                                                           "frac_of_peak": round(2.5 * N * count / us / 1e3 / 8000, 4)}}
    us = timed(lambda i: pt.quantize_dynamic_batch([xb[k] for k in idx(i)], dtype=torch.quint4x2, ctx=ctx, outs=[q4[k] for k in idx(i)], params=[recs[k] for k in idx(i)]))
    row["quantize_dynamic_batch_bf16_u4 (parameters + quantize, x read once)"] = {"us_per_launch": round(us, 2), "us_per_tensor": round(us / count, 2),
                                                                                  "frac_of_peak": round(2.5 * N * count / us / 1e3 / 8000, 4)}
    out["rows"].append(row)
# the single-tensor calls beside them
us = timed(lambda i: pt.dequantize_dynamic(q4[i % SETS], recs[i % SETS], dtype=torch.bfloat16, ctx=ctx, out=xb[i % SETS], quant_dtype=torch.quint4x2, shape=(N,)))
out["single dequantize_dp u4->bf16"] = {"us": round(us, 2), "frac_of_peak": round(2.5 * N / us / 1e3 / 8000, 4)}
print(json.dumps(out, indent=1))
