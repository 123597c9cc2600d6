#!/usr/bin/env python3
"""Where do the slow calls of a back-to-back loop of small piquant.torch.quantize calls come from?  One-off diagnosis: per-call host time stamps of
3000 calls for several variants of the same call; prints the mean by hundred and the calls that took more than 15 us."""
import gc
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "pi-quant_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402

import piquant  # noqa: E402
import piquant.torch  # noqa: E402
from piquant._bootstrap import C_LIB as C  # noqa: E402

NUMEL = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
x = torch.rand(NUMEL, device="cuda")
scale, zp = piquant.torch.compute_quant_params(x, dtype=torch.quint8) if NUMEL > 1 else (0.01, 3)
outq = torch.empty(x.shape, dtype=torch.quint8, device="cuda")
out8 = torch.empty(x.shape, dtype=torch.uint8, device="cuda")
ctx = piquant.Context.get(0)
pi, po, n = x.data_ptr(), out8.data_ptr(), x.numel()
gc.disable()
now = time.perf_counter_ns


def trace(f, calls=3000):
    f()
    torch.cuda.synchronize()
    ts = [now()]
    for _ in range(calls):
        f()
        ts.append(now())
    torch.cuda.synchronize()
    d = [(b - a) / 1e3 for a, b in zip(ts, ts[1:])]
    return {"by_hundred": [round(sum(d[i:i + 100]) / 100, 2) for i in range(0, calls, 100)],
            "calls_over_15us": [(i, round(v, 1)) for i, v in enumerate(d) if v > 15][:40],
            "median": round(sorted(d)[calls // 2], 2)}


variants = {
    "piquant.torch.quantize": lambda: piquant.torch.quantize(x, scale=scale, zero_point=zp, dtype=torch.quint8),
    "piquant.torch.quantize(out=)": lambda: piquant.torch.quantize(x, scale=scale, zero_point=zp, dtype=torch.quint8, out=outq),
    "piquant.torch.quantize(uniform=True)": lambda: piquant.torch.quantize(x, scale=scale, zero_point=zp, dtype=torch.quint8, uniform=True),
    "torch.quantize_per_tensor": lambda: torch.quantize_per_tensor(x, scale, zp, torch.quint8),
    "torch.empty(quint8)": lambda: torch.empty(x.shape, dtype=torch.quint8, device=x.device),
    "torch.add(x, 1)": lambda: torch.add(x, 1.0),
}
out = {}
for name, f in variants.items():
    out[name] = trace(f)
piquant.torch._ctx_for(x, None)
ctx.assume_device_pointers(True)
out["raw ctypes piquant_quantize"] = trace(lambda: C.piquant_quantize(ctx._ctx, pi, 0, po, 4, n, scale, zp, 0))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    out["piquant.torch.quantize on a side stream"] = trace(variants["piquant.torch.quantize"])
out["piquant.torch.quantize, again"] = trace(variants["piquant.torch.quantize"])
print(json.dumps(out))
