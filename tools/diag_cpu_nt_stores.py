#!/usr/bin/env python3
"""A/B of the companion library's store policy on this box's host cores: fp32 -> uint8 nearest at numel 27 264 000 (and uint8 -> fp32 SET),
cached stores against non-temporal stores (PIQUANT_CPU_NT_STORES=0 / 1, read once per process: one subprocess per arm, arms interleaved).
Prints one JSON object.  No GPU needed."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
ARM = r'''
import sys, time, json
import numpy as np
sys.path.insert(0, %r)
from piquant import cpu
n = 27_264_000
ctx = cpu.CpuContext(0)
rng = np.random.default_rng(0)
xs = [rng.uniform(-1, 1, n).astype(np.float32) for _ in range(8)]
qs = [np.zeros(n, dtype=np.uint8) for _ in range(8)]
res = {}
best = 1e9
for rot in range(6):
    t0 = time.perf_counter()
    for x, q in zip(xs, qs):
        ctx.quantize_ptr(x.ctypes.data, 0, q.ctypes.data, 4, n, 2 / 255, 127)
    if rot:
        best = min(best, (time.perf_counter() - t0) / 8)
res["quantize_f32_u8_ms"] = round(best * 1e3, 4)
best = 1e9
for rot in range(6):
    t0 = time.perf_counter()
    for x, q in zip(xs, qs):
        ctx.dequantize_ptr(q.ctypes.data, 4, x.ctypes.data, 0, n, 2 / 255, 127, 0)
    if rot:
        best = min(best, (time.perf_counter() - t0) / 8)
res["dequantize_u8_f32_set_ms"] = round(best * 1e3, 4)
res["threads"] = ctx.num_threads() if callable(ctx.num_threads) else ctx.num_threads
print(json.dumps(res))
''' % str(ROOT / "pi-quant_amd")

out = {"0_cached": [], "1_non_temporal": []}
for rep in range(3):
    for key, val in (("0_cached", "0"), ("1_non_temporal", "1")):
        r = subprocess.run([sys.executable, "-c", ARM], capture_output=True, text=True, env=dict(os.environ, PIQUANT_CPU_NT_STORES=val), timeout=600)
        out[key].append(json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": r.stderr[-500:]})
print(json.dumps(out))
