#!/usr/bin/env python3
"""Workload for counter passes over EVERY kernel the BASELINE configs use: 200 stream-ordered launches each (60 until round 6: the first launches of a process weighed 2-3 % in the averages), at
numel 27 264 000 on rotating buffer sets (> 256 MiB in total, so the Infinity Cache cannot serve them).

  configs[1]  quantize fp32 -> uint8 nearest
  configs[2]  quantize bf16 -> uint4 nearest; dequantize uint4 -> bf16 SET
  configs[3]  quantize fp32 -> uint8 stochastic; dequantize uint8 -> fp32 ADD
  configs[4]  min/max scan fp32 (N1 and, with --big, 2^28 elements) and bf16 (N1)
  f1          params + quantize in one launch (fused), fp32 -> uint8
  extras      dequantize uint8 -> fp32 SET

Run under rocprofv3 (tools/pmc_all_kernels.sh): one --stats pass, one pass per --pmc counter group.
"""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "pi-quant_amd"))

import torch  # noqa: E402

import piquant  # noqa: E402
from piquant import DataType, ReduceOp, RoundMode  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--numel", type=int, default=27_264_000)
ap.add_argument("--calls", type=int, default=200)
ap.add_argument("--big", action="store_true")
args = ap.parse_args()
N, SETS, CALLS = args.numel, 12, args.calls   # 12 sets: the 68 MB bf16/uint4 launches also rotate over > 3x the 256 MiB Infinity Cache

dev = torch.device("cuda")
ctx = piquant.Context()
s = torch.cuda.Stream()
ctx.set_stream(s.cuda_stream)
ctx.set_blocking(False)
ctx.set_stochastic_threshold(0.37)
xs = [torch.empty(N, device=dev).uniform_(-1, 1) for _ in range(SETS)]
xb = [x.to(torch.bfloat16) for x in xs]
q8 = [torch.empty(N, dtype=torch.uint8, device=dev) for _ in range(SETS)]
q4 = [torch.empty((N + 1) // 2, dtype=torch.uint8, device=dev) for _ in range(SETS)]
acc = [torch.zeros(N, device=dev) for _ in range(SETS)]
yb = [torch.empty(N, dtype=torch.bfloat16, device=dev) for _ in range(SETS)]
rec = torch.empty(16, dtype=torch.uint8, device=dev)
keys = torch.empty(2, dtype=torch.int32, device=dev)
scale, zp = piquant.torch.compute_quant_params(xs[0], dtype=torch.quint8)
s4, z4 = piquant.torch.compute_quant_params(xb[0], dtype=torch.quint4x2)
ctx.set_stream(s.cuda_stream)
ctx.set_blocking(False)
P = lambda ts: [t.data_ptr() for t in ts]   # noqa: E731
pxs, pxb, pq8, pq4, pacc, pyb = P(xs), P(xb), P(q8), P(q4), P(acc), P(yb)
with torch.cuda.stream(s):
    for i in range(CALLS):
        k = i % SETS
        ctx.quantize_ptr(pxs[k], DataType.F32, pq8[k], DataType.UINT8, N, scale, zp, RoundMode.NEAREST, _device_ptrs=True)
    for i in range(CALLS):
        k = i % SETS
        ctx.quantize_ptr(pxb[k], DataType.BF16, pq4[k], DataType.UINT4, N, s4, z4, RoundMode.NEAREST, _device_ptrs=True)
    for i in range(CALLS):
        k = i % SETS
        ctx.dequantize_ptr(pq4[k], DataType.UINT4, pyb[k], DataType.BF16, N, s4, z4, ReduceOp.SET, _device_ptrs=True)
    for i in range(CALLS):
        k = i % SETS
        ctx.quantize_ptr(pxs[k], DataType.F32, pq8[k], DataType.UINT8, N, scale, zp, RoundMode.STOCHASTIC, _device_ptrs=True)
    for i in range(CALLS):
        k = i % SETS
        ctx.dequantize_ptr(pq8[k], DataType.UINT8, pacc[k], DataType.F32, N, scale, zp, ReduceOp.ADD, _device_ptrs=True)
    for i in range(CALLS):
        k = i % SETS
        ctx.dequantize_ptr(pq8[k], DataType.UINT8, pacc[k], DataType.F32, N, scale, zp, ReduceOp.SET, _device_ptrs=True)
    for i in range(CALLS):
        ctx.minmax_keys_ptr(pxs[i % SETS], DataType.F32, N, keys.data_ptr(), True, _device_ptrs=True)
    for i in range(CALLS):      # the bf16 scan in steady state (round-4 verdict: the counter file held ONE cold launch of it)
        ctx.minmax_keys_ptr(pxb[i % SETS], DataType.BF16, N, keys.data_ptr(), True, _device_ptrs=True)
    for i in range(CALLS):
        k = i % SETS
        ctx.quantize_dynamic_ptr(pxs[k], DataType.F32, pq8[k], DataType.UINT8, N, rec.data_ptr(), RoundMode.NEAREST, _device_ptrs=True)
    torch.cuda.synchronize()
    if args.big:
        del xb, q4, acc, yb
        big = torch.empty(1 << 28, device=dev).uniform_(-1, 1)
        for i in range(20):
            ctx.minmax_keys_ptr(big.data_ptr(), DataType.F32, big.numel(), keys.data_ptr(), True, _device_ptrs=True)
        torch.cuda.synchronize()
print("done")
