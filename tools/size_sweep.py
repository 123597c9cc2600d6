#!/usr/bin/env python3
"""Kernel time vs tensor size through the C ABI (HIP events on the launch stream, rotating buffers beyond the Infinity Cache where
the size allows): quantize fp32->uint8, dequantize uint8->fp32 (SET / ADD), min/max scan, fused params+quantize.  One JSON
document: microseconds per call and algorithmic TB/s for numel from 10^5 to 2^30.

  python tools/size_sweep.py > profiles/rNN_size_sweep.json
"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "pi-quant_amd"))

import torch  # noqa: E402

import piquant  # noqa: E402
from piquant import DataType, ReduceOp, RoundMode  # noqa: E402


def timed(fn, reps, stream):
    for i in range(min(reps, 10)):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for i in range(reps):
        fn(i)
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    dev = torch.device("cuda")
    ctx = piquant.Context()
    stream = torch.cuda.Stream()
    ctx.set_stream(stream.cuda_stream)
    ctx.set_blocking(False)
    rows = []
    for n in (100_000, 1_000_000, 4_000_000, 10_000_000, 27_264_000, 100_000_000, 268_435_456, 1_073_741_824):
        sets = max(2, min(6, int(1.2e9 // (5 * n)) or 2)) if n <= 268_435_456 else 1
        xs = [torch.empty(n, device=dev).uniform_(-1, 1) for _ in range(sets)]
        qs = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(sets)]
        rec = torch.empty(16, dtype=torch.uint8, device=dev)
        keys = torch.empty(2, dtype=torch.int32, device=dev)
        reps = max(5, min(400, int(3e9 // (5 * n))))
        px, pq = [t.data_ptr() for t in xs], [t.data_ptr() for t in qs]
        with torch.cuda.stream(stream):
            row = {"numel": n, "buffer_sets": sets, "reps": reps}

            def put(name, us, bytes_per_elem):
                row[name] = {"us": round(us, 2), "TB/s": round(bytes_per_elem * n / us / 1e6, 3)}

            put("quantize_f32_u8", timed(lambda i: ctx.quantize_ptr(px[i % sets], DataType.F32, pq[i % sets], DataType.UINT8, n, 0.0078431377, 128,
                                                                    RoundMode.NEAREST, _device_ptrs=True), reps, stream), 5)
            put("dequantize_u8_f32_set", timed(lambda i: ctx.dequantize_ptr(pq[i % sets], DataType.UINT8, px[i % sets], DataType.F32, n, 0.0078431377, 128,
                                                                            ReduceOp.SET, _device_ptrs=True), reps, stream), 5)
            put("dequantize_u8_f32_add", timed(lambda i: ctx.dequantize_ptr(pq[i % sets], DataType.UINT8, px[i % sets], DataType.F32, n, 0.0078431377, 128,
                                                                            ReduceOp.ADD, _device_ptrs=True), reps, stream), 9)
            for t in xs:
                t.uniform_(-1, 1)
            put("minmax_f32", timed(lambda i: ctx.minmax_keys_ptr(px[i % sets], DataType.F32, n, keys.data_ptr(), True, _device_ptrs=True), reps, stream), 4)
            fused_us = timed(lambda i: ctx.quantize_dynamic_ptr(px[i % sets], DataType.F32, pq[i % sets], DataType.UINT8, n, rec.data_ptr(), RoundMode.NEAREST,
                                                                _device_ptrs=True), reps, stream)
            ctx.set_fusion(False)
            unfused_us = timed(lambda i: ctx.quantize_dynamic_ptr(px[i % sets], DataType.F32, pq[i % sets], DataType.UINT8, n, rec.data_ptr(), RoundMode.NEAREST,
                                                                  _device_ptrs=True), reps, stream)
            ctx.set_fusion(True)
            row["params_plus_quantize"] = {"us": round(fused_us, 2), "us_two_launches": round(unfused_us, 2)}
        rows.append(row)
        del xs, qs
        torch.cuda.empty_cache()
    print(json.dumps({"device": torch.cuda.get_device_name(0), "peak_TB/s": 8.0, "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
