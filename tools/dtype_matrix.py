#!/usr/bin/env python3
"""Every legal operator of the reference's dispatch tables (`kernels.inl:108-137`: 2 float types x 3 packed types x {nearest, stochastic}
quantize entries, 3 x 2 x {set, add} dequantize entries) at numel 27 264 000 through the C ABI, cold: 24 buffer sets in rotation (> 1.3 GB
for the smallest pair), `reps` calls captured once in a hipGraph and replayed, HIP events on the launch stream, best of 5 replays.  Per
pair: us per launch, algorithmic bytes per element, GB/s, fraction of the 8 TB/s HBM peak.  One JSON document.

  python tools/dtype_matrix.py > profiles/rNN_dtype_matrix.json
"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "pi-quant_amd"))
sys.path.insert(0, str(ROOT / "tools"))

import torch  # noqa: E402
from fit_fixed_cost import timed  # noqa: E402

import piquant  # noqa: E402
from piquant import DataType, ReduceOp, RoundMode  # noqa: E402

N, SETS, PEAK = 27_264_000, 24, 8.0e12
FLOATS = {"f32": (DataType.F32, 4), "bf16": (DataType.BF16, 2)}
PACKED = {"uint8": (DataType.UINT8, 8), "uint4": (DataType.UINT4, 4), "uint2": (DataType.UINT2, 2)}


def main():
    dev = torch.device("cuda")
    ctx = piquant.Context()
    stream = torch.cuda.Stream()
    ctx.set_stream(stream.cuda_stream)
    ctx.set_blocking(False)
    ctx.set_stochastic_threshold(0.37)
    rows = []
    with torch.cuda.stream(stream):
        xs32 = [torch.empty(N, device=dev).uniform_(-1, 1) for _ in range(SETS)]
        xs16 = [x.to(torch.bfloat16) for x in xs32]
        qs = [torch.empty(N, dtype=torch.uint8, device=dev) for _ in range(SETS)]
        torch.cuda.synchronize()
        px = {"f32": [t.data_ptr() for t in xs32], "bf16": [t.data_ptr() for t in xs16]}
        pq = [t.data_ptr() for t in qs]
        for qname, (qdt, bits) in PACKED.items():
            qmax = (1 << bits) - 1
            scale, zp = 2.0 / qmax, (qmax + 1) // 2
            for fname, (fdt, fbytes) in FLOATS.items():
                for mode in (RoundMode.NEAREST, RoundMode.STOCHASTIC):
                    bpe = fbytes + bits / 8
                    t = timed(lambda i: ctx.quantize_ptr(px[fname][i % SETS], fdt, pq[i % SETS], qdt, N, scale, zp, mode, _device_ptrs=True), 96, stream)
                    rows.append({"op": "quantize", "in": fname, "out": qname, "mode": mode.name.lower(), "bytes_per_elem": bpe, "us": round(t * 1e6, 2),
                                 "GB/s": round(bpe * N / t / 1e9, 1), "frac_of_peak": round(bpe * N / t / PEAK, 4)})
            # valid packed bytes for the dequantize side: the nearest result of set k
            for k in range(SETS):
                ctx.quantize_ptr(px["f32"][k], DataType.F32, pq[k], qdt, N, scale, zp, RoundMode.NEAREST, _device_ptrs=True)
            for fname, (fdt, fbytes) in FLOATS.items():
                for op in (ReduceOp.SET, ReduceOp.ADD):
                    bpe = bits / 8 + fbytes * (2 if op == ReduceOp.ADD else 1)
                    t = timed(lambda i: ctx.dequantize_ptr(pq[i % SETS], qdt, px[fname][i % SETS], fdt, N, scale, zp, op, _device_ptrs=True), 96, stream)
                    rows.append({"op": "dequantize", "in": qname, "out": fname, "mode": op.name.lower(), "bytes_per_elem": bpe, "us": round(t * 1e6, 2),
                                 "GB/s": round(bpe * N / t / 1e9, 1), "frac_of_peak": round(bpe * N / t / PEAK, 4)})
                    # the accumulators have drifted (ADD) or been overwritten (SET): restore the inputs of the next quantize rows
                    for k in range(SETS):
                        if fname == "f32":
                            xs32[k].uniform_(-1, 1)
                        else:
                            xs16[k].copy_(xs32[k])
            torch.cuda.synchronize()
    fr = [r["frac_of_peak"] for r in rows]
    print(json.dumps({"device": torch.cuda.get_device_name(0), "numel": N, "buffer_sets": SETS, "peak_GB/s": 8000,
                      "timing": "hipGraph of 96 calls through the C ABI replayed, HIP events on the launch stream, best of 5",
                      "min_frac": min(fr), "max_frac": max(fr), "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
