#!/usr/bin/env python3
"""compute_quant_params + quantize (piquant.torch.quantize_dynamic) fused vs unfused for every dtype pair, numel 27 264 000 and
(bf16) 54 528 000: microseconds per call from CUDA events over rotating buffers.  python tools/dynamic_quantize_matrix.py"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "pi-quant_amd"))

import torch  # noqa: E402

import piquant  # noqa: E402

SETS, REPS = 6, 200
ctx = piquant.Context()
rows = []
for fdt, n in ((torch.float32, 27_264_000), (torch.bfloat16, 27_264_000), (torch.bfloat16, 54_528_000)):
    xs = [torch.empty(n, device="cuda").uniform_(-1, 1).to(fdt) for _ in range(SETS)]
    for qdt, bits in ((torch.quint8, 8), (torch.quint4x2, 4), (torch.quint2x4, 2)):
        outs = [torch.empty(n, dtype=qdt, device="cuda") for _ in range(SETS)]
        rec = torch.empty(16, dtype=torch.uint8, device="cuda")
        row = {"input": str(fdt).replace("torch.", ""), "numel": n, "output": str(qdt).replace("torch.", "")}
        for fusion in (True, False):
            ctx.set_fusion(fusion)
            for i in range(20):
                piquant.torch.quantize_dynamic(xs[i % SETS], dtype=qdt, ctx=ctx, out=outs[i % SETS], params=rec)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(REPS):
                piquant.torch.quantize_dynamic(xs[i % SETS], dtype=qdt, ctx=ctx, out=outs[i % SETS], params=rec)
            e1.record()
            torch.cuda.synchronize()
            row["fused_us" if fusion else "unfused_us"] = round(e0.elapsed_time(e1) * 1e3 / REPS, 2)
        in_b = n * (4 if fdt == torch.float32 else 2)
        row["fits_on_chip"] = in_b <= 27 * 1024 * 16 * torch.cuda.get_device_properties(0).multi_processor_count
        row["speedup"] = round(row["unfused_us"] / row["fused_us"], 3)
        rows.append(row)
        del outs
    del xs
print(json.dumps(rows, indent=1))
