#!/usr/bin/env python3
"""Workload for profiling compute_quant_params + quantize: N fused calls (one launch, tensor resident on chip) and N calls with
fusion off (scan, parameter kernel, quantize), fp32 -> uint8 at numel 27 264 000 on rotating buffers.  Run under rocprofv3:

  rocprofv3 --kernel-trace --stats -f csv -d out -o dyn -- python tools/dynamic_quantize_workload.py
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d out_f -o dyn -- python tools/dynamic_quantize_workload.py
"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "pi-quant_amd"))

import torch  # noqa: E402

import piquant  # noqa: E402

N, SETS, CALLS = 27_264_000, 6, 60
xs = [torch.empty(N, device="cuda").uniform_(-1, 1) for _ in range(SETS)]
outs = [torch.empty(N, dtype=torch.uint8, device="cuda") for _ in range(SETS)]
rec = torch.empty(16, dtype=torch.uint8, device="cuda")
ctx = piquant.Context()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for fusion in (True, False):
        ctx.set_fusion(fusion)
        for i in range(CALLS):
            piquant.torch.quantize_dynamic(xs[i % SETS], dtype=torch.uint8, ctx=ctx, out=outs[i % SETS], params=rec)
        torch.cuda.synchronize()
print("done")
