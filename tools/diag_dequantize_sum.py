#!/usr/bin/env python3
"""piquant_hip_dequantize_sum (acc += sum of K quantized inputs with device-resident parameters: the reduction step of the mesh all-reduce) at
numel 27 264 000 for K = 1, 3, 7, 15, uint8 -> fp32 and uint4 -> bf16, cold (>= 1.6 GB in rotation), HIP events over 60 stream-ordered calls, median of
7 windows.  One CSV row per case for whichever build PIQUANT_HIP_LIBRARY names (TAG labels it): interleaved A/B runs of two builds."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "pi-quant_amd"))
import torch  # noqa: E402

import piquant  # noqa: E402
from piquant import DataType, ReduceOp  # noqa: E402

N = 27_264_000
tag = os.environ.get("TAG", "production")
ctx = piquant.Context()
stream = torch.cuda.Stream()
ctx.set_stream(stream.cuda_stream)
ctx.set_blocking(False)
src = [torch.empty(N, device="cuda").uniform_(-1, 1) for _ in range(4)]
for qname, qdt, qbytes, fdt, tdt, fbytes in (("uint8", DataType.UINT8, N, DataType.F32, torch.float32, 4), ("uint4", DataType.UINT4, (N + 1) // 2, DataType.BF16, torch.bfloat16, 2)):
    for K in (1, 3, 7, 15):
        groups = max(2, int(1.6e9 // (K * qbytes + 2 * fbytes * N)) + 1)
        qs = [[torch.empty(qbytes, dtype=torch.uint8, device="cuda") for _ in range(K)] for _ in range(groups)]
        recs = [[torch.empty(16, dtype=torch.uint8, device="cuda") for _ in range(K)] for _ in range(groups)]
        with torch.cuda.stream(stream):
            for g in range(groups):
                for i in range(K):
                    x = src[(g + i) % 4] if fdt == DataType.F32 else src[(g + i) % 4].to(torch.bfloat16)
                    ctx.quantize_dynamic_ptr(x.data_ptr(), fdt, qs[g][i].data_ptr(), qdt, N, recs[g][i].data_ptr(), piquant.RoundMode.NEAREST, _device_ptrs=True)
            accs = [torch.zeros(N, dtype=tdt, device="cuda") for _ in range(groups)]
            pq = [[t.data_ptr() for t in grp] for grp in qs]
            pr = [[t.data_ptr() for t in grp] for grp in recs]
            call = lambda i: ctx.dequantize_sum_ptr(pq[i % groups], pr[i % groups], qdt, accs[i % groups].data_ptr(), fdt, N, ReduceOp.ADD, _device_ptrs=True)  # noqa: E731
            for i in range(20):
                call(i)
            ev = []
            for w in range(7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record(stream)
                for i in range(60):
                    call(w * 60 + i)
                e1.record(stream)
                torch.cuda.synchronize()
                ev.append(e0.elapsed_time(e1) / 60 * 1e3)
        ev.sort()
        bpe = K * qbytes / N + 2 * fbytes
        print(f"{tag},{qname}->{'f32' if fbytes == 4 else 'bf16'},{K},{ev[3]:.2f},{ev[0]:.2f},{bpe * N / (ev[3] * 1e-6) / 8e12:.4f}", flush=True)
        del qs, recs, accs
