#!/usr/bin/env python3
"""The min/max scan (piquant_hip_minmax_keys, EP_KEYS_SET) at numel 27 264 000, fp32 and bf16, cold (24 sets), windows of 20 stream-ordered calls,
HIP events: one CSV row per dtype for whichever build of the library PIQUANT_HIP_LIBRARY names.  For interleaved A/B runs of two builds."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "pi-quant_amd"))
import torch  # noqa: E402

import piquant  # noqa: E402
from piquant import DataType  # noqa: E402

N, SETS, K, W = 27_264_000, 24, 20, 31
xs = [torch.empty(N, device="cuda").uniform_(-1, 1) for _ in range(SETS)]
xb = [x.to(torch.bfloat16) for x in xs]
keys = torch.empty(2, dtype=torch.int32, device="cuda")
ctx = piquant.Context()
stream = torch.cuda.Stream()
ctx.set_stream(stream.cuda_stream)
ctx.set_blocking(False)
tag = os.environ.get("TAG", "production")
for name, bufs, dt, bpe in (("f32", xs, DataType.F32, 4), ("bf16", xb, DataType.BF16, 2)):
    ev = []
    with torch.cuda.stream(stream):
        for i in range(200):
            ctx.minmax_keys_ptr(bufs[i % SETS].data_ptr(), dt, N, keys.data_ptr(), True, _device_ptrs=True)
        for w in range(W):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record(stream)
            for i in range(w * K, (w + 1) * K):
                ctx.minmax_keys_ptr(bufs[i % SETS].data_ptr(), dt, N, keys.data_ptr(), True, _device_ptrs=True)
            e1.record(stream)
            torch.cuda.synchronize()
            ev.append(e0.elapsed_time(e1) / K * 1e3)
    ev = sorted(ev[1:])
    med = ev[len(ev) // 2]
    print(f"{tag},{name},{med:.3f},{ev[0]:.3f},{bpe * N / (med * 1e-6) / 8e12:.4f}")
