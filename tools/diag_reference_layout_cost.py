#!/usr/bin/env python3
"""What reference-layout mode costs at numel 27 264 000 (cold rotation, stream-ordered calls through the C ABI): the default (position-independent) mode
against reference layout for a reference context of 1, 3 and 255 pool threads -- fp32 -> uint8 nearest, bf16 -> uint4 nearest, uint8 -> fp32 ADD,
uint4 -> bf16 SET.  CSV on stdout."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "pi-quant_amd"))
import torch  # noqa: E402

import piquant  # noqa: E402
from piquant import DataType, ReduceOp, RoundMode  # noqa: E402

N, SETS, REPS = 27_264_000, 12, 120
dev = torch.device("cuda", 0)
x32 = [torch.empty(N, device=dev).uniform_(-1, 1) for _ in range(SETS)]
x16 = [t.to(torch.bfloat16) for t in x32]
q8 = [torch.empty(N, dtype=torch.uint8, device=dev) for _ in range(SETS)]
q4 = [torch.empty(N // 2, dtype=torch.uint8, device=dev) for _ in range(SETS)]
acc = [torch.zeros(N, device=dev) for _ in range(SETS)]
b16 = [torch.empty(N, dtype=torch.bfloat16, device=dev) for _ in range(SETS)]
stream = torch.cuda.Stream()
ctx = piquant.Context()
ctx.set_stream(stream.cuda_stream)
ctx.set_blocking(False)
OPS = {
    "quantize_f32_u8": (5, lambda i: ctx.quantize_ptr(x32[i].data_ptr(), DataType.F32, q8[i].data_ptr(), DataType.UINT8, N, 0.0078431377, 128, RoundMode.NEAREST)),
    "quantize_bf16_u4": (2.5, lambda i: ctx.quantize_ptr(x16[i].data_ptr(), DataType.BF16, q4[i].data_ptr(), DataType.UINT4, N, 0.13333334, 8, RoundMode.NEAREST)),
    "dequantize_u8_f32_add": (9, lambda i: ctx.dequantize_ptr(q8[i].data_ptr(), DataType.UINT8, acc[i].data_ptr(), DataType.F32, N, 0.0078431377, 128, ReduceOp.ADD)),
    "dequantize_u4_bf16_set": (2.5, lambda i: ctx.dequantize_ptr(q4[i].data_ptr(), DataType.UINT4, b16[i].data_ptr(), DataType.BF16, N, 0.13333334, 8, ReduceOp.SET)),
}
print("operator,mode,us_per_call,frac_of_8TBs")
with torch.cuda.stream(stream):
    for name, (bpe, fn) in OPS.items():
        for mode, threads in (("default", 0), ("reference_layout_1_thread", 1), ("reference_layout_3_threads", 3), ("reference_layout_255_threads", 255)):
            ctx.set_reference_layout(threads > 0, threads=max(threads, 1))
            for i in range(SETS):
                fn(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for i in range(REPS):
                fn(i % SETS)
            e1.record(stream)
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / REPS * 1e3
            print(f"{name},{mode},{us:.2f},{bpe * N / (us * 1e-6) / 8e12:.3f}", flush=True)
ctx.set_reference_layout(False, threads=1)
