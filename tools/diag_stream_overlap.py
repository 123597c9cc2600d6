#!/usr/bin/env python3
"""Diagnostic: does a CU-holding kernel on one stream run NEXT TO a fused launch on another stream?  (tests/test_gpu_barrier.py)"""
import ctypes
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "pi-quant_amd"))
import torch  # noqa: E402

import piquant  # noqa: E402

so = Path("/tmp/libcuhog.so")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", str(ROOT / "tests" / "cu_hog.hip"), "-o", str(so)], check=True)
lib = ctypes.CDLL(str(so))
lib.cu_hog_launch.restype = ctypes.c_int
lib.cu_hog_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_void_p]
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
x = torch.empty(27_264_000, device="cuda").uniform_(-1, 1)
ctx = piquant.Context()
ctx.set_barrier_timeout_us(200)
piquant.torch.quantize_dynamic(x, dtype=torch.uint8, ctx=ctx)
torch.cuda.synchronize()
for label, side, work in (("hi-prio side, default work", torch.cuda.Stream(priority=-1), None),
                          ("side, explicit work stream", torch.cuda.Stream(), torch.cuda.Stream()),
                          ("hi-prio side, explicit work stream", torch.cuda.Stream(priority=-1), torch.cuda.Stream())):
    for blocks in (96, 200):
        torch.cuda.synchronize()
        b0 = ctx.barrier_bailouts()
        t0 = time.perf_counter()
        lib.cu_hog_launch(ctypes.c_void_p(side.cuda_stream), blocks, 20_000, ctypes.c_void_p(sink.data_ptr()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ws = work or torch.cuda.current_stream()
        with torch.cuda.stream(ws):
            e0.record()
            piquant.torch.quantize_dynamic(x, dtype=torch.uint8, ctx=ctx)
            e1.record()
        ws.synchronize()
        t_work = time.perf_counter() - t0
        side.synchronize()
        t_all = time.perf_counter() - t0
        print(f"{label:38s} hog blocks {blocks:3d}: fused done after {t_work * 1e3:7.2f} ms (events {e0.elapsed_time(e1):7.3f} ms), hog done after {t_all * 1e3:7.2f} ms, "
              f"bailouts +{ctx.barrier_bailouts() - b0}")
