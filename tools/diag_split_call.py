#!/usr/bin/env python3
"""A/B of ONE piquant_quantize call issued as one launch (production) and as one launch without the barrier bit of its dispatch packet
(piquant_hip_set_independent_calls: the caller declares consecutive calls independent).  The third variant behind profiles/r05_split_call_ab.csv --
the call as two launches on two streams between a fork and a join event, round-4 verdict item 2 -- lost (41 against 23.5 us per call) and its code
left the library again; commit 5514ca7 has it (PIQUANT_HIP_SPLIT_CALL=1, csrc/capi.cpp).  fp32 -> uint8 nearest at numel 27 264 000, 24 cold buffer sets, windows of
K stream-ordered calls through the C ABI, the two variants interleaved window by window; wall clock (first call -> completion) and HIP events.
CSV on stdout."""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "pi-quant_amd"))
import torch  # noqa: E402

import piquant  # noqa: E402
from piquant import DataType, RoundMode  # noqa: E402
from piquant._bootstrap import C_LIB  # noqa: E402

N, K, WINDOWS = int(os.environ.get("N", 27_264_000)), 20, 41
SETS = max(4, min(24, int(3.3e9 // (5 * N))))
dev = torch.device("cuda", 0)
xs = [torch.empty(N, device=dev).uniform_(-1, 1) for _ in range(SETS)]
outs = [torch.empty(N, dtype=torch.uint8, device=dev) for _ in range(SETS)]
scale, zp = piquant.torch.compute_quant_params(xs[0], dtype=torch.quint8)
stream = torch.cuda.Stream()


def make(independent):
    c = piquant.Context()
    c.set_independent_calls(independent)
    c.set_stream(stream.cuda_stream)
    c.set_blocking(False)
    c.assume_device_pointers(True)
    return c


ctxs = {"one_launch": make(False), "one_launch_without_barrier_bit_caller_declares_independence": make(True)}
args = {k: [(c._ctx, xs[i].data_ptr(), DataType.F32.value, outs[i].data_ptr(), DataType.UINT8.value, N, scale, zp, RoundMode.NEAREST.value) for i in range(SETS)]
        for k, c in ctxs.items()}
# same bytes
ref = None
for k in ctxs:
    outs[0].zero_()
    C_LIB.piquant_quantize(*args[k][0])
    torch.cuda.synchronize()
    got = outs[0].clone()
    assert ref is None or torch.equal(ref, got), "the any-order launch changed the bytes"
    ref = got
res = {k: {"wall": [], "ev": []} for k in ctxs}
with torch.cuda.stream(stream):
    for k in ctxs:
        for i in range(400):
            C_LIB.piquant_quantize(*args[k][i % SETS])
    torch.cuda.synchronize()
    base = 0
    for w in range(WINDOWS):
        for k in ctxs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e0.record(stream)
            for i in range(base, base + K):
                C_LIB.piquant_quantize(*args[k][i % SETS])
            e1.record(stream)
            while not e1.query():
                pass
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            res[k]["wall"].append((t1 - t0) / K * 1e6)
            res[k]["ev"].append(e0.elapsed_time(e1) / K * 1e3)
            base += K
# blocking calls (the reference's semantics): one at a time
blk = {}
for k, c in ctxs.items():
    c.set_blocking(True)
    c.assume_device_pointers(True)
    for i in range(40):
        C_LIB.piquant_quantize(*args[k][i % SETS])
    t0 = time.perf_counter()
    for i in range(300):
        C_LIB.piquant_quantize(*args[k][i % SETS])
    blk[k] = (time.perf_counter() - t0) / 300 * 1e6
    c.set_blocking(False)
print("variant,numel,buffer_sets,windows,calls_per_window,us_per_call_wall_median,us_per_call_wall_min,us_per_call_events_median,frac_of_8TBs_events,us_per_blocking_call")
for k in ctxs:
    wall, ev = sorted(res[k]["wall"][1:]), sorted(res[k]["ev"][1:])
    m = len(wall) // 2
    print(f"{k},{N},{SETS},{len(wall)},{K},{wall[m]:.3f},{wall[0]:.3f},{ev[m]:.3f},{5 * N / (ev[m] * 1e-6) / 8e12:.4f},{blk[k]:.3f}")
