#!/usr/bin/env python3
"""Where a small call's time goes (reference python/benchmark/benchmark.py: NUMEL = 1e6, 1000 runs, torch vs piquant).

For every variant: wall-clock per call over a back-to-back loop (what benchmark.py measures), the device time per call over the same loop (HIP events
around it: if it equals the wall clock the loop is bound by the GPU -- kernel + dispatch gap -- not by the host), and the host-only cost per call
(the same loop with the stream kept busy by nothing: calls issued while the device is idle cannot be told apart, so the host cost is taken from a
loop over a tensor of ONE element, whose kernel is all fixed cost).  Prints one JSON object."""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "pi-quant_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

import piquant  # noqa: E402
from piquant import DataType, RoundMode  # noqa: E402
from piquant._bootstrap import C_LIB as C  # noqa: E402


def timed(f, n=20000):
    for _ in range(200):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    return {"wall_us": round(t_all / n * 1e6, 3), "issue_us": round(t_issue / n * 1e6, 3), "device_us": round(e0.elapsed_time(e1) / n * 1e3, 3)}


def main():
    out = {"device": torch.cuda.get_device_name(0), "native_front_end": piquant.torch._native is not None}
    for numel in (1_000_000, 1, 100_000, 10_000_000):
        x = torch.rand(numel, device="cuda")
        scale, zp = piquant.torch.compute_quant_params(x, dtype=torch.quint8) if numel > 1 else (0.01, 3)
        out8 = torch.empty(x.shape, dtype=torch.uint8, device="cuda")
        outq = torch.empty(x.shape, dtype=torch.quint8, device="cuda")
        ctx = piquant.Context.get(0)
        pi, po, n = x.data_ptr(), out8.data_ptr(), x.numel()
        rec = {}
        rec["piquant.torch.quantize(quint8)"] = timed(lambda: piquant.torch.quantize(x, scale=scale, zero_point=zp, dtype=torch.quint8))
        rec["piquant.torch.quantize(out=)"] = timed(lambda: piquant.torch.quantize(x, scale=scale, zero_point=zp, dtype=torch.quint8, out=outq))
        rec["piquant.torch.quantize(uniform)"] = timed(lambda: piquant.torch.quantize(x, scale=scale, zero_point=zp, dtype=torch.quint8, uniform=True))
        piquant.torch._ctx_for(x, None)
        ctx.assume_device_pointers(True)
        rec["raw ctypes piquant_quantize"] = timed(lambda: C.piquant_quantize(ctx._ctx, pi, 0, po, 4, n, scale, zp, 0))
        rec["raw ctypes piquant_hip_quantize_uniform"] = timed(lambda: C.piquant_hip_quantize_uniform(ctx._ctx, pi, 0, po, 4, n, scale, zp, 0))
        rec["torch.quantize_per_tensor"] = timed(lambda: torch.quantize_per_tensor(x, scale, zp, torch.quint8))
        rec["torch.empty(quint8)"] = timed(lambda: torch.empty(x.shape, dtype=torch.quint8, device=x.device))
        q = piquant.torch.quantize(x, scale=scale, zero_point=zp, dtype=torch.quint8)
        tq = torch.quantize_per_tensor(x, scale, zp, torch.quint8)
        rec["piquant.torch.dequantize"] = timed(lambda: piquant.torch.dequantize(q, scale=scale, zero_point=zp, dtype=torch.float32))
        rec["torch.dequantize"] = timed(lambda: torch.dequantize(tq))
        # pure device time of the kernels: the same launches replayed from a hipGraph (no host in the loop)
        for label, uniform in (("piquant kernel, graph of 50", False), ("piquant kernel (uniform), graph of 50", True)):
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                piquant.torch.quantize(x, scale=scale, zero_point=zp, dtype=torch.quint8, out=outq, uniform=uniform)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    for _ in range(50):
                        piquant.torch.quantize(x, scale=scale, zero_point=zp, dtype=torch.quint8, out=outq, uniform=uniform)
                r = timed(g.replay, n=200)
                rec[label] = {k: round(v / 50, 3) for k, v in r.items()}
        out[f"numel={numel}"] = rec
    print(json.dumps(out))


if __name__ == "__main__":
    main()
