#!/usr/bin/env python3
"""What an UNPREPARED caller of the host-memory companion gets (tools; run on the GPU box's host): one fp32 tensor of numel 27 264 000 allocated and
filled by the main thread (all its pages on the main thread's NUMA node) per call, twelve of them in rotation (1.6 GB: DRAM, not L3),
piquant.cpu fp32 -> uint8 with T active workers, unpinned and pinned (bench.py's order: physical cores first, socket by socket); best mean
per call over whole rotations.  Prints GiB/s by thread count and the library's default worker count."""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "pi-quant_amd"))
import bench  # noqa: E402  (host_cpu_order)
from piquant import cpu as pcpu  # noqa: E402

n = 27_264_000
SETS = 12                                   # 1.6 GB in rotation: beyond the two sockets' 512 MB of L3
rng = np.random.default_rng(0)
xs = [rng.uniform(-1, 1, n).astype(np.float32) for _ in range(SETS)]
outs = [np.zeros(n, dtype=np.uint8) for _ in range(SETS)]          # touched here as well: outputs the caller has used before
order, per_socket, cores, sockets = bench.host_cpu_order()
res = {"numel": n, "sockets": sockets, "cores": cores, "buffer_sets": SETS, "rows": []}
for pinned in (False, True):
    ctx = pcpu.CpuContext(len(os.sched_getaffinity(0)))
    if pinned:
        ctx.set_affinity(order[: ctx.num_threads])
    for t in (8, 16, 32, 64, 96, 128, ctx.num_threads):
        if t > ctx.num_threads:
            continue
        ctx.set_active_threads(t)
        best = 1e9
        for rot in range(4):
            t0 = time.perf_counter()
            for k in range(SETS):
                ctx.quantize_ptr(xs[k].ctypes.data, 0, outs[k].ctypes.data, 4, n, 2.0 / 255.0, 128)
            if rot:
                best = min(best, (time.perf_counter() - t0) / SETS)
        res["rows"].append({"pinned": pinned, "threads": t, "ms": round(best * 1e3, 4), "GiB/s": round(n * 4 / 2**30 / best, 1)})
    ctx.close()
d = pcpu.CpuContext(0)
res["default_workers"] = d.num_threads
d.close()
print(json.dumps(res, indent=1))
