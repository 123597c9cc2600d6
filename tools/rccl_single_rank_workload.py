#!/usr/bin/env python3
"""Workload for a kernel trace showing RCCL's kernels next to the fused params+quantize kernel: a one-rank `nccl`
(= RCCL) group runs the transport branches of piquant.distributed -- send/recv to self, all_to_all_single, all_gather_into_tensor --
and both quantized all-reduce schedules end to end, while a second stream keeps launching fused kernels.

  rocprofv3 --kernel-trace --stats -f csv -d out -o rccl -- python tools/rccl_single_rank_workload.py
"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "pi-quant_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import piquant  # noqa: E402
import piquant.distributed as D  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29655")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
N = 27_264_000
x = torch.empty(N, device="cuda").uniform_(-1, 1)
y = torch.empty(N, device="cuda").uniform_(-1, 1)
side, side_ctx = torch.cuda.Stream(), piquant.Context()
buf = torch.randint(0, 256, (N // 8 + 16,), dtype=torch.uint8, device="cuda")
out = torch.empty_like(buf)
for it in range(10):
    with torch.cuda.stream(side):
        for _ in range(8):
            piquant.torch.quantize_dynamic(y, dtype=torch.uint8, ctx=side_ctx)
    D._exchange(buf, out, 0, 0, None)
    D._all_to_all(buf, out, None)
    D._all_gather(buf, out, None)
    for algorithm in ("ring", "direct"):
        t = x.clone()
        D.quantized_all_reduce(t, quant_dtype=torch.uint8, algorithm=algorithm, _single_rank_collectives=True)
    s, z = D.compute_quant_params(x, dtype=torch.quint8)
torch.cuda.synchronize()
print("params", s, z, "bailouts on the side stream", side_ctx.barrier_bailouts())
dist.destroy_process_group()
