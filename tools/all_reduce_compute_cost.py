#!/usr/bin/env python3
"""Per-rank KERNEL time of a quantized all-reduce (no wire): the sequence of encode / decode / dequantize_sum calls one rank
issues for a `world`-way all-reduce of an fp32 tensor, replayed on one GPU with stand-in receive buffers.  Shows what the
schedules cost in HBM time next to the xGMI transfer time they overlap with (per-link ~153 GB/s, guides/MI355X_MICROARCH.md).

  python tools/all_reduce_compute_cost.py [--numel 27264000] [--world 8]
"""
import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "pi-quant_amd"))

import torch  # noqa: E402

import piquant  # noqa: E402
import piquant.distributed as D  # noqa: E402

HEADER = 16


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--numel", type=int, default=27_264_000)
    ap.add_argument("--world", type=int, default=8)
    args = ap.parse_args()
    W, n = args.world, args.numel
    dev = torch.device("cuda")
    ops = D._DeviceOps(piquant.Context())
    x = torch.empty(n, device=dev).uniform_(-1, 1)
    out = {"numel": n, "world": W}
    for qname, bits in (("uint8", 8), ("quint4x2", 4)):
        qdt = getattr(torch, qname)
        chunks = D.ring_chunks(n, W, bits)
        nbytes = [HEADER + piquant.DataType.UINT8.packed_nbytes(e - b) if bits == 8 else HEADER + piquant.DataType.UINT4.packed_nbytes(e - b) for b, e in chunks]
        slot = -(-max(nbytes) // 16) * 16
        bufs = torch.zeros(W * slot, dtype=torch.uint8, device=dev)
        mine = torch.zeros(slot, dtype=torch.uint8, device=dev)
        for j, (b, e) in enumerate(chunks):          # valid wire content in every slot
            ops.encode(x[b:e], bufs[j * slot: j * slot + nbytes[j]], qdt, "nearest")

        def ring():
            ops.encode(x[chunks[0][0]:chunks[0][1]], bufs[0: nbytes[0]], qdt, "nearest")
            for step in range(W - 1):                # reduce-scatter hops: add what arrives, forward the re-quantized sum (one launch)
                r = (0 - step - 1) % W
                ops.reduce_encode([bufs[r * slot: r * slot + nbytes[r]]], x[chunks[r][0]:chunks[r][1]], mine[: nbytes[r]], qdt, "nearest")
            for j in range(W):                       # all-gather: every chunk decoded once
                ops.decode(bufs[j * slot: j * slot + nbytes[j]], x[chunks[j][0]:chunks[j][1]], qdt, "set")

        def direct():
            peers = list(range(1, W))                # one launch quantizes the chunk of every peer
            ops.encode_batch([x[chunks[j][0]:chunks[j][1]] for j in peers], [bufs[j * slot: j * slot + nbytes[j]] for j in peers], qdt, "nearest")
            ops.reduce_encode([bufs[i * slot: i * slot + nbytes[0]] for i in range(1, W)], x[chunks[0][0]:chunks[0][1]], mine[: nbytes[0]], qdt, "nearest")
            ops.decode_batch([bufs[j * slot: j * slot + nbytes[j]] for j in range(W)], [x[chunks[j][0]:chunks[j][1]] for j in range(W)], qdt, "set")

        row = {}
        for name, fn in (("ring", ring), ("direct", direct)):
            x.uniform_(-1, 1)
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                fn()
                torch.cuda.synchronize()
                with torch.cuda.graph(g, stream=s):
                    fn()
                g.replay()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    g.replay()
                e1.record()
                torch.cuda.synchronize()
            row[name + "_kernel_us_per_all_reduce"] = round(e0.elapsed_time(e1) * 1e3 / 20, 1)
            x.uniform_(-1, 1)                        # values drift under repeated accumulation; irrelevant for timing
        chunk_wire = max(nbytes)
        row["wire_bytes_per_rank"] = 2 * (W - 1) * chunk_wire
        row["xgmi_us_ring_one_link_153GBps"] = round(2 * (W - 1) * chunk_wire / 153e9 * 1e6, 1)
        row["xgmi_us_mesh_all_links"] = round(2 * chunk_wire / 153e9 * 1e6, 1)
        out[qname] = row
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
