#!/usr/bin/env python3
"""The quantized mesh all-reduce over peer-mapped buffers (piquant.distributed, transport='p2p') next to the collective transport and
the backend's fp32 all-reduce, on a process group of its own -- one JSON line on rank 0.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/p2p_all_reduce_bench.py

bench.py runs this as a CHILD job from its rank 0 while its own ranks wait on the CPU: the peer-to-peer transport has never run between two
GPUs, and a peer mapping that faults there takes the faulting process with it -- this process, not the one that owes the driver its line.
(--share-gpu / --backend gloo: every rank on cuda:0, as in the one-GPU tests.)"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "pi-quant_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--numel", type=int, default=27_264_000)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--share-gpu", action="store_true")
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if args.backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(args.backend)
    import piquant.distributed as pqd

    # Startup self-test: can every GPU of the job address every other one?  If not, say so in the JSON line and leave with rc 0 -- before a
    # single IPC handle is opened, let alone a kernel pointed at a peer's memory.
    try:
        pqd._check_peers_reachable(None, dev, world, rank)
    except RuntimeError as exc:
        if rank == 0:
            print(json.dumps({"refused": str(exc), "ranks": world}), flush=True)
        dist.destroy_process_group()
        return

    n, warm, reps = args.numel, 3, args.reps
    g = torch.Generator(device=dev)
    g.manual_seed(9000 + rank)
    x = torch.empty(n, dtype=torch.float32, device=dev).uniform_(-1.0, 1.0, generator=g)

    def fp32(t):
        if args.backend == "nccl":
            dist.all_reduce(t)
        else:
            h = t.cpu()
            dist.all_reduce(h)
            t.copy_(h)

    exact = x.clone()
    fp32(exact)
    copies = [torch.empty_like(x) for _ in range(warm + reps)]

    def timed(fn):
        for c in copies:
            c.copy_(x)
        for c in copies[:warm]:
            fn(c)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for c in copies[warm:]:
            fn(c)
        torch.cuda.synchronize()
        t = torch.tensor([(time.perf_counter() - t0) / reps], dtype=torch.float64, device=dev)
        if args.backend == "nccl":
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        else:
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX)
            t = h
        return float(t[0])

    out = {"numel": n, "ranks": world, "backend": "RCCL" if args.backend == "nccl" else args.backend, "reps": reps,
           "devices": sorted({f"cuda:{local}"}) if args.share_gpu else f"one per rank ({torch.cuda.device_count()} visible)"}
    t_fp32 = timed(fp32)
    out["all_reduce_fp32_ms"] = round(t_fp32 * 1e3, 4)
    bound = (world * (2.0 / 255) + 2.0 * world / 255) * 0.5 + 1e-5
    results = {}
    for name, kwargs in (("collective", dict(transport="collective")), ("p2p", dict(transport="p2p"))):
        t = timed(lambda c, kw=kwargs: pqd.quantized_all_reduce(c, quant_dtype=torch.uint8, algorithm="direct", **kw))
        res = copies[-1].clone()
        results[name] = res
        err = float((res - exact).abs().max())
        out[f"quantized_all_reduce_direct_u8_{name}"] = {"ms": round(t * 1e3, 4), "algbw_GB/s": round(n * 4 / t / 1e9, 1), "speedup_vs_fp32": round(t_fp32 / t, 3),
                                                         "max_abs_err_vs_fp32_sum": round(err, 6), "within_bound": err <= bound}
    out["p2p_bit_identical_to_collective"] = bool(torch.equal(results["collective"].view(torch.int32), results["p2p"].view(torch.int32)))
    del copies, results, exact
    # BASELINE configs[4] (compute_quant_params of a 2^30-element tensor over the ranks): the same call with its 8-byte MIN all-reduce done by the
    # backend's collective and by the one-wave mailbox kernel
    total = 1 << 30
    b5, e5 = pqd.shard_range(total, rank, world, 8)
    g5 = torch.Generator(device=dev)
    g5.manual_seed(77 + rank)
    shard = torch.empty(e5 - b5, dtype=torch.float32, device=dev).uniform_(-1.0, 1.0, generator=g5)
    if rank == 0:
        shard[12345] = -7.5
    if rank == world - 1:
        shard[-6] = 9.25
    cfg5 = {"numel_total": total, "numel_per_gpu": e5 - b5}
    for name in ("collective", "p2p"):
        for _ in range(3):
            got = pqd.compute_quant_params(shard, dtype=torch.quint8, transport=name)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(20):
            got = pqd.compute_quant_params(shard, dtype=torch.quint8, transport=name)
        t = torch.tensor([(time.perf_counter() - t0) / 20], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX) if args.backend != "nccl" else None
        if args.backend == "nccl":
            td = t.to(dev)
            dist.all_reduce(td, op=dist.ReduceOp.MAX)
            t = td.cpu()
        cfg5[f"ms_per_call_{name}"] = round(float(t[0]) * 1e3, 5)
        cfg5[f"result_{name}"] = list(got)
    import piquant

    cfg5["both_correct"] = tuple(cfg5["result_collective"]) == tuple(cfg5["result_p2p"]) == piquant.quant_params_from_minmax(-7.5, 9.25, piquant.DataType.UINT8)
    out["config5_compute_quant_params_2p30"] = cfg5
    pqd.release_peer_meshes()
    if rank == 0:
        print(json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
