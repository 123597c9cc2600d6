#!/usr/bin/env python3
"""bench.py's side measurements (`extras`): everything that is NOT the contract line.

bench.py imports this module only when extras are on (the default) and calls it behind try/except and -- for N > 1 -- a watchdog: nothing in
here can cost the headline.  `B` is bench.py's state after the timed region (a SimpleNamespace: args, ctx, stream, dev, the rotating buffer
sets xs / outs with their pointers, the prepared C calls, scale / zp, rank / world / use_dist, time_loop, shard_check, ...).

  multi_rank(B)   N > 1: the N = 1 point of the same run (rank 0 alone), both schedules of the quantized all-reduce against RCCL's fp32
                  all-reduce and the bare 8-byte MIN all-reduce, the K steps replayed from a hipGraph, BASELINE configs[4] in its own shape
                  with and without its collective plus the native piquant_hip_compute_quant_params_dist, the weak-scaling variant.
  p2p_child(B)    N > 1, rank 0, AFTER the line has been printed: tools/p2p_all_reduce_bench.py as a child job (the peer-to-peer transport has
                  never run between two GPUs); its record goes to stderr.
  single_gpu(B)   N = 1: graph replay, two streams, config 5 on one GPU, the other operators of the path at the headline size (configs 3 and
                  4, the dynamic path, the scans), blocking calls by wait mode, host-pointer calls.
"""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
HBM_PEAK_GBS = 8000.0
ALGO_BYTES_PER_ELEM = 5
ROUND1_SETS = 6                    # round 1 rotated 6 sets (818 MB): its six 27 MB output buffers stay in the 256 MiB Infinity Cache
DEFAULT_BLOCKING_WAIT = "kernel"   # the library's default (csrc/context.cpp kDefaultBlockingWait)


def max_over_ranks(seconds, dev, use_dist):
    t = torch.tensor([seconds], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def fp32_all_reduce(t):
    """SUM all-reduce of a device fp32 tensor: RCCL moves it as it is; backends without device collectives (gloo in the one-GPU tests) are staged"""
    if dist.get_backend() == "nccl":
        dist.all_reduce(t)
        return
    h = t.cpu()
    dist.all_reduce(h)
    t.copy_(h)


def all_reduce_extras(args, pqd, dev, rank, world, n_total):
    """SURVEY 8(f2) / 8(e2) on N > 1 ranks: both schedules of the quantized all-reduce against the fp32 all-reduce of the same 109 MB tensor, the
    bare 8-byte MIN all-reduce, and what the collective adds to compute_quant_params.  Runs on every rank (collectives inside)."""
    nccl = args.backend == "nccl"
    warm, reps = (3, 10) if nccl else (1, 2)
    g = torch.Generator(device=dev)
    g.manual_seed(9000 + rank)
    x = torch.empty(n_total, dtype=torch.float32, device=dev).uniform_(-1.0, 1.0, generator=g)
    exact = x.clone()
    fp32_all_reduce(exact)
    copies = [torch.empty_like(x) for _ in range(warm + reps)]
    out = {"numel": n_total, "MB_fp32": round(n_total * 4 / 1e6, 1), "ranks": world, "backend": "RCCL" if nccl else args.backend,
           "reps": reps, "timing": "wall clock from a barrier to torch.cuda.synchronize() over `reps` all-reduces of distinct tensors, max over ranks",
           "design_prediction": "DESIGN.md section 7 (8 GPUs, uint8 wire): direct/mesh 78 us of kernels per rank around ~45 us of wire (2 x 7/8 x 27 MB over 7 xGMI links), "
                                "ring 152 us of kernels + 14 hops; fp32 RCCL all-reduce moves 4x the bytes"}

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
        return max_over_ranks((time.perf_counter() - t0) / reps, dev, True)

    t = timed(fp32_all_reduce)
    out["all_reduce_fp32"] = {"ms": round(t * 1e3, 4), "algbw_GB/s": round(n_total * 4 / t / 1e9, 1)}
    for algo in ("direct", "ring"):
        try:
            t = timed(lambda c, a=algo: pqd.quantized_all_reduce(c, quant_dtype=torch.uint8, algorithm=a))
            res = copies[-1]
            err = float((res - exact).abs().max())
            # every rank must hold the same bits (all ranks decode the same gathered bytes)
            digest = res.view(torch.int32).to(torch.int64).sum().reshape(1)
            lo, hi = digest.clone(), digest.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            # every value is quantized twice (direct) or up to G times (ring) on a grid of (range / 255): per quantization half a step of
            # a range that is at most 2 (one rank's values) resp. 2 G (the sum)
            bound = (world * (2.0 / 255) + 2.0 * world / 255) * 0.5 * (1 if algo == "direct" else world) + 1e-5
            out[f"quantized_all_reduce_{algo}_u8"] = {"ms": round(t * 1e3, 4), "algbw_GB/s": round(n_total * 4 / t / 1e9, 1),
                                                        "speedup_vs_fp32": round(out["all_reduce_fp32"]["ms"] / (t * 1e3), 3),
                                                        "max_abs_err_vs_fp32_sum": round(err, 6), "err_bound": round(bound, 6), "within_bound": err <= bound,
                                                        "ranks_bit_identical": int(lo[0]) == int(hi[0])}
        except Exception as exc:
            out[f"quantized_all_reduce_{algo}_u8"] = {"error": repr(exc)}
    del copies, exact
    # the path's only collective: 2 x int32 MIN
    keys = torch.zeros(2, dtype=torch.int32, device=dev)
    for _ in range(5):
        dist.all_reduce(keys, op=dist.ReduceOp.MIN)
    torch.cuda.synchronize()
    dist.barrier()
    kreps = 100 if nccl else 20
    t0 = time.perf_counter()
    for _ in range(kreps):
        dist.all_reduce(keys, op=dist.ReduceOp.MIN)
        torch.cuda.synchronize()
    out["min_all_reduce_8_bytes"] = {"us_per_call": round(max_over_ranks((time.perf_counter() - t0) / kreps, dev, True) * 1e6, 2),
                                     "note": "dist.all_reduce(int32[2], MIN) + synchronize, one at a time: latency, not bandwidth"}
    return out, x


def p2p_all_reduce_child_job(args, world):
    """Rank 0 only: tools/p2p_all_reduce_bench.py as a child job of `world` ranks on the same GPUs (the mesh all-reduce over peer-mapped buffers next
    to the collective transport and the fp32 all-reduce).  A separate job because the peer-to-peer transport has never run between two GPUs: a
    peer mapping that faults takes the faulting PROCESS with it -- the child, not the process that owes the driver its line."""
    import signal
    import socket
    import subprocess

    # PIQUANT_BENCH_P2P: "0" never, "1" always; unset = only where it cannot cost a LATER run -- a scaling sweep runs N = 1, 2, 4, 8 back to back on one
    # node, and a transport that has never run between two GPUs gets its first contact behind the LAST of them (N >= 8), not between them
    # (--share-gpu: the one-GPU test plumbing, where it has run many times)
    want = os.environ.get("PIQUANT_BENCH_P2P", "auto")
    if want == "0":
        return "not run: PIQUANT_BENCH_P2P=0"
    if want != "1" and not (world >= 8 or args.share_gpu):
        return f"not run at N = {world}: first contact of the peer-to-peer transport with real peers is kept behind the last run of a sweep (N >= 8); PIQUANT_BENCH_P2P=1 runs it anyway"
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           str(ROOT / "tools" / "p2p_all_reduce_bench.py"), "--numel", str(args.numel), "--backend", args.backend] + (["--share-gpu"] if args.share_gpu else [])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE",
                                                           "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT",
                                                           "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_USE_AGENT_STORE", "TORCH_NCCL_ASYNC_ERROR_HANDLING")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    limit = float(os.environ.get("PIQUANT_BENCH_P2P_LIMIT_S", "90"))
    try:
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        try:
            so, se = proc.communicate(timeout=limit)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)      # the child job's own process group (start_new_session): nobody else's
            so, se = proc.communicate()
            return {"error": f"child job did not finish within {limit} s", "stderr_tail": se[-600:]}
        lines = [ln for ln in so.splitlines() if ln.startswith("{")]
        if proc.returncode != 0 or not lines:
            return {"error": f"child job exit code {proc.returncode}", "stderr_tail": se[-600:]}
        rec = json.loads(lines[-1])
        rec["how"] = "tools/p2p_all_reduce_bench.py as a child job of this run (own processes and process group on the same GPUs; this run's ranks idle on the CPU meanwhile)"
        return rec
    except Exception as exc:
        return {"error": repr(exc)}


def native_dist_entry(args, ctx, shard, dev, rank, world, want):
    """piquant_hip_compute_quant_params_dist (csrc/capi.cpp: scan + ncclAllReduce(2 x int32, ncclMin) on the context's stream + epilogue, no Python
    between them) on a communicator of its own over all ranks: rank 0 draws the unique id, the process group carries it to the others."""
    import ctypes

    from piquant import DataType

    rccl = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]

    rccl.ncclGetUniqueId.argtypes = [ctypes.POINTER(UniqueId)]
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    uid = UniqueId()
    box = [None]
    if rank == 0:
        assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
        box[0] = bytes(ctypes.string_at(ctypes.addressof(uid), 128))
    dist.broadcast_object_list(box, src=0)
    ctypes.memmove(ctypes.addressof(uid), box[0], 128)
    comm = ctypes.c_void_p()
    rc = rccl.ncclCommInitRank(ctypes.byref(comm), world, uid, rank)
    if rc != 0:
        raise RuntimeError(f"ncclCommInitRank -> {rc}")
    try:
        for _ in range(3):
            got = ctx.compute_quant_params_dist_ptr(shard.data_ptr(), DataType.F32, shard.numel(), DataType.UINT8, comm.value)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(20):
            got = ctx.compute_quant_params_dist_ptr(shard.data_ptr(), DataType.F32, shard.numel(), DataType.UINT8, comm.value)
        t = max_over_ranks((time.perf_counter() - t0) / 20, dev, True)
    finally:
        rccl.ncclCommDestroy(comm)
    return {"ms_per_call": round(t * 1e3, 5), "result": list(got), "result_correct": tuple(got) == want,
            "note": f"C entry point on its own {world}-rank RCCL communicator: scan, ncclAllReduce, 8-byte D2H, epilogue -- one call, synchronous"}


def n1_reference(B):
    """N > 1: the N = 1 point of the SAME run -- rank 0 alone quantizes the whole tensor with the headline's protocol (K steps per window, cold
    rotation of args.sets full-size sets) while the other ranks wait at the barrier behind it."""
    args, ctx, stream, dev, rank, n_total, scale, zp, gib_per_step, c_quantize, time_loop = B.args, B.ctx, B.stream, B.dev, B.rank, B.n_total, B.scale, B.zp, B.gib_per_step, B.c_quantize, B.time_loop
    from piquant import DataType, RoundMode
    from piquant._bootstrap import C_LIB
    c_quantize = C_LIB.piquant_quantize     # the whole tensor in one call: the plain call, as in the N = 1 line (the ranks' shard calls use the position-independent twin)
    n1_ref = None
    try:
        if rank == 0:
            rx, ro = [], []
            for s_ in range(args.sets):
                g = torch.Generator(device=dev)
                g.manual_seed(700_000 + s_)
                rx.append(torch.empty(n_total, dtype=torch.float32, device=dev).uniform_(-1.0, 1.0, generator=g))
                ro.append(torch.empty(n_total, dtype=torch.uint8, device=dev))
            rargs = [(ctx._ctx, rx[k].data_ptr(), DataType.F32.value, ro[k].data_ptr(), DataType.UINT8.value, n_total, scale, zp, RoundMode.NEAREST.value)
                     for k in range(args.sets)]

            def rstep(i):
                c_quantize(*rargs[i % args.sets])

            with torch.cuda.stream(stream):
                for i in range(200 + args.warmup):
                    rstep(i)
                rw, re = [], []
                for w in range(min(args.windows, 15)):
                    a, b_ = time_loop(rstep, args.steps, stream, base=w * args.steps)
                    rw.append(a)
                    re.append(b_)
            rw.sort()
            re.sort()
            rmed, remed = rw[len(rw) // 2], re[len(re) // 2]
            n1_ref = {"GiB/s": round(gib_per_step * args.steps / rmed, 2), "ms_per_step": round(rmed / args.steps * 1e3, 6),
                      "avg_launch_us": round(remed / args.steps * 1e6, 3), "roofline_frac": round(ALGO_BYTES_PER_ELEM * n_total / (remed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                      "bit_exact": B.shard_check(rx[0], ro[0], scale, zp), "windows": len(rw),
                      "note": f"rank 0 alone, the whole {n_total}-element tensor on one GPU, {args.sets} cold buffer sets, median window of K = {args.steps} steps: "
                              "the N = 1 point measured inside this N > 1 run (compare with the driver's N = 1 line)"}
            del rx, ro
    except Exception as exc:
        n1_ref = {"error": repr(exc)}
    dist.barrier()
    return n1_ref


def all_reduce_109mb(B):
    """N > 1: both schedules of the quantized all-reduce (collective transport) against the fp32 all-reduce of the same 109 MB tensor + the 8-byte MIN."""
    args, ctx, stream, dev, rank, world, n_total = B.args, B.ctx, B.stream, B.dev, B.rank, B.world, B.n_total
    import piquant.distributed as pqd
    try:
        all_reduce, _x = all_reduce_extras(args, pqd, dev, rank, world, n_total)
        del _x
    except Exception as exc:
        all_reduce = {"error": repr(exc)}
    ctx.set_stream(stream.cuda_stream)
    ctx.set_blocking(False)

    return all_reduce


def graph_replay(B):
    """The same K steps replayed from a hipGraph (every stream-ordered call of the library is capturable): what is left of a step when the host's
    per-launch work is taken out of it.  Runs on every rank (barriers)."""
    args, ctx, stream, dev, rank, use_dist, gib_per_step, step = B.args, B.ctx, B.stream, B.dev, B.rank, B.use_dist, B.gib_per_step, B.step
    graphed = None
    try:
        g, captured = None, 1
        try:
            with torch.cuda.stream(stream):
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                # thread_local: with RCCL the process group's watchdog thread polls events while this thread captures; in the default (global)
                # mode that would invalidate the capture
                with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
                    for i in range(args.steps):
                        step(i)
        except Exception as exc:
            captured, capture_error = 0, repr(exc)
        if use_dist:      # the replay loop below has barriers: every rank runs it or none does
            flag = torch.tensor([captured], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag[0]) == 0 and captured:
                captured, capture_error = 0, "capture failed on another rank"
        if not captured:
            raise RuntimeError(capture_error)
        with torch.cuda.stream(stream):
            gw = []
            for _ in range(3 + min(args.windows, 15)):
                if use_dist:
                    dist.barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                g.replay()
                eg = torch.cuda.Event()
                eg.record(stream)
                while not eg.query():
                    pass
                gw.append(time.perf_counter() - t0)
                torch.cuda.synchronize()
            gw = gw[3:]
        tg = torch.tensor(gw, dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        gs = sorted(float(v) for v in tg)
        gmed = gs[len(gs) // 2]
        graphed = {"GiB/s": round(gib_per_step * args.steps / gmed, 2), "ms_per_step": round(gmed / args.steps * 1e3, 6), "windows": len(gs),
                   "GiB/s_min": round(gib_per_step * args.steps / gs[-1], 2), "GiB/s_max": round(gib_per_step * args.steps / gs[0], 2),
                   "note": f"the K = {args.steps} steps of a window captured once into a hipGraph and replayed; same barrier + synchronize bracket, median window, max over ranks"}
        del g
        ctx.set_stream(stream.cuda_stream)
        ctx.set_blocking(False)
    except Exception as exc:
        graphed = {"error": repr(exc)}
    return graphed


def two_streams(B):
    """Independent calls issued alternately on two streams (a context each): the next tensor's ramp runs under this one's drain.  What a caller with
    many tensors and no order between them can have; never `value` (whose steps share ONE stream, as a plain caller's do)."""
    args, stream, nsets, gib_per_step, call_args, c_quantize = B.args, B.stream, B.nsets, B.gib_per_step, B.call_args, B.c_quantize
    import piquant
    two_streams = None
    two_streams = None
    try:
        s2 = [torch.cuda.Stream(), torch.cuda.Stream()]
        c2 = [piquant.Context(), piquant.Context()]
        for c, s in zip(c2, s2):
            c.set_stream(s.cuda_stream)
            c.set_blocking(False)
            c.assume_device_pointers(True)
        a2 = [[(c._ctx,) + call_args[k][1:] for k in range(nsets)] for c in c2]
        tw = []
        for w in range(13):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(w * args.steps, (w + 1) * args.steps):
                c_quantize(*a2[i & 1][i % nsets])
            ends = [torch.cuda.Event(), torch.cuda.Event()]
            for e, s in zip(ends, s2):
                e.record(s)
            while not (ends[0].query() and ends[1].query()):
                pass
            tw.append(time.perf_counter() - t0)
            torch.cuda.synchronize()
        tw = sorted(tw[3:])
        tmed = tw[len(tw) // 2]
        two_streams = {"GiB/s": round(gib_per_step * args.steps / tmed, 2), "ms_per_step": round(tmed / args.steps * 1e3, 6), "windows": len(tw),
                       "note": f"the same K = {args.steps} calls per window, even ones on one stream and odd ones on another (two contexts); wall clock from the first call "
                               "to the completion of both streams, median window"}
        del c2
    except Exception as exc:
        two_streams = {"error": repr(exc)}
    return two_streams


def independent_calls(B):
    """The headline's K steps per window on ONE stream with the context's calls declared independent (piquant_hip_set_independent_calls: no
    barrier bit on the dispatch packets, a call's ramp runs under the previous call's drain).  The steps of the benchmark ARE independent --
    distinct buffer sets -- but that is the caller's knowledge, not the library's: the headline keeps the default, ordered semantics, this
    is what a caller that says so gets.  Every rank runs it; max over ranks."""
    args, stream, dev, use_dist = B.args, B.stream, B.dev, B.use_dist
    import piquant

    try:
        c = piquant.Context()
        c.set_stream(stream.cuda_stream)
        c.set_blocking(False)
        c.assume_device_pointers(True)
        c.set_independent_calls(True)
        calls = [(c._ctx,) + a[1:] for a in B.call_args]
        walls, evs = [], []
        with torch.cuda.stream(stream):
            for i in range(200):
                B.c_quantize(*calls[i % B.nsets])
            for w in range(min(args.windows, 15)):
                if use_dist:
                    dist.barrier()
                a, b_ = B.time_loop(lambda i: B.c_quantize(*calls[i % B.nsets]), args.steps, stream, base=w * args.steps)
                walls.append(a)
                evs.append(b_)
        t = torch.tensor([walls, evs], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ws, es = sorted(float(v) for v in t[0]), sorted(float(v) for v in t[1])
        wmed, emed = ws[len(ws) // 2], es[len(es) // 2]
        n_max = -(-B.n_total // B.world)
        rec = {"GiB/s": round(B.gib_per_step * args.steps / wmed, 2), "ms_per_step": round(wmed / args.steps * 1e3, 6), "windows": len(ws),
               "us_per_call_from_events": round(emed / args.steps * 1e6, 3),
               "roofline_frac": round(ALGO_BYTES_PER_ELEM * n_max / (emed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
               "bit_exact": B.shard_check(B.xs[0], B.outs[0], B.scale, B.zp),
               "note": "same calls, same stream, same protocol as the headline; the context was told its calls are independent, so the kernels are launched without "
                       "the barrier bit (hipExtAnyOrderLaunch) and overlap at their edges: the time per call is the stream's, not one kernel's duration"}
        del c
        return rec
    except Exception as exc:
        return {"error": repr(exc)}


def config5(B):
    """BASELINE configs[4]: compute_quant_params over a 2^30-element fp32 tensor sharded across the ranks -- every rank scans its shard in HBM,
    ONE 8-byte all_reduce(MIN) over RCCL/xGMI, identical double-precision epilogue everywhere.  Runs on every rank (it contains the collective)."""
    args, ctx, stream, dev, rank, world, use_dist = B.args, B.ctx, B.stream, B.dev, B.rank, B.world, B.use_dist
    import piquant
    import piquant.distributed as pqd
    from piquant import DataType
    rec5, native5 = None, None
    rec5 = None
    try:
        total5 = 1 << 30
        b5, e5 = pqd.shard_range(total5, rank, world, 8)
        g5 = torch.Generator(device=dev)
        g5.manual_seed(77 + rank)
        shard = torch.empty(e5 - b5, dtype=torch.float32, device=dev).uniform_(-1.0, 1.0, generator=g5)
        if rank == 0:
            shard[12345] = -7.5            # the global extremes live on different ranks
        if rank == world - 1:
            shard[-6] = 9.25
        with torch.cuda.stream(stream):
            for _ in range(3):
                got5 = pqd.compute_quant_params(shard, dtype=torch.quint8, ctx=ctx)
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            t0 = time.perf_counter()
            for _ in range(20):
                got5 = pqd.compute_quant_params(shard, dtype=torch.quint8, ctx=ctx)
            torch.cuda.synchronize()
            t5 = (time.perf_counter() - t0) / 20
        t5t = torch.tensor([t5], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(t5t, op=dist.ReduceOp.MAX)
        want5 = piquant.quant_params_from_minmax(-7.5, 9.25, DataType.UINT8)
        rec5 = {"numel_total": total5, "numel_per_gpu": e5 - b5, "ms_per_call": round(float(t5t[0]) * 1e3, 5),
                   "aggregate_GB/s": round(4.0 * total5 / float(t5t[0]) / 1e9, 1), "result": list(got5), "result_correct": tuple(got5) == want5,
                   "note": "HIP scan of the local shard + " + (f"one 8-byte all_reduce(MIN) over {'RCCL' if args.backend == 'nccl' else args.backend} ({world} ranks)"
                                                              if world > 1 else "no collective (one rank: the all-reduce is skipped)") +
                           " + host epilogue, synchronous per call"}
        if world > 1:
            # the same call without its collective (local scan + 8-byte D2H + epilogue): what the all-reduce adds
            with torch.cuda.stream(stream):
                for _ in range(3):
                    pqd.local_minmax_keys(shard, ctx).cpu()
                torch.cuda.synchronize()
                dist.barrier()
                t0 = time.perf_counter()
                for _ in range(20):
                    pqd.local_minmax_keys(shard, ctx).cpu()
                tl = max_over_ranks((time.perf_counter() - t0) / 20, dev, True)
            rec5["ms_per_call_without_collective"] = round(tl * 1e3, 5)
            rec5["collective_adds_ms"] = round((float(t5t[0]) - tl) * 1e3, 5)
            if args.backend == "nccl":
                try:
                    ctx.set_blocking(True)
                    native5 = native_dist_entry(args, ctx, shard, dev, rank, world, want5)
                except Exception as exc:
                    native5 = {"error": repr(exc)}
                rec5["native_entry_piquant_hip_compute_quant_params_dist"] = native5
        del shard
        ctx.set_stream(stream.cuda_stream)
        ctx.set_blocking(False)
    except Exception as exc:   # never lose the headline line to the secondary measurement
        rec5 = {"error": repr(exc)}
        ctx.set_stream(stream.cuda_stream)
        ctx.set_blocking(False)
    return rec5


def weak_scaling(B):
    """N > 1: every rank quantizes its OWN full-size tensor (the data-parallel gradient case), same protocol."""
    args, ctx, stream, dev, rank, world, n_total, scale, zp, gib_per_step, c_quantize, time_loop = B.args, B.ctx, B.stream, B.dev, B.rank, B.world, B.n_total, B.scale, B.zp, B.gib_per_step, B.c_quantize, B.time_loop
    from piquant import DataType, RoundMode
    weak = None
    weak = None
    try:
        wsets = args.sets
        wx, wo = [], []
        for s_ in range(wsets):
            g = torch.Generator(device=dev)
            g.manual_seed(500_000 + 1000 * rank + s_)
            wx.append(torch.empty(n_total, dtype=torch.float32, device=dev).uniform_(-1.0, 1.0, generator=g))
            wo.append(torch.empty(n_total, dtype=torch.uint8, device=dev))
        pwi, pwo = [t_.data_ptr() for t_ in wx], [t_.data_ptr() for t_ in wo]

        wargs = [(ctx._ctx, pwi[k], DataType.F32.value, pwo[k], DataType.UINT8.value, n_total, scale, zp, RoundMode.NEAREST.value) for k in range(wsets)]

        def wstep(i):
            c_quantize(*wargs[i % wsets])

        with torch.cuda.stream(stream):
            for i in range(max(args.warmup, 20)):
                wstep(i)
            torch.cuda.synchronize()
            dist.barrier()
            ww, we = time_loop(wstep, args.steps, stream)
            dist.barrier()
        wt = torch.tensor([ww, we], dtype=torch.float64, device=dev)
        dist.all_reduce(wt, op=dist.ReduceOp.MAX)
        weak = {"scaling": "weak", "numel_per_gpu": n_total, "GiB/s": round(world * gib_per_step * args.steps / float(wt[0]), 2),
                "ms_per_step": round(float(wt[0]) / args.steps * 1e3, 6), "avg_launch_us": round(float(wt[1]) / args.steps * 1e6, 3),
                "per_gpu_roofline_frac": round(ALGO_BYTES_PER_ELEM * n_total / (float(wt[1]) / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                "note": "every rank quantizes its own 27 264 000-element tensor (round 1's headline for N > 1); value = all ranks' bytes / max time"}
        del wx, wo
    except Exception as exc:
        weak = {"error": repr(exc)}
    return weak


def _rearm(B):
    B.ctx.set_stream(B.stream.cuda_stream)
    B.ctx.set_blocking(False)


def multi_rank(B):
    """N > 1, every rank (collectives inside).  Results are put into B.side as they finish: what the watchdog's line carries."""
    B.result["n1_reference"] = n1_reference(B)        # in the line even if a later side measurement runs into the watchdog
    todo = (("config5_sharded_compute_quant_params", config5),)      # the default for N > 1: the one path with a collective (bench.py --extras: all of them)
    if B.args.extras:
        todo = (("independent_calls_one_stream", independent_calls), ("all_reduce_109MB", all_reduce_109mb), ("steps_replayed_from_a_hipgraph", graph_replay)) + todo + \
               (("weak_scaling_own_tensor_per_gpu", weak_scaling),)
    for key, fn in todo:
        try:
            B.side[key] = fn(B)
        except Exception as exc:
            B.side[key] = {"error": repr(exc)}
        _rearm(B)
    return dict(B.side)


def p2p_child(B):
    """N > 1, rank 0, after the line is out: the peer-to-peer transport as a child job; one line on stderr (`bench.py p2p_transport_child_job: {json}`)."""
    rec = p2p_all_reduce_child_job(B.args, B.world)
    print("bench.py p2p_transport_child_job: " + json.dumps(rec), file=sys.stderr, flush=True)
    return rec


def single_gpu(B):
    """N = 1: the side measurements of the single-GPU line."""
    ctx, stream, dev, n, nsets, scale, zp, gib_per_step, xs, ptr_in, ptr_out, call_args, c_quantize, step, time_loop = B.ctx, B.stream, B.dev, B.n, B.nsets, B.scale, B.zp, B.gib_per_step, B.xs, B.ptr_in, B.ptr_out, B.call_args, B.c_quantize, B.step, B.time_loop
    import piquant
    from piquant import DataType, RoundMode
    xs0_host = B.xs0_host
    graphed = graph_replay(B)
    _rearm(B)
    two = two_streams(B)
    indep = independent_calls(B)
    rec5 = config5(B)
    _rearm(B)
    extras = {"steps_replayed_from_a_hipgraph": graphed, "independent_calls_on_two_streams": two, "independent_calls_one_stream": indep,
              "config5_sharded_compute_quant_params": rec5}

    def gbs_plain(bytes_per_elem, ev_s, reps):
        return round(bytes_per_elem * n / (ev_s / reps) / 1e9, 1)

    with torch.cuda.stream(stream):
        # same kernel with everything resident in the Infinity Cache (one 136 MB set): NOT the headline
        w, e = time_loop(lambda i: ctx.quantize_ptr(ptr_in[0], DataType.F32, ptr_out[0], DataType.UINT8, n, scale, zp, RoundMode.NEAREST, _device_ptrs=True), 200, stream)
        extras["warm_cache_single_set"] = {"GiB/s": round(gib_per_step * 200 / w, 1), "avg_launch_us": round(e / 200 * 1e6, 3)}
        # round 1's protocol: the same launches rotating over 6 sets (818 MB) only
        w, e = time_loop(lambda i: c_quantize(*call_args[i % ROUND1_SETS]), 600, stream)
        extras["rotation_of_6_sets_818MB_round1_protocol"] = {"GiB/s": round(gib_per_step * 600 / w, 1), "avg_launch_us": round(e / 600 * 1e6, 3),
                                                           "GB/s": gbs_plain(5, e, 600),
                                                           "note": "what round 1 reported as the headline: its six output buffers (164 MB) fit in the 256 MiB Infinity Cache"}
        # cold inputs, ONE output buffer: what a caller that quantizes tensor after tensor into the same staging buffer sees (the 27 MB of
        # output stay in the Infinity Cache; every input byte still comes from HBM).  NOT the headline, which writes to cold buffers too.
        reuse_args = [(ctx._ctx, ptr_in[k], DataType.F32.value, ptr_out[0], DataType.UINT8.value, n, scale, zp, RoundMode.NEAREST.value) for k in range(nsets)]
        w, e = time_loop(lambda i: c_quantize(*reuse_args[i % nsets]), 600, stream)
        extras["cold_inputs_one_output_buffer"] = {"GiB/s": round(gib_per_step * 600 / w, 1), "avg_launch_us": round(e / 600 * 1e6, 3), "GB/s": gbs_plain(5, e, 600),
                                                   "note": f"inputs rotate over the {nsets} cold sets, every launch writes the same 27 MB output buffer"}
        # Reference layout (the default of the plain calls since round 6: the bytes of a CPU reference context of num_threads pool threads),
        # for one partition, for 255 (the reference's Python default on this host: cpu_count - 1) and switched off.  A wave tile that a partition's scalar head
        # or tail reaches into handles those positions itself, inside the one vector launch (round 5: a second, dependent patch launch, 26 us at 255 threads).
        try:
            layout = {}
            for key, on, threads in (("1_reference_threads", True, 1), ("255_reference_threads", True, 255), ("layout_off_position_independent", False, 1)):
                ctx.set_reference_layout(on, threads=threads)
                for i in range(nsets):
                    step(i)
                w, e = time_loop(step, 300, stream)
                layout[key] = {"us_per_call": round(e / 300 * 1e6, 3), "roofline_frac": round(5 * n / (e / 300) / 1e9 / 8000.0, 4)}
            extras["reference_layout_mode"] = dict(layout, note="the headline's calls for three settings of the context: the output equals, byte for byte, what the reference's "
                                                   "AVX-512 context of that many pool threads writes (the headline itself runs with the context's default: the layout of "
                                                   f"num_threads = {ctx._num_threads} pool threads); round 6: handled inside the vector launch, no patch kernel")
        except Exception as exc:
            extras["reference_layout_mode"] = {"error": repr(exc)}
        finally:
            ctx.set_reference_layout(True)      # the default again: the context's own num_threads
        # reference semantics: every call waits for completion (blocking context); A/B of the three ways to wait (csrc/context.cpp wait_stream)
        ctx.set_blocking(True)
        ctx.assume_device_pointers(True)      # step() makes the raw C call: the context must know these are device pointers
        blocking = {}
        for mode in ("sync", "write32", "kernel", "event"):
            ctx.set_blocking_wait(mode)
            for i in range(20):
                step(i)
            t0 = time.perf_counter()
            for i in range(300):
                step(i)
            tb = time.perf_counter() - t0
            blocking[mode] = {"GiB/s": round(gib_per_step * 300 / tb, 1), "ms_per_call": round(tb / 300 * 1e3, 5)}
        ctx.set_blocking_wait(DEFAULT_BLOCKING_WAIT)
        ctx.set_blocking(False)
        extras["blocking_calls"] = dict(blocking[DEFAULT_BLOCKING_WAIT], wait=DEFAULT_BLOCKING_WAIT, by_wait_mode=blocking,
                                        note="piquant_quantize returning after completion, as the reference's calls do; sync = hipStreamSynchronize, "
                                             "write32 = hipStreamWriteValue32 into a pinned host word + host spin, kernel = one-thread kernel writing that word, "
                                             "event = the work kernel's own stop event (hipExtLaunchKernelGGL) polled with hipEventQuery")

        def gbs(bytes_per_elem, ev_s, reps):
            return round(bytes_per_elem * n / (ev_s / reps) / 1e9, 1)

        reps = 200

        def time_loop(fn, reps_, stream_, _one=time_loop):   # noqa: F811 -- the per-kernel extras below: median of five windows of `reps` launches, not one
            es = sorted(_one(fn, reps_, stream_)[1] for _ in range(5))   # window (a single window of a 12 us kernel is 2.4 ms: one slow start moves it by 3 %)
            return None, es[2]
        # config 3 moves 68 MB per launch: as many buffer sets as the headline (1.6 GB) -- with the 4 sets of round 1 (272 MB) the 256 MiB
        # Infinity Cache served a good part of the traffic and both kernels looked 1-1.5 us faster than they are from HBM
        nb = nsets
        xb = [x.to(torch.bfloat16) for x in xs]
        q4 = [torch.empty((n + 1) // 2, dtype=torch.uint8, device=dev) for _ in range(nb)]
        s4, z4 = piquant.torch.compute_quant_params(xb[0], dtype=torch.quint4x2)
        ctx.set_stream(stream.cuda_stream)
        ctx.set_blocking(False)
        _, e = time_loop(lambda i: ctx.quantize_ptr(xb[i % nb].data_ptr(), DataType.BF16, q4[i % nb].data_ptr(), DataType.UINT4, n, s4, z4, RoundMode.NEAREST, _device_ptrs=True), reps, stream)
        extras["quantize_bf16_u4"] = {"GB/s": gbs(2.5, e, reps), "avg_launch_us": round(e / reps * 1e6, 3), "buffer_sets": nb}
        _, e = time_loop(lambda i: ctx.dequantize_ptr(q4[i % nb].data_ptr(), DataType.UINT4, xb[i % nb].data_ptr(), DataType.BF16, n, s4, z4, piquant.ReduceOp.SET, _device_ptrs=True), reps, stream)
        extras["dequantize_u4_bf16_set"] = {"GB/s": gbs(2.5, e, reps), "avg_launch_us": round(e / reps * 1e6, 3), "buffer_sets": nb}
        del xb, q4
        _, e = time_loop(lambda i: ctx.quantize_ptr(ptr_in[i % nsets], DataType.F32, ptr_out[i % nsets], DataType.UINT8, n, scale, zp, RoundMode.STOCHASTIC), reps, stream)
        extras["quantize_f32_u8_stochastic"] = {"GB/s": gbs(5, e, reps), "avg_launch_us": round(e / reps * 1e6, 3)}
        _, e = time_loop(lambda i: ctx.dequantize_ptr(ptr_out[i % nsets], DataType.UINT8, ptr_in[i % nsets], DataType.F32, n, scale, zp, piquant.ReduceOp.ADD), reps, stream)
        extras["dequantize_u8_f32_add"] = {"GB/s": gbs(9, e, reps), "avg_launch_us": round(e / reps * 1e6, 3)}
        y = [torch.empty_like(x) for x in xs[:8]]
        _, e = time_loop(lambda i: ctx.quantize_dequantize_ptr(ptr_in[i % 8], DataType.F32, y[i % 8].data_ptr(), DataType.UINT8, n, scale, zp,
                                                                RoundMode.NEAREST, piquant.ReduceOp.SET), reps, stream)
        extras["requantize_f32_u8_set"] = {"GB/s": gbs(8, e, reps), "avg_launch_us": round(e / reps * 1e6, 3),
                                           "note": "fused quantize->dequantize, 4 B read + 4 B written per element"}
        del y
        rec = torch.empty(16, dtype=torch.uint8, device=dev)
        rec_ptr = rec.data_ptr()
        _, e = time_loop(lambda i: ctx.quantize_dynamic_ptr(ptr_in[i % nsets], DataType.F32, ptr_out[i % nsets], DataType.UINT8, n, rec_ptr, RoundMode.NEAREST,
                                                            _device_ptrs=True), reps, stream)
        extras["quantize_dynamic_f32_u8"] = {"GB/s": gbs(5, e, reps), "avg_us_per_call": round(e / reps * 1e6, 3),
                                             "note": "compute_quant_params + quantize as ONE launch: the tensor stays in VGPRs/LDS between the min/max pass and "
                                                     "the quantization (5 B/elem of HBM traffic, x read once); no host sync"}
        ctx.set_fusion(False)
        _, e = time_loop(lambda i: ctx.quantize_dynamic_ptr(ptr_in[i % nsets], DataType.F32, ptr_out[i % nsets], DataType.UINT8, n, rec_ptr, RoundMode.NEAREST,
                                                            _device_ptrs=True), reps, stream)
        ctx.set_fusion(True)
        extras["quantize_dynamic_f32_u8_unfused"] = {"GB/s": gbs(9, e, reps), "avg_us_per_call": round(e / reps * 1e6, 3),
                                                     "note": "same call with fusion off: scan (parameter epilogue in its last block) + quantize, 9 B/elem: x read twice"}
        # reduction step of the mesh all-reduce: 7 quantized chunks from 7 peers summed into the accumulator in one pass
        groups = 4                        # 4 x (7 x 27 MB of chunks + a 109 MB accumulator read and written) = 1.6 GB in rotation
        recs7 = [[torch.empty(16, dtype=torch.uint8, device=dev) for _ in range(7)] for _ in range(groups)]
        q7 = [[torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(7)] for _ in range(groups)]
        for g_ in range(groups):
            for i in range(7):
                piquant.torch.quantize_dynamic(xs[(7 * g_ + i) % nsets], dtype=torch.uint8, ctx=ctx, out=q7[g_][i], params=recs7[g_][i])
        accs = [torch.zeros(n, device=dev) for _ in range(groups)]
        ptr_q7 = [[t_.data_ptr() for t_ in grp] for grp in q7]
        ptr_r7 = [[t_.data_ptr() for t_ in grp] for grp in recs7]
        _, e = time_loop(lambda i: ctx.dequantize_sum_ptr(ptr_q7[i % groups], ptr_r7[i % groups], DataType.UINT8, accs[i % groups].data_ptr(), DataType.F32, n,
                                                          piquant.ReduceOp.ADD, _device_ptrs=True), 100, stream)
        extras["dequantize_sum_7x_u8_f32_add"] = {"GB/s": gbs(15, e, 100), "avg_launch_us": round(e / 100 * 1e6, 3),
                                                  "note": "acc += sum of 7 quantized inputs with device-resident parameters, one pass (15 B/elem); "
                                                          "7 dequantize(ADD) calls move 63 B/elem"}
        del q7, accs
        ctx.set_stream(stream.cuda_stream)
        ctx.set_blocking(False)
        keys = torch.empty(2, dtype=torch.int32, device=dev)
        _, e = time_loop(lambda i: ctx.minmax_keys_ptr(ptr_in[i % nsets], DataType.F32, n, keys.data_ptr(), True), reps, stream)
        extras["minmax_f32"] = {"GB/s": gbs(4, e, reps), "avg_launch_us": round(e / reps * 1e6, 3),
                                "note": "piquant_hip_minmax_keys: one launch, the highest block sweeps the per-block result words into the key pair (a read-only sweep of the same "
                                        "bytes with no arithmetic and no end: 16.6-18.3 us; the scan's loop alone 17.1, + block reduction 17.7, profiles/r04_tune_mm8_summary.txt)"}
        xb16 = [x.to(torch.bfloat16) for x in xs]
        _, e = time_loop(lambda i: ctx.minmax_keys_ptr(xb16[i % nsets].data_ptr(), DataType.BF16, n, keys.data_ptr(), True), reps, stream)
        extras["minmax_bf16"] = {"GB/s": gbs(2, e, reps), "avg_launch_us": round(e / reps * 1e6, 3), "buffer_sets": nsets,
                                 "note": "the same scan over bf16 (54.5 MB per launch: half the bytes behind the same fixed ramp and end)"}
        del xb16
        t0 = time.perf_counter()
        for i in range(50):
            piquant.torch.compute_quant_params(xs[i % nsets], dtype=torch.quint8)
        extras["compute_quant_params_f32_call"] = {"ms_per_call": round((time.perf_counter() - t0) / 50 * 1e3, 5),
                                                   "note": "full C-ABI call through piquant.torch: scan whose last block publishes the keys into a pinned host mailbox + host spin + double epilogue"}
        ctx.set_stream(stream.cuda_stream)
        ctx.set_blocking(False)
    # the reference's own calling convention: host buffers in, host buffers out, blocking (never `value`)
    if xs0_host is not None:
        def host_rotation(hctx):
            # eight tensors of the caller's in rotation (1.1 GB: DRAM, not the sockets' 512 MB of L3), all allocated and filled by this thread
            hxs = [xs0_host] + [xs0_host.copy() for _ in range(7)]
            houts = [np.zeros(n, dtype=np.uint8) for _ in range(8)]
            best = float("inf")
            for rot in range(4):
                t0 = time.perf_counter()
                for hx, ho in zip(hxs, houts):
                    hctx.quantize_ptr(hx.ctypes.data, DataType.F32, ho.ctypes.data, DataType.UINT8, n, scale, zp, RoundMode.NEAREST)
                if rot:
                    best = min(best, (time.perf_counter() - t0) / len(hxs))
            return best, houts[0]

        try:   # what an UNCHANGED caller of the reference gets: a fresh context, nothing set
            hctx = piquant.Context()
            served_by = hctx.host_path_in_effect()
            best, hq = host_rotation(hctx)
            extras["host_pointers_default"] = {"GiB/s": round(gib_per_step / best, 2), "ms_per_call": round(best * 1e3, 3), "served_by": served_by,
                                               "bit_equal_to_the_device_path": None,
                                               "note": "pageable host in/out through piquant_quantize with a default context (PIQUANT_HIP_HOST_PATH_AUTO): 'cpu' = handed whole to "
                                                       "libpiquant_cpu.so (AVX-512, one worker per physical core, unpinned; eight tensors in rotation = 1.1 GB that this thread "
                                                       "allocated and filled, nothing first-touched per worker: what an unprepared caller gets), 'stage' = no companion / no AVX-512: "
                                                       "PCIe staging; best mean per call over whole rotations"}
            # the bytes, against the HIP kernel on the same tensor (outside any timed region)
            dx = torch.from_numpy(xs0_host).to(dev)      # xs[0] itself has been an accumulator of the ADD measurement above
            dq = piquant.torch.quantize(dx, scale=scale, zero_point=zp, dtype=torch.uint8)
            torch.cuda.synchronize()
            extras["host_pointers_default"]["bit_equal_to_the_device_path"] = bool(np.array_equal(hq, dq.cpu().numpy()))
            del dx, dq
            ctx.set_stream(stream.cuda_stream)
            ctx.set_blocking(False)
        except Exception as exc:
            extras["host_pointers_default"] = {"error": repr(exc)}
        try:   # asked for: every element computed by the GPU
            hctx = piquant.Context()
            hctx.set_host_path("stage")
            hout = np.empty(n, dtype=np.uint8)
            hctx.quantize_ptr(xs0_host.ctypes.data, DataType.F32, hout.ctypes.data, DataType.UINT8, n, scale, zp, RoundMode.NEAREST)
            t0 = time.perf_counter()
            for _ in range(3):
                hctx.quantize_ptr(xs0_host.ctypes.data, DataType.F32, hout.ctypes.data, DataType.UINT8, n, scale, zp, RoundMode.NEAREST)
            th = (time.perf_counter() - t0) / 3
            extras["host_pointers_pcie_inclusive"] = {"GiB/s": round(gib_per_step / th, 2), "ms_per_call": round(th * 1e3, 3),
                                                      "note": "same call with piquant_hip_set_host_path(ctx, STAGE): pageable host in/out, chunked H2D -> HIP kernel -> D2H on two streams"}
        except Exception as exc:
            extras["host_pointers_pcie_inclusive"] = {"error": repr(exc)}
    for rec_ in extras.values():        # every side measurement that has an algorithmic rate also carries its fraction of the HBM peak
        if isinstance(rec_, dict) and "GB/s" in rec_:
            rec_["roofline_frac"] = round(rec_["GB/s"] / HBM_PEAK_GBS, 4)
    return extras
