#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native pi-quant hot path: the contract line and nothing else.

Metric (BASELINE.json): GiB/s of fp32 input quantized to uint8 (nearest rounding) on ONE 27 264 000-element tensor, plus the fraction of the
HBM roofline, on 1/2/4/8 GPUs.  A "step" is one quantization of that tensor: every rank makes ONE piquant_quantize call through the C ABI of
libpiquant.so (fp32 -> uint8, NEAREST) over ITS shard -- elements shard_range(numel, rank, N), the reference's pool split
(src/piquant.cpp:145-157) with GPUs in place of threads -- already resident in its HBM.  quantize needs no collective; (scale, zero_point) are
the tensor's global parameters (sharded min/max scan + one 8-byte MIN all-reduce, once, before the timed region).  Total work is fixed as N
grows -> STRONG scaling; value = the tensor's fp32 GiB x K / max-over-ranks time.  Steps rotate over 24 distinct buffer sets (3.3 GB per GPU at
every N: the 256 MiB Infinity Cache holds none of it, outputs included), so the number is an HBM number.

This file: the timed region, `roofline`, the self-check against the checker, `cpu_baseline` (tools/bench_cpu_baseline.py, rank 0 at N = 1).
Everything else -- graph replay, the other operators, configs 3-5, all-reduce schedules, weak scaling -- is tools/bench_extras.py, imported only
when extras are on and unable to cost the line.  N = 1: on by default.  N > 1 (round 6): the default is the line, its N = 1 reference point and
config 5 (the one path with a collective) -- nothing that has never run between two GPUs stands near the first scaling run; `--extras` adds the
rest (a watchdog prints the headline with whatever has finished, and the peer-to-peer child job starts only AFTER the line is out).
`--no-extras` prints the bare headline in a few seconds.

Launch: python bench.py [--gpus N]     (N > 1 without a launcher environment: bench.py starts its own N ranks through torch.distributed.run)
        python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W
Prints ONE JSON line on rank 0; `ranks_seen`, `shard_bit_exact` and (N > 1) `n1_reference` let the line validate itself.
"""
import argparse
import json
import os
import sys
import threading
import time
from pathlib import Path
from types import SimpleNamespace

ROOT = Path(__file__).resolve().parent
for _p in (str(ROOT), str(ROOT / "pi-quant_amd"), str(ROOT / "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

NUMEL = 27_264_000
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
ALGO_BYTES_PER_ELEM = 5        # 4 B read + 1 B written
PREWARM = 2000                 # untimed launches (~45 ms at N=1) so that short K/W runs are not measured on ramping clocks
EXTRAS_LIMIT_S = float(os.environ.get("PIQUANT_BENCH_EXTRAS_LIMIT_S", "240"))   # N > 1: the side measurements get this long before the headline is printed without them


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--numel", type=int, default=NUMEL)
    ap.add_argument("--windows", type=int, default=25, help="timed windows of --steps steps each; `value` is the median window")
    ap.add_argument("--sets", type=int, default=24, help="distinct buffer sets rotated through at N=1 (24 x 136 MB = 3.3 GB); N>1 keeps the same bytes per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="the bare contract line: no side measurements, no CPU baseline")
    ap.add_argument("--extras", action="store_true", default=os.environ.get("PIQUANT_BENCH_EXTRAS") == "1",
                    help="N > 1: ALL side measurements (all-reduce schedules, hipGraph replay, weak scaling, the peer-to-peer child job); the default for N > 1 is the "
                         "contract line, its N = 1 reference point and config 5 (sharded compute_quant_params with its 8-byte RCCL all-reduce) only")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU baseline time budget")
    # test plumbing: the N > 1 control flow on a box with ONE GPU (all ranks on cuda:0, gloo instead of RCCL, which refuses two ranks on one device)
    ap.add_argument("--backend", default="nccl", help=argparse.SUPPRESS)
    ap.add_argument("--share-gpu", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


def time_loop(fn, steps, stream, base=0):
    """Enqueue `steps` calls fn(base) .. fn(base + steps - 1) on `stream`; returns (wall seconds, HIP-event seconds)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record(stream)
    for i in range(base, base + steps):
        fn(i)
    e1.record(stream)
    while not e1.query():      # busy-wait for the last step: a sleeping hipDeviceSynchronize wakes up tens of microseconds late,
        pass                   # which is visible when K is small (20 steps are 440 us of work)
    t1 = time.perf_counter()   # every timed step ran on `stream` before e1: the work is complete here
    torch.cuda.synchronize()
    return t1 - t0, e0.elapsed_time(e1) * 1e-3


def shard_check(x_dev, out_dev, scale, zp):
    """This rank's output bytes against the checker -- the repository's C restatement of the reference arithmetic (oracle/, test
    infrastructure; never on the product path and never inside a timed region) run on the same shard on the host."""
    try:
        import oracle as O

        return bool(np.array_equal(out_dev.cpu().numpy(), O.quantize(x_dev.cpu().numpy(), O.F32, O.UINT8, scale, zp)))
    except Exception as exc:      # a box without the prebuilt checker: say so, do not claim
        print(f"bench.py: shard check unavailable: {exc!r}", file=sys.stderr, flush=True)
        return None


def device_identity(dev):
    props = torch.cuda.get_device_properties(dev)
    ident = getattr(props, "uuid", None)
    if ident is None:
        ident = f"pci {getattr(props, 'pci_domain_id', 0):04x}:{getattr(props, 'pci_bus_id', -1):02x}:{getattr(props, 'pci_device_id', -1):02x}"
    return f"cuda:{dev.index} {props.name} {ident}"


def median(sorted_values):
    m = len(sorted_values) // 2
    return sorted_values[m] if len(sorted_values) % 2 else 0.5 * (sorted_values[m - 1] + sorted_values[m])


def launch_own_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks here -- the same torch.distributed.run command the
    driver uses, rendezvous on 127.0.0.1 and a port that is free right now -- and pass their exit code on."""
    import socket
    import subprocess

    if not args.share_gpu and args.gpus > torch.cuda.device_count():      # loud and at once: RCCL with two ranks on one device does not fail, it hangs
        sys.exit(f"bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} visible device(s); one rank per GPU "
                 "(--share-gpu with --backend gloo is test plumbing for one-GPU boxes)")
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # the host driver only supports dmabuf IPC (RCCL between processes needs it)
    env.setdefault("OMP_NUM_THREADS", "8")                # torch.distributed.run would set 1; the CPU legs size their own pools
    env["PIQUANT_BENCH_SELF_LAUNCHED"] = "1"
    print(f"bench.py: --gpus {args.gpus} without a launcher environment: starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    sys.stdout.flush()
    sys.exit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ      # launched by torch.distributed.run (any N, also N=1)
    if args.gpus > 1 and not use_dist:
        launch_own_ranks(args)
    # The contract is ONE JSON line on stdout.  Native libraries (RCCL prints a version banner at communicator creation) write to fd 1 behind
    # Python's back, so everything but the final line is diverted to stderr at the fd level.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world, rank, local_rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a ROCm GPU: the product has no CPU path"
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher environment says WORLD_SIZE={world}: pass --gpus {world}, or run without a launcher")
    if args.share_gpu:
        local_rank = 0
    elif world > torch.cuda.device_count():
        sys.exit(f"bench.py: WORLD_SIZE={world} but only {torch.cuda.device_count()} visible device(s); one rank per GPU "
                 "(--share-gpu with --backend gloo is test plumbing for one-GPU boxes)")
    torch.cuda.set_device(local_rank if use_dist else 0)
    if use_dist:
        dist.init_process_group(args.backend, **({"device_id": torch.device("cuda", local_rank)} if args.backend == "nccl" else {}))
    dev = torch.device("cuda", torch.cuda.current_device())

    import piquant
    import piquant.distributed as pqd
    from piquant import DataType, RoundMode
    from piquant._bootstrap import C_LIB

    n_total = args.numel
    b0, e0 = pqd.shard_range(n_total, rank, world, 8)      # this rank's shard of the ONE logical tensor (src/piquant.cpp:145-157, ranks for threads)
    n = e0 - b0
    ctx = piquant.Context()
    stream = torch.cuda.Stream()

    def rearm():
        ctx.set_stream(stream.cuda_stream)
        ctx.set_blocking(False)

    rearm()
    # Rotating buffer sets: the same bytes per GPU at every N -- shards shrink with N, the number of sets grows.  Logical tensor s is
    # x_s ~ U(-1,1) fp32; a rank generates only its shard of it (seeded per rank and set).
    nsets = args.sets if world == 1 else max(args.sets, -(-args.sets * n_total // max(n, 1)))
    xs, outs = [], []
    for s in range(nsets):
        g = torch.Generator(device=dev)
        g.manual_seed(1000 * rank + s)
        xs.append(torch.empty(n, dtype=torch.float32, device=dev).uniform_(-1.0, 1.0, generator=g))
        outs.append(torch.empty(n, dtype=torch.uint8, device=dev))
    torch.cuda.synchronize()
    # global parameters of tensor 0: local scan + ONE 8-byte all_reduce(MIN) + epilogue (identical on every rank); world 1: the plain call
    scale, zp = (piquant.torch if world == 1 else pqd).compute_quant_params(xs[0], dtype=torch.quint8)
    torch.cuda.synchronize()
    rearm()
    want_cpu = rank == 0 and world == 1 and not (args.no_cpu_baseline or args.no_extras)
    xs0_host = xs[0].cpu().numpy() if want_cpu else None
    ptr_in, ptr_out = [t.data_ptr() for t in xs], [t.data_ptr() for t in outs]

    # One step = one piquant_quantize call through the C ABI, made through ctypes with its nine arguments built once per buffer set: 3.9 us of
    # host time per call instead of the 4.7 us of Context.quantize_ptr (enum lookups, asserts), which matters at N = 8, where a shard is a
    # 4.9 us kernel (profiles/r02_host_call_cost.json).  The context is stream-ordered, non-blocking and in device-pointer mode.
    ctx.assume_device_pointers(True)
    # N = 1: the plain piquant_quantize -- its bytes are those of the reference context `ctx` stands for (cpu_count - 1 pool threads, the reference's
    # Python default).  N > 1: a rank's call covers a SHARD of the tensor, and ranks are not pool threads: the position-independent twin.
    c_quantize = C_LIB.piquant_quantize if world == 1 else C_LIB.piquant_hip_quantize_uniform
    call_args = [(ctx._ctx, ptr_in[k], DataType.F32.value, ptr_out[k], DataType.UINT8.value, n, scale, zp, RoundMode.NEAREST.value) for k in range(nsets)]

    def step(i):
        c_quantize(*call_args[i % nsets])

    if use_dist:
        dist.barrier()     # the first collective sets the communicator up (hundreds of ms with RCCL): here, not next to the timed region
    with torch.cuda.stream(stream):
        for i in range(PREWARM + args.warmup):
            step(i)
        torch.cuda.synchronize()
        # The timed region: WINDOWS consecutive windows of EXACTLY K steps, every window bracketed by barrier + torch.cuda.synchronize() on both
        # sides (time_loop synchronizes at its start and end).  One window is what the contract describes; K = 20 steps are 0.45 ms of work, and
        # a single 0.45 ms window carries whatever the host happened to do in it.  `value` is the MEDIAN window -- max over ranks per window
        # first -- with the fastest and slowest beside it; the rotation runs on across the windows, so every launch of every window is cold.
        walls, evs = [], []
        for w in range(args.windows):
            if use_dist:
                dist.barrier()
            wall, ev = time_loop(step, args.steps, stream, base=w * args.steps)
            walls.append(wall)
            evs.append(ev)
        if use_dist:
            dist.barrier()

    # Self-validation, outside the timed region and before any side measurement reuses the buffers: every rank compares the bytes the timed
    # calls left in output buffer 0 with the checker run on ITS shard of tensor 0, and says which device it ran on.
    mine = {"rank": rank, "shard": [b0, e0], "bit_exact": shard_check(xs[0], outs[0], scale, zp), "device": device_identity(dev)}
    per_rank = [mine]
    t = torch.tensor([walls, evs], dtype=torch.float64, device=dev)
    if use_dist:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    walls_max, evs_max = sorted(float(v) for v in t[0]), sorted(float(v) for v in t[1])
    wall_max, ev_max = median(walls_max), median(evs_max)

    gib_per_step = n_total * 4 / 2**30                              # one step quantizes the whole logical tensor (all shards)
    kernel_s = ev_max / args.steps                                   # average launch duration: HIP events on the launch stream (slowest rank), median window
    n_max = -(-n_total // world)                                     # the largest shard
    achieved = ALGO_BYTES_PER_ELEM * n_max / kernel_s / 1e9          # per-GPU HBM rate of the dominant kernel
    result = {
        "metric": "GiB/s quantize fp32->uint8 (numel=27.26M, nearest)", "value": round(gib_per_step * args.steps / wall_max, 2), "unit": "GiB/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(wall_max / args.steps * 1e3, 6),
        "timed_windows": {"count": args.windows, "steps_each": args.steps, "value_is": "median window (max over ranks per window)",
                          "value_min": round(gib_per_step * args.steps / walls_max[-1], 2), "value_max": round(gib_per_step * args.steps / walls_max[0], 2),
                          "value_from_events": round(gib_per_step * args.steps / ev_max, 2),
                          "ms_per_step_min": round(walls_max[0] / args.steps * 1e3, 6), "ms_per_step_max": round(walls_max[-1] / args.steps * 1e3, 6)},
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"BASELINE configs[1]: fp32->uint8 nearest-round on MI355X, ONE tensor of numel={n_total}" +
                        (" on one GPU" if world == 1 else f" sharded over {world} GPUs by the reference's range split (src/piquant.cpp:145-157), "
                                                          f"{n_max} elements per GPU, no collective in the timed region") +
                        f", inputs resident in HBM, {nsets} rotating buffer sets ({nsets * ALGO_BYTES_PER_ELEM * n / 1e6:.0f} MB per GPU) to defeat the 256 MiB Infinity Cache",
            "numel_total": n_total, "numel_per_gpu": n_max, "round_mode": "nearest", "scale": scale, "zero_point": zp,
            "api": ("piquant_quantize (C ABI, libpiquant.so; reference layout of a " + str(ctx._num_threads) + "-thread reference context, the default)" if world == 1 else
                    "piquant_hip_quantize_uniform (C ABI, libpiquant.so; piquant_quantize in the position-independent form shards need)") + ", stream-ordered, one call per GPU per step",
            "parallelism": f"dp{world} (one shard of the tensor per GPU, no collective)", "prewarm_launches": PREWARM, "buffer_sets": nsets,
        },
        "roofline": {
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
            "kernel": "pq::quantize_kernel<f32,u8,nearest>", "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ELEM * n_max,
            "avg_launch_us": round(kernel_s * 1e6, 3), "timing": "HIP events on the launch stream around the K timed launches / K, median of the timed windows",
            "rotation": f"{nsets} buffer sets = {nsets * ALGO_BYTES_PER_ELEM * n / 1e9:.2f} GB per GPU: cold, outputs included",
        },
        "shard_bit_exact": [r["bit_exact"] for r in per_rank],
        "self_check": {"what": "output bytes of buffer set 0 after the timed region == the checker (oracle/: C restatement of the reference arithmetic, pinned against "
                               "the reference's own kernels in tests/) on each rank's shard_range of tensor 0; null = checker not available on this box",
                       "shards": [r["shard"] for r in per_rank], "all_bit_exact": all(r["bit_exact"] is True for r in per_rank)},
        "ranks_seen": {"world_size": dist.get_world_size() if use_dist else 1, "backend": (dist.get_backend() if use_dist else None),
                       "launcher": ("bench.py's own torch.distributed.run" if os.environ.get("PIQUANT_BENCH_SELF_LAUNCHED") else "external") if use_dist else None,
                       "devices": [r["device"] for r in per_rank], "distinct_devices": len({r["device"] for r in per_rank})},
    }
    if world > 1:
        result["roofline"]["scope"] = (f"per GPU: each launch moves {ALGO_BYTES_PER_ELEM * n_max} algorithmic bytes; at {n_max} elements per GPU a launch is "
                                       "dominated by its fixed ~2.4 us dispatch ramp/drain (DESIGN.md section 4), so the per-GPU fraction falls with N")
        result["roofline"]["aggregate_GB/s"] = round(ALGO_BYTES_PER_ELEM * n_total / kernel_s / 1e9, 1)
    tr = ROOT / "profiles" / "hbm_traffic.json"
    if tr.exists() and world == 1 and n_total == NUMEL:      # the PMC passes were taken on the full-size launch
        try:
            rec = json.loads(tr.read_text()).get("quantize_f32_u8") or {}
            result["roofline"]["traffic"], result["roofline"]["traffic_source"] = rec.get("bytes_per_launch"), rec.get("source")
        except Exception:
            pass

    line_lock, line_printed = threading.Lock(), [False]

    def emit(res):
        with line_lock:
            if rank == 0 and not line_printed[0]:
                sys.stdout.flush()
                os.write(real_stdout, (json.dumps(res) + "\n").encode())
            line_printed[0] = True

    # From here on nothing may cost the headline.  Side measurements live in tools/bench_extras.py; with N > 1 they contain collectives, and a
    # collective that one rank never reaches hangs the others for RCCL's timeout: every rank arms a watchdog that prints the line with whatever
    # has finished (rank 0) and leaves the process.
    B = SimpleNamespace(args=args, ctx=ctx, stream=stream, dev=dev, rank=rank, world=world, use_dist=use_dist, n=n, n_total=n_total, nsets=nsets, scale=scale, zp=zp,
                        gib_per_step=gib_per_step, xs=xs, outs=outs, ptr_in=ptr_in, ptr_out=ptr_out, call_args=call_args, c_quantize=c_quantize, step=step,
                        time_loop=time_loop, shard_check=shard_check, xs0_host=xs0_host, result=result, side={})
    if not args.no_extras:
        watchdog = None
        try:
            import bench_extras

            if world > 1:
                def bail():
                    emit(dict(result, extras=dict(B.side, error=f"the multi-rank side measurements did not finish within {EXTRAS_LIMIT_S} s; headline and what had finished by then")))
                    os._exit(0)

                watchdog = threading.Timer(EXTRAS_LIMIT_S, bail)
                watchdog.daemon = True
                watchdog.start()
                result["extras"] = bench_extras.multi_rank(B)
            else:
                result["extras"] = bench_extras.single_gpu(B)
        except Exception as exc:      # a fault in a side measurement is a field of the line, never the loss of it
            result["extras"] = dict(B.side, error=repr(exc))
        if watchdog is not None:
            watchdog.cancel()
        rearm()
    if want_cpu:
        try:
            from bench_cpu_baseline import CPU_SETS, cpu_baseline

            result["cpu_baseline"] = cpu_baseline(xs0_host, scale, zp, args.cpu_seconds, CPU_SETS)
        except Exception as exc:   # the baseline is a reported figure, never a reason to lose the GPU measurement
            result["cpu_baseline"] = {"value": None, "unit": "GiB/s", "cores": 0, "kind": "port", "sample": f"failed: {exc!r}"}
    emit(result)
    # The peer-to-peer transport has never run between two GPUs: its child job starts only now, with the line already out (its record: stderr).
    if world > 1 and args.extras and not args.no_extras:
        try:
            cpu_group = dist.new_group(backend="gloo")      # everybody waits on the CPU, not in a collective kernel spinning on the GPUs the child measures on
            torch.cuda.synchronize()
            dist.barrier(group=cpu_group)
            if rank == 0:
                bench_extras.p2p_child(B)
            dist.barrier(group=cpu_group)
        except Exception as exc:
            print(f"bench.py: p2p child job not run: {exc!r}", file=sys.stderr, flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
