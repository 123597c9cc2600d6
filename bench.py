#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native pi-quant hot path.

Metric (BASELINE.json): GiB/s of fp32 input quantized to uint8 (nearest rounding) on ONE 27 264 000-element
tensor, plus the fraction of the HBM roofline, on 1/2/4/8 GPUs.

A "step" is one quantization of that tensor: every rank makes ONE piquant_quantize call through the C ABI of
libpiquant.so (fp32 -> uint8, NEAREST) over ITS shard of the tensor -- elements shard_range(numel, rank, N), the
reference's pool split (src/piquant.cpp:145-157) with GPUs in place of threads -- already resident in its HBM.
quantize needs no collective; (scale, zero_point) are the tensor's global parameters (sharded min/max scan + one 8-byte
MIN all-reduce, done once before the timed region).  Total work is fixed as N grows -> STRONG scaling;
value = the tensor's fp32 bytes x K / max-over-ranks time.  At N = 1 the shard is the whole tensor.  Steps rotate over
24 distinct buffer sets (3.3 GB per GPU at every N; round 1's 6 sets kept their six 27 MB output buffers in the 256 MiB Infinity Cache,
see ROUND1_SETS below): the number is an HBM number.  The weak-scaling variant (every rank its own 27 264 000-element tensor, the data-parallel
gradient case) is timed separately into extras.weak_scaling_own_tensor_per_gpu for N > 1.

Launch: python bench.py [--gpus N]            (N > 1 without a launcher environment: bench.py starts its own N ranks through
                                                torch.distributed.run on 127.0.0.1 and a free port, and hands their one line on)
        python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
               bench.py --gpus N --steps K --warmup W        (the driver's form for N > 1: used as it is)
Prints ONE JSON line on rank 0.  For N > 1 the line validates itself: `ranks_seen` (what the process group reports, and the devices behind
the ranks), `shard_bit_exact` (every rank's output bytes of tensor 0 against the checker on its shard_range, outside the timed region) and
`n1_reference` (rank 0 alone, the whole tensor, same protocol: the N = 1 point of the same run).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for _p in (str(ROOT), str(ROOT / "pi-quant_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

NUMEL = 27_264_000
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
ALGO_BYTES_PER_ELEM = 5        # 4 B read + 1 B written
DEFAULT_BLOCKING_WAIT = "kernel"   # the library's default (csrc/context.cpp kDefaultBlockingWait)
# Rotation that really is cold.  Round 1 rotated 6 sets (818 MB, SURVEY 8d asked for > 512 MB); measured in round 2 on the same box, same kernel:
# 21.65 us per launch with 6 sets, 22.71 with 12 (1.6 GB), 22.86 with 24 (3.3 GB) -- the six 27 MB OUTPUT buffers of the
# 6-set rotation (164 MB) stay in the 256 MiB Infinity Cache and absorb the stores (extras.cold_inputs_one_output_buffer: 24 cold inputs into ONE
# output buffer run at the 6-set rate).  The headline therefore rotates 24 sets, inputs and outputs; the 6-set figure is kept in extras for continuity.
ROUND1_SETS = 6
EXTRAS_LIMIT_S = float(os.environ.get("PIQUANT_BENCH_EXTRAS_LIMIT_S", "240"))   # N > 1: the side measurements (graph replay, config 5, weak scaling) get this long before the headline is printed without them
CPU_SETS = 16                    # the host side keeps 2.2 GB in rotation: four times the 2 x 256 MB of L3 of the GPU box's two sockets (with 6 sets = 818 MB, pinned
                                 # workers that always meet the same partitions got a large part of their reads from their own CCD's L3: 1 300 GiB/s "from DRAM")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--numel", type=int, default=NUMEL)
    ap.add_argument("--windows", type=int, default=25, help="timed windows of --steps steps each; `value` is the median window")
    ap.add_argument("--sets", type=int, default=24, help="distinct buffer sets rotated through at N=1 (24 x 136 MB = 3.3 GB, see COLD_SETS); N>1 keeps the same bytes per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU baseline time budget")
    # test plumbing: exercise the N > 1 control flow on a box with ONE GPU (all ranks on cuda:0, gloo instead of RCCL, which
    # refuses two ranks on one device); the numbers of such a run mean nothing
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


def host_cpu_order():
    """Logical CPUs this process may use, ordered so that the first T of them are the natural placement of T workers: one hardware
    thread per physical core first, socket by socket (T <= cores of one socket stays on one socket / NUMA node), and within a socket
    round-robin over the last-level-cache domains (on EPYC a CCD's link to memory is much narrower than the socket's DRAM: eight
    workers belong on eight CCDs, not on one); SMT siblings last.  Returns (order, cores_per_socket, physical_cores, sockets)."""
    allowed = sorted(os.sched_getaffinity(0))
    info = {}
    for c in allowed:
        base = Path(f"/sys/devices/system/cpu/cpu{c}")
        try:
            pkg = int((base / "topology" / "physical_package_id").read_text())
            core = int((base / "topology" / "core_id").read_text())
        except (OSError, ValueError):
            pkg, core = 0, c
        try:
            llc = (base / "cache" / "index3" / "shared_cpu_list").read_text().strip()
        except OSError:
            llc = "all"
        info[c] = (pkg, core, llc)
    first, later, seen = [], [], set()
    for c in allowed:
        if info[c][:2] in seen:
            later.append(c)
        else:
            seen.add(info[c][:2])
            first.append(c)
    sockets = sorted({info[c][0] for c in allowed})
    order = []
    for s_ in sockets:
        domains = {}
        for c in first:
            if info[c][0] == s_:
                domains.setdefault(info[c][2], []).append(c)
        queues = [domains[k] for k in sorted(domains, key=lambda k: domains[k][0])]
        while any(queues):
            for q in queues:
                if q:
                    order.append(q.pop(0))
    per_socket = max(sum(1 for c in first if info[c][0] == s_) for s_ in sockets)
    return order + later, per_socket, len(first), len(sockets)


class _RefBackend:
    """the reference's own kernel units (oracle/_ref, compiled from the reference sources by oracle/Makefile) behind the rotation protocol"""
    kind = "reference"

    def __init__(self):
        import oracle as O

        self.O, self.R = O, O.Ref()
        self.isa = self.R.best_isa()
        built_here = Path("/root/reference").exists()
        self.what = (f"reference {self.R.isa_name(self.isa)} kernels (oracle/_ref: the reference's kernel translation units compiled from its sources, "
                     f"{'built on this box' if built_here else 'shipped prebuilt with the repository snapshot -- /root/reference does not exist here'}), "
                     "static range split (the reference's partition rule, src/piquant.cpp:145-157) over a persistent std::thread pool standing in for its un-vendored thread pool")

    def place(self, x_host, threads, order, nsets):
        self.R.set_pinning(order[:threads])
        self.threads = threads
        return [self.R.partition_copy(x_host, np.empty_like(x_host), threads) for _ in range(nsets)]

    def quantize(self, xin, out, scale, zp):
        self.R.quantize(xin, self.O.F32, self.O.UINT8, scale, zp, isa=self.isa, threads=self.threads, out=out)

    def done(self):
        self.R.set_pinning([])


class _PortBackend:
    """libpiquant_cpu.so: this repository's own AVX-512 restatement (pi-quant_amd/csrc/cpu), reproducible from a clean checkout"""
    kind = "port"

    def __init__(self, max_threads):
        from piquant import cpu

        self.cpu = cpu
        self.ctx = cpu.CpuContext(max_threads)
        self.what = ("libpiquant_cpu.so, this repository's own " + ("AVX-512" if cpu.has_avx512() else "scalar (host without AVX-512)") +
                     " restatement of the path (pi-quant_amd/csrc/cpu; bit-equal to the reference's kernels, tests/test_cpu_path.py), "
                     "static range split (src/piquant.cpp:145-157) over its persistent pool")

    def place(self, x_host, threads, order, nsets):
        self.ctx.set_active_threads(threads)
        self.ctx.set_affinity(order[:threads])
        ins = []
        for _ in range(nsets):
            dst = np.empty_like(x_host)
            self.ctx.partition_copy_ptr(x_host.ctypes.data, dst.ctypes.data, 0, x_host.size)
            ins.append(dst)
        return ins

    def quantize(self, xin, out, scale, zp):
        self.ctx.quantize_ptr(xin.ctypes.data, 0, out.ctypes.data, 4, xin.size, scale, zp)

    def done(self):
        self.ctx.set_affinity([])
        self.ctx.close()


def _cpu_rotation(backend, x_host, scale, zp, budget_s, nsets, counts, order):
    """best mean-per-call over whole rotations through `nsets` buffer sets, for every thread count; NUMA-fair: workers pinned (one per
    physical core, socket by socket), buffers allocated fresh per count and every partition first touched by the worker that processes it"""
    n = x_host.size
    gib = n * 4 / 2**30
    per = budget_s / (len(counts) + 1)
    times = {}
    for t in counts:
        ins = backend.place(x_host, t, order, nsets)
        outs = [np.empty(n, dtype=np.uint8) for _ in range(nsets)]    # untouched: first written by the workers in the first rotation
        best, t_end, rounds = float("inf"), time.perf_counter() + per, 0
        while rounds < 3 or time.perf_counter() < t_end:
            t0 = time.perf_counter()
            for k in range(nsets):
                backend.quantize(ins[k], outs[k], scale, zp)
            if rounds > 0:           # the first rotation faults the output pages in
                best = min(best, (time.perf_counter() - t0) / nsets)
            rounds += 1
        times[t] = best
        del ins, outs
    best_t = min(times, key=times.get)
    ins = backend.place(x_host, best_t, order, 1)
    out = np.empty(n, dtype=np.uint8)
    hot, t_end = float("inf"), time.perf_counter() + per      # cache-resident variant: one buffer set, best single call
    while time.perf_counter() < t_end:
        t0 = time.perf_counter()
        backend.quantize(ins[0], out, scale, zp)
        hot = min(hot, time.perf_counter() - t0)
    backend.done()
    return {"value": round(gib / times[best_t], 3), "unit": "GiB/s", "cores": best_t, "kind": backend.kind, "ms_per_call": round(times[best_t] * 1e3, 4),
            "GiB/s_by_threads": {str(t): round(gib / v, 2) for t, v in times.items()}, "cache_resident_single_buffer_GiB/s": round(gib / hot, 2)}


def cpu_baseline(x_host: np.ndarray, scale: float, zp: int, budget_s: float, nsets: int):
    """fp32 -> uint8 nearest on this box's host cores, same protocol as the GPU side: calls rotate over `nsets` distinct input/output buffer
    sets (818 MB for 6 sets, more than the host's last-level cache) so the figure is a DRAM figure; the cache-resident single-buffer figure
    is reported separately.  Two implementations, same protocol, same thread counts: the reference's own kernels (oracle/_ref, when the
    prebuilt objects are present) and this repository's AVX-512 restatement (libpiquant_cpu.so, always: reproducible from a clean checkout).
    The headline entry is the reference's where available ("kind": "reference"), with the port beside it under "port"."""
    import oracle as O

    n = x_host.size
    order, per_socket, physical, sockets = host_cpu_order()
    ncpu = len(order)
    counts = sorted({t for t in (1, 8, 16, 32, per_socket, physical, ncpu) if 1 <= t <= ncpu})
    named = {1: "1 thread", per_socket: f"one socket ({per_socket} cores)", physical: f"all {physical} physical cores", ncpu: f"all {ncpu} hardware threads"}
    protocol = (f"fp32->uint8 nearest on the full {n}-element tensor, calls rotating over {nsets} buffer sets ({nsets * 5 * n / 1e6:.0f} MB, beyond the host LLC) "
                f"like the GPU side, best mean per call over whole rotations; numa: {sockets} socket(s) x {per_socket} cores, workers pinned one per physical core, "
                f"socket by socket and round-robin over the L3 domains within a socket (SMT siblings last), every buffer partition first touched by the worker "
                f"that processes it; host has {ncpu} usable hardware threads")
    have_ref = O.ref_available()
    port_counts = counts if not have_ref else sorted({t for t in (1, 32, per_socket, physical) if 1 <= t <= ncpu})
    pb = _PortBackend(ncpu)
    port = _cpu_rotation(pb, x_host, scale, zp, budget_s * (0.4 if have_ref else 1.0), nsets, port_counts, order)
    port["sample"] = f"{pb.what}; {protocol}; best at {port['cores']} threads"
    port["GiB/s_named"] = {named[t]: port["GiB/s_by_threads"][str(t)] for t in port_counts if t in named}
    if not have_ref:
        return port
    rb = _RefBackend()
    ref = _cpu_rotation(rb, x_host, scale, zp, budget_s * 0.6, nsets, counts, order)
    ref["sample"] = f"{rb.what}; {protocol}; best at {ref['cores']} threads"
    if ncpu > physical and str(ncpu) in ref["GiB/s_by_threads"]:
        ref["beyond_the_physical_cores"] = (f"{ncpu} threads = both hardware threads of every core: {ref['GiB/s_by_threads'][str(ncpu)]} GiB/s against "
                                            f"{ref['GiB/s_by_threads'][str(physical)]} on the {physical} physical cores -- a static range split ends with its slowest worker, SMT siblings share a "
                                            "core's load/store pipes, and the stand-in pool wakes its sleepers through a condition variable (milliseconds for 255 of them); the reference's "
                                            "own pool is not vendored, so the physical-core count is the last point that says something about its kernels")
    ref["GiB/s_named"] = {named[t]: ref["GiB/s_by_threads"][str(t)] for t in counts if t in named}
    ref["port"] = port
    return ref


def shard_check(x_dev, out_dev, scale, zp):
    """This rank's output bytes of buffer set 0 against the checker -- the repository's C restatement of the reference arithmetic
    (oracle/, test infrastructure; never on the product path and never inside a timed region) run on the same shard on the host."""
    try:
        import oracle as O

        want = O.quantize(x_dev.cpu().numpy(), O.F32, O.UINT8, scale, zp)
        return bool(np.array_equal(out_dev.cpu().numpy(), want))
    except Exception as exc:      # a box without the prebuilt checker: say so, do not claim
        print(f"bench.py: shard check unavailable: {exc!r}", file=sys.stderr, flush=True)
        return None


def device_identity(dev):
    props = torch.cuda.get_device_properties(dev)
    ident = getattr(props, "uuid", None)
    if ident is None:
        ident = f"pci {getattr(props, 'pci_domain_id', 0):04x}:{getattr(props, 'pci_bus_id', -1):02x}:{getattr(props, 'pci_device_id', -1):02x}"
    return f"cuda:{dev.index} {props.name} {ident}"


def gather_objects(obj, world, use_dist):
    if not use_dist:
        return [obj]
    got = [None] * world
    dist.all_gather_object(got, obj)
    return got


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

    if os.environ.get("PIQUANT_BENCH_P2P", "1") == "0":
        return "not run: PIQUANT_BENCH_P2P=0"
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


def launch_own_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment (RANK / MASTER_ADDR): start the N ranks here -- the same
    torch.distributed.run command the driver uses for N > 1, rendezvous on 127.0.0.1 (the container hostname may not resolve) and a port that is
    free right now -- and pass their exit code on.  Rank 0 of the children prints the one JSON line on the stdout they inherit."""
    import socket
    import subprocess

    if not args.share_gpu:
        seen = torch.cuda.device_count()
        if args.gpus > seen:      # loud and at once: RCCL with two ranks on one device does not fail, it hangs
            sys.exit(f"bench.py: --gpus {args.gpus} but only {seen} visible device(s); one rank per GPU "
                     "(--share-gpu with --backend gloo is test plumbing for one-GPU boxes)")
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # the host driver only supports dmabuf IPC (RCCL between processes needs it)
    env.setdefault("OMP_NUM_THREADS", "8")                # torch.distributed.run would set 1 (and say so on stderr); the CPU legs size their own pools
    env["PIQUANT_BENCH_SELF_LAUNCHED"] = "1"
    print(f"bench.py: --gpus {args.gpus} without a launcher environment: starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    sys.stdout.flush()
    sys.exit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if args.gpus > 1 and not ("RANK" in os.environ and "MASTER_ADDR" in os.environ):
        launch_own_ranks(args)
    # The contract is ONE JSON line on stdout.  Native libraries (RCCL prints a version banner at communicator creation)
    # write to fd 1 behind Python's back, so everything but the final line is diverted to stderr at the fd level.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a ROCm GPU: the product has no CPU path"
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ      # launched by torch.distributed.run (any N, also N=1)
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher environment says WORLD_SIZE={world}: pass --gpus {world}, or run without a launcher "
                 "(bench.py starts its own ranks)")
    if args.share_gpu:
        local_rank = 0
    elif world > torch.cuda.device_count():
        # fail at once and loudly: RCCL with two ranks on one device does not fail, it hangs
        sys.exit(f"bench.py: WORLD_SIZE={world} but only {torch.cuda.device_count()} visible device(s); one rank per GPU "
                 "(--share-gpu with --backend gloo is test plumbing for one-GPU boxes)")
    torch.cuda.set_device(local_rank if use_dist else 0)
    if use_dist:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    dev = torch.device("cuda", torch.cuda.current_device())

    import piquant
    from piquant import DataType, RoundMode

    import piquant.distributed as pqd

    n_total = args.numel
    # This rank's shard of the ONE logical tensor (the reference's split rule, src/piquant.cpp:145-157, ranks for threads).
    b0, e0 = pqd.shard_range(n_total, rank, world, 8)
    n = e0 - b0
    ctx = piquant.Context()
    stream = torch.cuda.Stream()
    ctx.set_stream(stream.cuda_stream)
    ctx.set_blocking(False)

    # Rotating buffer sets: the same bytes per GPU at every N (818 MB with the default 6 sets at N = 1), so that the 256 MiB
    # Infinity Cache never holds the working set -- shards shrink with N, the number of sets grows.
    nsets = args.sets if world == 1 else max(args.sets, -(-args.sets * n_total // max(n, 1)))
    # synthetic data: logical tensor s is x_s ~ U(-1,1) fp32; a rank generates only its shard of it (seeded per rank and set)
    xs, outs = [], []
    for s in range(nsets):
        g = torch.Generator(device=dev)
        g.manual_seed(1000 * rank + s)
        xs.append(torch.empty(n, dtype=torch.float32, device=dev).uniform_(-1.0, 1.0, generator=g))
        outs.append(torch.empty(n, dtype=torch.uint8, device=dev))
    torch.cuda.synchronize()
    # global parameters of tensor 0: local scan + ONE 8-byte all_reduce(MIN) + epilogue (identical on every rank); world 1: the plain call
    if world == 1:
        scale, zp = piquant.torch.compute_quant_params(xs[0], dtype=torch.quint8)
    else:
        scale, zp = pqd.compute_quant_params(xs[0], dtype=torch.quint8)
    torch.cuda.synchronize()
    ctx.set_stream(stream.cuda_stream)
    ctx.set_blocking(False)

    xs0_host = xs[0].cpu().numpy() if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    ptr_in = [t.data_ptr() for t in xs]
    ptr_out = [t.data_ptr() for t in outs]

    # One step = one piquant_quantize call through the C ABI.  The call is made through ctypes with its nine arguments built once per
    # buffer set: 3.9 us of host time per call instead of the 4.7 us of Context.quantize_ptr (enum lookups, asserts), which matters at
    # N = 8, where a 3.4 M-element shard is a 4.9 us kernel and a slower host would leave the queue empty between launches
    # (profiles/r02_host_call_cost.json).  The context is already stream-ordered, non-blocking and in device-pointer mode.
    from piquant._bootstrap import C_LIB

    ctx.assume_device_pointers(True)
    c_quantize = C_LIB.piquant_quantize
    call_args = [(ctx._ctx, ptr_in[k], DataType.F32.value, ptr_out[k], DataType.UINT8.value, n, scale, zp, RoundMode.NEAREST.value) for k in range(nsets)]

    def step(i):
        c_quantize(*call_args[i % nsets])

    PREWARM = 2000
    if use_dist:
        # The first collective of a process group sets the communicator up (hundreds of milliseconds with RCCL).  Done here, the barrier next
        # to the timed region is a ~30 us affair; done there, the GPU would sit idle long enough to drop its clocks and the K timed steps
        # (0.45 ms at K = 20) would run on the ramp: measured with a one-rank RCCL group, 23.6 us per launch instead of 22.1.
        dist.barrier()
    with torch.cuda.stream(stream):
        for i in range(PREWARM):         # untimed pre-warm (~45 ms at N=1) so short K/W runs are not measured on ramping clocks;
            step(i)                      # reported as config.prewarm_launches
        for i in range(args.warmup):
            step(i)
        torch.cuda.synchronize()
        # The timed region: WINDOWS consecutive windows of EXACTLY K steps, every window bracketed by barrier + torch.cuda.synchronize() on both
        # sides (time_loop synchronizes at its start and end).  One window is what the contract describes; K = 20 steps are 0.45 ms of work, and a
        # single 0.45 ms window carries whatever the host happened to do in it (round 2: 4 146 GiB/s at the driver against 4 440 in the
        # builder's runs of the same command, same kernel time).  `value` is the MEDIAN window -- max over ranks per window first -- with the
        # fastest and slowest beside it; the buffer rotation runs on across the windows, so every launch of every window is cold.
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
    per_rank = gather_objects(mine, world, use_dist)

    t = torch.tensor([walls, evs], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    walls_max = sorted(float(v) for v in t[0])
    evs_max = sorted(float(v) for v in t[1])
    wall_max = walls_max[len(walls_max) // 2] if len(walls_max) % 2 else 0.5 * (walls_max[len(walls_max) // 2 - 1] + walls_max[len(walls_max) // 2])
    ev_max = evs_max[len(evs_max) // 2] if len(evs_max) % 2 else 0.5 * (evs_max[len(evs_max) // 2 - 1] + evs_max[len(evs_max) // 2])

    gib_per_step = n_total * 4 / 2**30                              # one step quantizes the whole logical tensor (all shards)
    value = gib_per_step * args.steps / wall_max
    kernel_s = ev_max / args.steps                                   # average launch duration from HIP events on the launch stream (slowest rank), median window
    n_max = -(-n_total // world)                                     # the largest shard
    achieved = ALGO_BYTES_PER_ELEM * n_max / kernel_s / 1e9          # per-GPU HBM rate of the dominant kernel

    result = {
        "metric": "GiB/s quantize fp32->uint8 (numel=27.26M, nearest)",
        "value": round(value, 2),
        "unit": "GiB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(wall_max / args.steps * 1e3, 6),
        "timed_windows": {"count": args.windows, "steps_each": args.steps, "value_is": "median window (max over ranks per window)",
                          "value_min": round(gib_per_step * args.steps / walls_max[-1], 2), "value_max": round(gib_per_step * args.steps / walls_max[0], 2),
                          "value_from_events": round(gib_per_step * args.steps / ev_max, 2),
                          "ms_per_step_min": round(walls_max[0] / args.steps * 1e3, 6), "ms_per_step_max": round(walls_max[-1] / args.steps * 1e3, 6)},
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"BASELINE configs[1]: fp32->uint8 nearest-round on MI355X, ONE tensor of numel={n_total}" +
                        (" on one GPU" if world == 1 else f" sharded over {world} GPUs by the reference's range split (src/piquant.cpp:145-157), "
                                                          f"{n_max} elements per GPU, no collective in the timed region") +
                        f", inputs resident in HBM, {nsets} rotating buffer sets ({nsets * ALGO_BYTES_PER_ELEM * n / 1e6:.0f} MB per GPU) to defeat the "
                        "256 MiB Infinity Cache",
            "numel_total": n_total, "numel_per_gpu": n_max, "round_mode": "nearest", "scale": scale, "zero_point": zp,
            "api": "piquant_quantize (C ABI, libpiquant.so), stream-ordered, one call per GPU per step",
            "parallelism": f"dp{world} (one shard of the tensor per GPU, no collective)",
            "prewarm_launches": PREWARM, "buffer_sets": nsets,
        },
        "roofline": {
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": None,
            "kernel": "pq::quantize_kernel<f32,u8,nearest>", "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ELEM * n_max,
            "avg_launch_us": round(kernel_s * 1e6, 3), "timing": "HIP events on the launch stream around the K timed launches / K, median of the timed windows",
            "rotation": f"{nsets} buffer sets = {nsets * ALGO_BYTES_PER_ELEM * n / 1e9:.2f} GB per GPU: cold (round 1 rotated 6 sets = 818 MB, whose six 27 MB output buffers stay in the 256 MiB "
                        "Infinity Cache: ~1.1 us per launch faster; that figure: extras.rotation_of_6_sets_818MB_round1_protocol, and extras.cold_inputs_one_output_buffer)",
        },
    }
    result["shard_bit_exact"] = [r["bit_exact"] for r in per_rank]
    result["self_check"] = {"what": "output bytes of buffer set 0 after the timed region == the checker (oracle/: C restatement of the reference arithmetic, pinned against "
                                    "the reference's own kernels in tests/) on each rank's shard_range of tensor 0; null = checker not available on this box",
                            "shards": [r["shard"] for r in per_rank], "all_bit_exact": all(r["bit_exact"] is True for r in per_rank)}
    result["ranks_seen"] = {"world_size": dist.get_world_size() if use_dist else 1, "backend": (dist.get_backend() if use_dist else None),
                            "launcher": ("bench.py's own torch.distributed.run" if os.environ.get("PIQUANT_BENCH_SELF_LAUNCHED") else "external") if use_dist else None,
                            "devices": [r["device"] for r in per_rank], "distinct_devices": len({r["device"] for r in per_rank})}
    if world > 1:
        result["roofline"]["scope"] = (f"per GPU: each launch moves {ALGO_BYTES_PER_ELEM * n_max} algorithmic bytes; at {n_max} elements per GPU a launch is "
                                       "dominated by its fixed ~2.4 us dispatch ramp/drain (DESIGN.md section 4), so the per-GPU fraction falls with N")
        result["roofline"]["aggregate_GB/s"] = round(ALGO_BYTES_PER_ELEM * n_total / kernel_s / 1e9, 1)

    tr = ROOT / "profiles" / "hbm_traffic.json"
    if tr.exists() and world == 1 and n_total == NUMEL:      # the PMC passes were taken on the full-size launch
        try:
            rec = json.loads(tr.read_text()).get("quantize_f32_u8")
            if rec:
                result["roofline"]["traffic"] = rec.get("bytes_per_launch")
                result["roofline"]["traffic_source"] = rec.get("source")
        except Exception:
            pass

    # The same K steps replayed from a hipGraph (every stream-ordered call of the library is capturable): what is left of a step when the host's
    # per-launch work is taken out of it.  At N = 1 that is little (a launch is 4 us of host time behind a 22.7 us kernel); at N = 8 a shard is a
    # 5 us kernel and the host, not the GPU, sets the pace of directly issued steps.  Runs on every rank (barriers); extras, never `value`.
    # From here on nothing may cost the headline.  With N > 1 the side measurements below contain collectives, and a collective that one rank
    # never reaches (an exception on that rank only) hangs the others for RCCL's ten-minute timeout: every rank arms a watchdog that, when the
    # side measurements overrun, prints the line without them (rank 0) and leaves the process.
    import threading

    line_lock = threading.Lock()
    line_printed = [False]

    def emit(res):
        with line_lock:
            if rank == 0 and not line_printed[0]:
                sys.stdout.flush()
                os.write(real_stdout, (json.dumps(res) + "\n").encode())
            line_printed[0] = True

    watchdog = None
    side = {}      # N > 1: side measurements as they finish (what the watchdog's line carries)
    if world > 1 and not args.no_extras:
        def bail():
            headline = dict(result)
            headline["extras"] = dict(side, error=f"the multi-rank side measurements did not finish within {EXTRAS_LIMIT_S} s; headline and what had finished by then")
            emit(headline)
            os._exit(0)

        watchdog = threading.Timer(EXTRAS_LIMIT_S, bail)
        watchdog.daemon = True
        watchdog.start()

    # N > 1: the N = 1 point of the SAME run -- rank 0 alone quantizes the whole tensor with the headline's protocol (K steps per window, cold
    # rotation of args.sets full-size sets) while the other ranks wait at the barrier behind it.
    n1_ref = None
    if world > 1 and not args.no_extras:
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
                          "bit_exact": shard_check(rx[0], ro[0], scale, zp), "windows": len(rw),
                          "note": f"rank 0 alone, the whole {n_total}-element tensor on one GPU, {args.sets} cold buffer sets, median window of K = {args.steps} steps: "
                                  "the N = 1 point measured inside this N > 1 run (compare with the driver's N = 1 line)"}
                del rx, ro
        except Exception as exc:
            n1_ref = {"error": repr(exc)}
        result["n1_reference"] = n1_ref      # in the line even if a later side measurement runs into the watchdog
        dist.barrier()

    all_reduce, native5 = None, None
    if world > 1 and not args.no_extras:
        try:
            all_reduce, _x = all_reduce_extras(args, pqd, dev, rank, world, n_total)
            del _x
        except Exception as exc:
            all_reduce = {"error": repr(exc)}
        # the peer-to-peer transport, as a child job (rank 0 starts it; everybody waits on the CPU -- a gloo barrier, not a collective kernel
        # spinning on the GPUs the child measures on)
        try:
            cpu_group = dist.new_group(backend="gloo")
            torch.cuda.synchronize()
            dist.barrier(group=cpu_group)
            p2p = p2p_all_reduce_child_job(args, world) if rank == 0 else None
            dist.barrier(group=cpu_group)
            if isinstance(all_reduce, dict) and rank == 0:
                all_reduce["p2p_transport_child_job"] = p2p
        except Exception as exc:
            if isinstance(all_reduce, dict):
                all_reduce["p2p_transport_child_job"] = {"error": repr(exc)}
        side["all_reduce_109MB"] = all_reduce
        ctx.set_stream(stream.cuda_stream)
        ctx.set_blocking(False)

    graphed = None
    if not args.no_extras:
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
        side["steps_replayed_from_a_hipgraph"] = graphed

    # Independent calls issued alternately on two streams (a context each): a stream runs its kernels one after the other, and the ~2 us in which
    # a launch ramps up and drains move no bytes; on two streams the next tensor's ramp runs under this one's drain.  What a
    # caller with many tensors and no order between them can have; extras, never `value` (whose steps share ONE stream, as a plain caller's do).
    two_streams = None
    if not args.no_extras and world == 1:
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

    # BASELINE configs[4]: compute_quant_params over a 2^30-element fp32 tensor sharded across the ranks -- every rank scans
    # its shard in HBM, ONE 8-byte all_reduce(MIN) over RCCL/xGMI, identical double-precision epilogue everywhere.  Runs on
    # every rank (it contains the collective); reported next to the headline, not as `value`.
    config5 = None
    if not args.no_extras:
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
            config5 = {"numel_total": total5, "numel_per_gpu": e5 - b5, "ms_per_call": round(float(t5t[0]) * 1e3, 5),
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
                config5["ms_per_call_without_collective"] = round(tl * 1e3, 5)
                config5["collective_adds_ms"] = round((float(t5t[0]) - tl) * 1e3, 5)
                if args.backend == "nccl":
                    try:
                        ctx.set_blocking(True)
                        native5 = native_dist_entry(args, ctx, shard, dev, rank, world, want5)
                    except Exception as exc:
                        native5 = {"error": repr(exc)}
                    config5["native_entry_piquant_hip_compute_quant_params_dist"] = native5
            del shard
            ctx.set_stream(stream.cuda_stream)
            ctx.set_blocking(False)
        except Exception as exc:   # never lose the headline line to the secondary measurement
            config5 = {"error": repr(exc)}
            ctx.set_stream(stream.cuda_stream)
            ctx.set_blocking(False)
        side["config5_sharded_compute_quant_params"] = config5

    # N > 1: the weak-scaling variant next to the strong-scaling headline -- every rank quantizes its OWN full-size tensor (the
    # data-parallel gradient case), same protocol; runs on every rank, reported under extras.
    weak = None
    if not args.no_extras and world > 1:
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

    if rank == 0 and not args.no_extras and world > 1:
        result["extras"] = {"steps_replayed_from_a_hipgraph": graphed, "config5_sharded_compute_quant_params": config5, "weak_scaling_own_tensor_per_gpu": weak,
                            "all_reduce_109MB": all_reduce}
    if rank == 0 and not args.no_extras and world == 1:     # the single-GPU side measurements stay out of the multi-rank runs
        extras = {"steps_replayed_from_a_hipgraph": graphed, "independent_calls_on_two_streams": two_streams, "config5_sharded_compute_quant_params": config5}

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
        result["extras"] = extras

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            result["cpu_baseline"] = cpu_baseline(xs0_host, scale, zp, args.cpu_seconds, CPU_SETS)
        except Exception as exc:   # the baseline is a reported figure, never a reason to lose the GPU measurement
            result["cpu_baseline"] = {"value": None, "unit": "GiB/s", "cores": 0, "kind": "port", "sample": f"failed: {exc!r}"}

    if watchdog is not None:
        watchdog.cancel()
    emit(result)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
