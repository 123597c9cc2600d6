#!/usr/bin/env python3
"""bench.py's `cpu_baseline` leg: fp32 -> uint8 nearest on the GPU box's HOST cores under the same rotating-buffer protocol as the GPU side.

Two implementations side by side: the reference's own kernel units (oracle/_ref, compiled from the reference sources by oracle/Makefile; kind
"reference") where the prebuilt objects are present, and this repository's AVX-512 restatement (libpiquant_cpu.so; kind "port"), always.
Test/measurement infrastructure: imported by bench.py only, after the timed region, on rank 0 at N = 1."""
import os
import time
from pathlib import Path

import numpy as np

CPU_SETS = 16                    # the host side keeps 2.2 GB in rotation: four times the 2 x 256 MB of L3 of the GPU box's two sockets (with 6 sets = 818 MB, pinned
                                 # workers that always meet the same partitions got a large part of their reads from their own CCD's L3: 1 300 GiB/s "from DRAM")


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
                     "static range split (the reference's partition rule, src/piquant.cpp:145-157) over a persistent pool of pinned, spin-waiting workers standing in for its "
                     "un-vendored thread pool (oracle/ref_driver.cpp: a call is dispatched and joined through two cache lines -- no mutex, no futex wake)")

    def place(self, x_host, threads, order, nsets):
        self.R.set_pinning(order[:threads])
        self.threads = threads
        return [self.R.partition_copy(x_host, np.empty_like(x_host), threads) for _ in range(nsets)]

    def quantize(self, xin, out, scale, zp):   # straight to the driver: no numpy checks inside the timed loop
        self.R.lib.ref_quantize(self.isa, xin.ctypes.data, self.O.F32, out.ctypes.data, self.O.UINT8, xin.size, scale, int(zp), self.O.NEAREST, 0.0, self.threads)

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
    counts = sorted({t for t in (1, 8, 16, 32, per_socket, (per_socket + physical) // 2, physical, ncpu) if 1 <= t <= ncpu})
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
    ref["GiB/s_named"] = {named[t]: ref["GiB/s_by_threads"][str(t)] for t in counts if t in named}
    ref["port"] = port
    return ref
