"""GPU: the fused kernel's grid barrier when its grid is NOT fully resident (VERDICT r01 "co-residency is assumed, not guaranteed").

The one-launch params+quantize kernel waits at a grid barrier.  These tests put it where a plain launch gives no guarantee --
other kernels holding CUs, two barrier kernels launched at the same moment from two threads -- and where round 1 spun until a
trap: it must finish, and its bytes and parameter record must equal the two-launch path's (which equals the oracle, test_gpu_parity).
"""
import ctypes
import subprocess
import threading
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent
N1 = 27_264_000


@pytest.fixture(scope="module")
def hog(tmp_path_factory):
    so = tmp_path_factory.mktemp("hog") / "libcuhog.so"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", str(ROOT / "tests" / "cu_hog.hip"), "-o", str(so)], check=True)
    import piquant  # noqa: F401  (torch + libpiquant first: one HIP runtime in the process)

    lib = ctypes.CDLL(str(so))
    lib.cu_hog_launch.restype = ctypes.c_int
    lib.cu_hog_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_void_p]
    return lib


def _reference(ctx, x, qdtype):
    """two launches (scan with the parameter epilogue, then quantize): no grid barrier involved"""
    import piquant

    ctx.set_fusion(False)
    q, rec = piquant.torch.quantize_dynamic(x, dtype=qdtype, ctx=ctx)
    torch.cuda.synchronize()
    ctx.set_fusion(True)
    return piquant.torch.packed_bytes(q).clone(), rec.clone()


@pytest.mark.parametrize("fdtype,qname", [(torch.float32, "uint8"), (torch.bfloat16, "quint4x2"), (torch.float32, "quint2x4")])
def test_barrier_timeout_of_one_microsecond_still_gives_the_right_bytes(fdtype, qname):
    """With a 1 us limit every block that arrives more than 1 us before the last one leaves, and the few blocks still there adopt
    their shares from HBM: the RACY form of the hand-over (marks taken back when the barrier opens meanwhile).  Same bytes, same
    record.  Whether the limit fires is up to the GPU's scheduling and is only reported -- the path itself is pinned, counter and
    all, by test_hand_over_always_* below (round 4: an assertion on this counter stopped the driver's run)."""
    import piquant

    ctx = piquant.Context()
    qdtype = getattr(torch, qname)
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    for n in (N1, N1 + 5, 3_000_001):
        x = torch.empty(n, device="cuda").uniform_(-1, 1, generator=g).to(fdtype)
        want_q, want_rec = _reference(ctx, x, qdtype)
        before = ctx.barrier_bailouts()
        ctx.set_barrier_timeout_us(1)
        for _ in range(3):
            q, rec = piquant.torch.quantize_dynamic(x, dtype=qdtype, ctx=ctx)
        torch.cuda.synchronize()
        ctx.set_barrier_timeout_us(0)
        assert torch.equal(piquant.torch.packed_bytes(q), want_q) and torch.equal(rec, want_rec), (n, qname)
        print(f"n={n} {qname}: {ctx.barrier_bailouts() - before} hand-overs under the 1 us limit")
        # and the state is fit for ordinary launches again
        q2, rec2 = piquant.torch.quantize_dynamic(x, dtype=qdtype, ctx=ctx)
        torch.cuda.synchronize()
        assert torch.equal(piquant.torch.packed_bytes(q2), want_q) and torch.equal(rec2, want_rec)


def _blocks_handed_over(ctx, launch, launches=3):
    """runs `launch` `launches` times with every block but the last of each tensor handing over; returns (last result, hand-overs per launch).
    Deterministic: the count must be the same every time and at least one."""
    ctx.set_barrier_timeout_us(ctx.HAND_OVER_ALWAYS)
    per_launch = []
    out = None
    for _ in range(launches):
        before = ctx.barrier_bailouts()
        out = launch()
        torch.cuda.synchronize()
        per_launch.append(ctx.barrier_bailouts() - before)
    ctx.set_barrier_timeout_us(0)
    assert per_launch[0] >= 1 and len(set(per_launch)) == 1, per_launch
    return out, per_launch[0]


@pytest.mark.parametrize("fdtype,qname", [(torch.float32, "uint8"), (torch.bfloat16, "quint4x2"), (torch.float32, "quint2x4")])
def test_hand_over_always_gives_the_right_bytes(fdtype, qname):
    """The deterministic form: 255 of 256 blocks mark their share and leave, the last one adopts everything (block 0's duties -- the
    record, the generation word -- and the ragged tail included).  Bytes and record equal the two-launch path's, the counter moves by
    exactly blocks - 1 per launch, and the state serves ordinary launches afterwards."""
    import piquant

    ctx = piquant.Context()
    qdtype = getattr(torch, qname)
    g = torch.Generator(device="cuda")
    g.manual_seed(13)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    for n in (N1, N1 + 5, 3_000_001, 2 * N1 + 3):   # the last: part of every share is streamed a second time
        x = torch.empty(n, device="cuda").uniform_(-1, 1, generator=g).to(fdtype)
        want_q, want_rec = _reference(ctx, x, qdtype)
        (q, rec), handed = _blocks_handed_over(ctx, lambda: piquant.torch.quantize_dynamic(x, dtype=qdtype, ctx=ctx))
        assert torch.equal(piquant.torch.packed_bytes(q), want_q) and torch.equal(rec, want_rec), (n, qname)
        if n >= N1:
            assert handed == min(cus, 256) - 1, (handed, cus)
        before = ctx.barrier_bailouts()
        q2, rec2 = piquant.torch.quantize_dynamic(x, dtype=qdtype, ctx=ctx)
        torch.cuda.synchronize()
        assert torch.equal(piquant.torch.packed_bytes(q2), want_q) and torch.equal(rec2, want_rec)
        assert ctx.barrier_bailouts() == before, "an ordinary launch on an idle GPU handed shares over"


def test_fused_calls_next_to_a_kernel_that_holds_cus(hog):
    """96 CUs are held for 30 ms by another stream; fused calls issued meanwhile cannot get their whole grid resident.  Round 1
    would have waited (and trapped after ~2 s had the holder depended on it); now the resident blocks hand their shares over
    after 200 us, the rest of the grid starts on the freed CUs, and the results are identical."""
    import piquant

    ctx = piquant.Context()
    sink = torch.zeros(4, dtype=torch.int32, device="cuda")
    g = torch.Generator(device="cuda")
    g.manual_seed(4)
    x = torch.empty(N1, device="cuda").uniform_(-1, 1, generator=g)
    want_q, want_rec = _reference(ctx, x, torch.uint8)
    ctx.set_barrier_timeout_us(200)
    got_in_the_way = False
    # HIP multiplexes streams onto a few hardware queues, and two streams that share one run one after the other: try holders on
    # a high-priority stream (its own queue) and on a few ordinary ones until one really runs next to the context's stream
    for side in [torch.cuda.Stream(priority=-1)] + [torch.cuda.Stream() for _ in range(6)]:
        before = ctx.barrier_bailouts()
        torch.cuda.synchronize()
        assert hog.cu_hog_launch(ctypes.c_void_p(side.cuda_stream), 96, 30_000, ctypes.c_void_p(sink.data_ptr())) == 0
        outs = [piquant.torch.quantize_dynamic(x, dtype=torch.uint8, ctx=ctx) for _ in range(8)]
        torch.cuda.synchronize()
        for q, rec in outs:
            assert torch.equal(q, want_q) and torch.equal(rec, want_rec)
        if ctx.barrier_bailouts() > before:
            got_in_the_way = True
            break
    ctx.set_barrier_timeout_us(0)
    if not got_in_the_way:   # scheduling luck, not a property of the library: the bytes above were checked either way
        import warnings

        warnings.warn("no holder ever ran next to the fused kernel on this box: only the byte comparison was exercised")


def test_reduce_variant_hands_over_too():
    """the owner's step of the mesh all-reduce (acc + sum of dequantized chunks, quantized in the same launch) under a 1 us limit"""
    import piquant

    ctx = piquant.Context()
    n = 3_408_000 * 4
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    acc = torch.empty(n, device="cuda").uniform_(-1, 1, generator=g)
    chunks = [torch.empty(n, device="cuda").uniform_(-1, 1, generator=g) for _ in range(3)]
    qs = [piquant.torch.quantize_dynamic(c, dtype=torch.uint8, ctx=ctx) for c in chunks]
    ctx.set_fusion(False)
    want_q, want_rec = piquant.torch.reduce_quantize_dynamic(acc.clone(), [q for q, _ in qs], [r for _, r in qs], dtype=torch.uint8, ctx=ctx)
    torch.cuda.synchronize()
    ctx.set_fusion(True)
    ctx.set_barrier_timeout_us(1)   # the racy form: bytes only
    q, rec = piquant.torch.reduce_quantize_dynamic(acc.clone(), [q for q, _ in qs], [r for _, r in qs], dtype=torch.uint8, ctx=ctx)
    torch.cuda.synchronize()
    ctx.set_barrier_timeout_us(0)
    assert torch.equal(q, want_q) and torch.equal(rec, want_rec)
    # the deterministic form: the adopting block re-adds the terms to the shares it picks up
    (q, rec), handed = _blocks_handed_over(
        ctx, lambda: piquant.torch.reduce_quantize_dynamic(acc.clone(), [q for q, _ in qs], [r for _, r in qs], dtype=torch.uint8, ctx=ctx))
    assert torch.equal(q, want_q) and torch.equal(rec, want_rec)
    assert handed >= 63, handed


def test_two_threads_two_contexts_launch_fused_kernels_at_once():
    """two threads, two contexts, two streams -- their barrier kernels must never end up interleaved between each
    other's wait and record (csrc/context.cpp FusedLaunchOrder holds the per-device lock across wait + launch + record)."""
    import piquant

    g = torch.Generator(device="cuda")
    g.manual_seed(6)
    xs = [torch.empty(N1, device="cuda").uniform_(-1, 1, generator=g) * (i + 1) for i in range(2)]
    ref_ctx = piquant.Context()
    wants = [_reference(ref_ctx, x, torch.uint8) for x in xs]
    errors = []

    def work(i):
        try:
            torch.cuda.set_device(0)
            ctx = piquant.Context()
            ctx.set_barrier_timeout_us(50_000)   # a hand-over below must mean "two barrier kernels ran side by side", not "a block once waited 1 ms"
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                outs = [piquant.torch.quantize_dynamic(xs[i], dtype=torch.uint8, ctx=ctx) for _ in range(150)]
            stream.synchronize()
            for q, rec in outs[::10] + outs[-1:]:
                if not (torch.equal(q, wants[i][0]) and torch.equal(rec, wants[i][1])):
                    errors.append((i, "mismatch"))
            if ctx.barrier_bailouts() != 0:
                errors.append((i, f"{ctx.barrier_bailouts()} blocks left a barrier early although launches are ordered"))
        except Exception as exc:   # noqa: BLE001
            errors.append((i, repr(exc)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors


def test_device_record_of_an_empty_or_all_nan_tensor_is_the_degenerate_one():
    """nothing scanned (max < min, the armed identities) must not leave a negative scale in a device record."""
    import piquant

    ctx = piquant.Context()
    for x in (torch.empty(0, device="cuda"), torch.full((1000,), float("nan"), device="cuda"), torch.empty(0, device="cuda", dtype=torch.bfloat16)):
        for qdtype, zp in ((torch.uint8, 127), (torch.quint4x2, 7), (torch.quint2x4, 1)):
            rec = piquant.torch.compute_quant_params_device(x, dtype=qdtype, ctx=ctx)
            assert piquant.torch.params_to_host(rec) == (1.0, zp)
            q, rec2 = piquant.torch.quantize_dynamic(x, dtype=qdtype, ctx=ctx)
            assert piquant.torch.params_to_host(rec2) == (1.0, zp) and q.numel() == x.numel()


def test_batched_launch_hands_over_too():
    """several tensors in one launch, each sub-grid with its own barrier and orphan bitmap, under a 1 us limit"""
    import piquant

    ctx = piquant.Context()
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    xs = [torch.empty(n, device="cuda").uniform_(-1, 1, generator=g) * (i + 1) for i, n in enumerate((3_408_000, 1_000_003 + 1, 3_408_000, 4096, 2_000_000))]
    ctx.set_fusion(False)
    want = piquant.torch.quantize_dynamic_batch(xs, dtype=torch.uint8, ctx=ctx)
    torch.cuda.synchronize()
    ctx.set_fusion(True)
    ctx.set_barrier_timeout_us(1)   # the racy form: bytes only
    for _ in range(3):
        got = piquant.torch.quantize_dynamic_batch(xs, dtype=torch.uint8, ctx=ctx)
    torch.cuda.synchronize()
    ctx.set_barrier_timeout_us(0)
    for (q, r), wq, wr in zip(zip(*got), *want):
        assert torch.equal(q, wq) and torch.equal(r, wr)
    # the deterministic form: every sub-grid keeps its last block only
    got, handed = _blocks_handed_over(ctx, lambda: piquant.torch.quantize_dynamic_batch(xs, dtype=torch.uint8, ctx=ctx))
    for (q, r), wq, wr in zip(zip(*got), *want):
        assert torch.equal(q, wq) and torch.equal(r, wr)
    assert handed >= len(xs), handed


def _two_process_worker(rank, out_q):
    import sys

    root = Path(__file__).resolve().parent.parent
    for p in (str(root), str(root / "pi-quant_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import piquant

    torch.cuda.set_device(0)
    ctx = piquant.Context()
    g = torch.Generator(device="cuda")
    g.manual_seed(100 + rank)
    x = torch.empty(N1, device="cuda").uniform_(-1, 1, generator=g)
    ctx.set_fusion(False)
    wq, wr = piquant.torch.quantize_dynamic(x, dtype=torch.uint8, ctx=ctx)
    torch.cuda.synchronize()
    ctx.set_fusion(True)
    ok = True
    for _ in range(10):
        outs = [piquant.torch.quantize_dynamic(x, dtype=torch.uint8, ctx=ctx) for _ in range(30)]
        torch.cuda.synchronize()
        ok = ok and all(torch.equal(q, wq) and torch.equal(r, wr) for q, r in outs)
    out_q.put((rank, ok, ctx.barrier_bailouts()))


def test_two_processes_sharing_the_gpu_launch_fused_kernels():
    """Two PROCESSES on one GPU, each issuing 300 barrier kernels with no coordination between them (nothing the library can order):
    round 1 had to switch the fused path off for this.  Both must finish with the right bytes; how many blocks handed over is reported."""
    import torch.multiprocessing as mp

    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_two_process_worker, args=(r, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in results), results
    print("hand-overs per process:", {r: b for r, _, b in results})
