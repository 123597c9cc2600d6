"""GPU: piquant.distributed on a real device -- the HIP scan feeding the RCCL all-reduce (single rank here:
the box has one GPU; world_size 2/3 is covered on CPU over gloo in test_distributed_cpu.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pg():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()


def test_sharded_params_through_rccl(pg, oracle_mod):
    import piquant.distributed as D

    O = oracle_mod
    x = np.random.default_rng(5).normal(size=3_000_001).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    for tdt, odt in ((torch.quint8, O.UINT8), (torch.quint4x2, O.UINT4), (torch.quint2x4, O.UINT2)):
        assert D.compute_quant_params(xd, dtype=tdt) == O.compute_quant_params(x, O.F32, odt)
    xb = O.f32_to_bf16(x)
    xbd = torch.from_numpy(xb.view(np.int16)).cuda().view(torch.bfloat16)
    assert D.compute_quant_params(xbd, dtype=torch.quint8) == O.compute_quant_params(xb, O.BF16, O.UINT8)


def test_transport_branches_run_on_rccl(pg):
    """the `nccl` branches of _exchange / _all_to_all / _all_gather had never executed.  A one-rank RCCL group can run all
    three (send/recv to self inside one group call, all_to_all_single, all_gather_into_tensor): device buffers, no host staging."""
    import piquant.distributed as D

    assert dist.get_backend() == "nccl"
    g = torch.Generator(device="cuda")
    g.manual_seed(9)
    for nbytes in (16, 4097, 3_408_016):
        send = torch.randint(0, 256, (nbytes,), dtype=torch.uint8, device="cuda", generator=g)
        recv = torch.zeros_like(send)
        D._exchange(send, recv, 0, 0, None)
        torch.cuda.synchronize()
        assert torch.equal(send, recv)
        recv.zero_()
        D._all_to_all(send, recv, None)
        torch.cuda.synchronize()
        assert torch.equal(send, recv)
        everyone = torch.zeros_like(send)
        D._all_gather(send, everyone, None)
        torch.cuda.synchronize()
        assert torch.equal(send, everyone)


@pytest.mark.parametrize("algorithm", ["ring", "direct"])
@pytest.mark.parametrize("fdtype,qname,numel", [(torch.float32, "uint8", 27_264_000), (torch.bfloat16, "quint4x2", 1_000_003), (torch.float32, "quint2x4", 4099)])
def test_quantized_all_reduce_end_to_end_on_a_one_rank_rccl_group(pg, oracle_mod, algorithm, fdtype, qname, numel):
    """The whole schedule with RCCL as the transport (test hook: the rank is its own only peer): fused encode, the group's collectives,
    decode -- while a second stream keeps launching fused kernels, the combination VERDICT r01 called untested.  With one rank the
    sum has one term, so every element must equal dequantize(quantize(x)) with parameters from x: checked against the oracle."""
    import piquant
    import piquant.distributed as D

    O = oracle_mod
    qdtype = getattr(torch, qname)
    odt = {"uint8": O.UINT8, "quint4x2": O.UINT4, "quint2x4": O.UINT2}[qname]
    g = torch.Generator(device="cuda")
    g.manual_seed(10)
    x = torch.empty(numel, device="cuda").uniform_(-1, 1, generator=g).to(fdtype)
    host = x.view(torch.int16).cpu().numpy().view(np.uint16) if fdtype == torch.bfloat16 else x.cpu().numpy()
    fdt = O.BF16 if fdtype == torch.bfloat16 else O.F32
    # a second stream issuing barrier kernels of its own meanwhile
    side, side_ctx = torch.cuda.Stream(), piquant.Context()
    y = torch.empty(27_264_000, device="cuda").uniform_(-1, 1, generator=g)
    with torch.cuda.stream(side):
        side_out = [piquant.torch.quantize_dynamic(y, dtype=torch.uint8, ctx=side_ctx) for _ in range(40)]
    t = x.clone()
    D.quantized_all_reduce(t, quant_dtype=qdtype, algorithm=algorithm, _single_rank_collectives=True)
    torch.cuda.synchronize()
    scale, zp = O.compute_quant_params(host, fdt, odt)
    want = O.dequantize(O.quantize(host, fdt, odt, scale, zp), odt, fdt, numel, scale, zp)
    got = t.view(torch.int16).cpu().numpy().view(np.uint16) if fdtype == torch.bfloat16 else t.cpu().numpy()
    assert np.array_equal(got.view(np.uint16 if fdtype == torch.bfloat16 else np.uint32), want.view(np.uint16 if fdtype == torch.bfloat16 else np.uint32))
    assert all(torch.equal(q, side_out[0][0]) and torch.equal(r, side_out[0][1]) for q, r in side_out[1:])


def test_keys_fold_like_an_all_reduce(pg, oracle_mod):
    """MIN over the key pairs of several shards == keys of the whole tensor: what all_reduce(MIN) computes."""
    import piquant
    import piquant.distributed as D

    O = oracle_mod
    x = np.random.default_rng(6).uniform(-3, 5, 1_000_000).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    world = 8
    parts = []
    for r in range(world):
        b, e = D.shard_range(x.size, r, world, 4)
        parts.append(D.local_minmax_keys(xd[b:e]))
    folded = torch.stack(parts).min(dim=0).values.cpu()
    whole = D.local_minmax_keys(xd).cpu()
    assert torch.equal(folded, whole)
    assert piquant.decode_minmax_keys(int(folded[0]), int(folded[1])) == O.minmax(x, O.F32)


# ---------------------------------------------------------------------------------------------------------------
# quantized ring all-reduce with the real HIP ops: two (three) processes share the box's single GPU, the transport is
# gloo staged through host memory (RCCL refuses two ranks on one device); the schedule and every kernel are the real
# ones, so each rank must reproduce the oracle simulation bit for bit.
# ---------------------------------------------------------------------------------------------------------------
def _ring_gpu_worker(rank, world, port, numel, qname, out_q, algorithm="ring", transport="collective", repeats=1, numels=None):
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    for p in (str(root), str(root / "pi-quant_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    # several processes share ONE GPU here (not the deployment model, which is one process per GPU).  Round 1 switched the one-launch
    # params + quantize kernel off for this (two barrier kernels of different processes can each get part of the CUs); its waits are
    # bounded now and a block that cannot be waited for hands its share over, so the fused path stays on
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import piquant.distributed as D

        torch.cuda.set_device(0)
        outs = []
        for rep in range(repeats):      # rep > 0: fresh data through the same buffers (the p2p transport alternates two parities of them)
            n_rep = numels[rep] if numels else numel
            x = torch.from_numpy(np.random.default_rng(100 + rank + 1000 * rep).uniform(-1, 1, n_rep).astype(np.float32)).cuda()
            D.quantized_all_reduce(x, quant_dtype=getattr(torch, qname), algorithm=algorithm, transport=transport)
            outs.append(x)
        torch.cuda.synchronize()
        if transport == "p2p":
            D.release_peer_meshes()
        out_q.put((rank, outs[0].cpu().numpy() if repeats == 1 else [o.cpu().numpy() for o in outs]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("algorithm", ["ring", "direct"])
@pytest.mark.parametrize("world,numel,qname", [(2, 1_000_003, "uint8"), (3, 300_000, "quint4x2")])
def test_quantized_all_reduce_with_hip_kernels(oracle_mod, world, numel, qname, algorithm):
    import sys

    import torch.multiprocessing as mp

    sys.path.insert(0, os.path.dirname(__file__))
    import piquant.distributed as D
    from ring_sim import simulate, simulate_direct

    O = oracle_mod
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ring_gpu_worker, args=(r, world, port, numel, qname, q, algorithm)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    xs = [np.random.default_rng(100 + r).uniform(-1, 1, numel).astype(np.float32) for r in range(world)]
    qd, bits = {"uint8": (O.UINT8, 8), "quint4x2": (O.UINT4, 4)}[qname]
    want = (simulate if algorithm == "ring" else simulate_direct)(O, xs, qd, D.ring_chunks(numel, world, bits))
    for r in range(world):
        assert np.array_equal(results[r], want[r]), r
    exact = np.sum(xs, axis=0)
    assert np.abs(results[0] - exact).max() <= world * (2.0 * world / ((1 << bits) - 1)) * 0.5 + 1e-5


@pytest.mark.parametrize("world,numel,qname", [(2, 1_000_003, "uint8"), (3, 300_000, "quint4x2")])
def test_quantized_all_reduce_p2p_transport_equals_the_collective_one(oracle_mod, world, numel, qname):
    """transport='p2p': the encode kernels store straight into the peers' receive buffers (IPC-mapped device memory; here 2-3 processes on
    the one GPU map each other's allocations), flags order the steps, the decode kernels read the finished chunks from their owners -- no
    collective.  Three all-reduces in a row through the same buffers (both parities, and a parity reused): every rank, every time, byte
    for byte what the oracle simulation of the mesh schedule gives -- which is what the collective transport is held to above."""
    import sys

    import torch.multiprocessing as mp

    sys.path.insert(0, os.path.dirname(__file__))
    import piquant.distributed as D
    from ring_sim import simulate_direct

    O = oracle_mod
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    reps = 3
    procs = [ctx.Process(target=_ring_gpu_worker, args=(r, world, port, numel, qname, q, "direct", "p2p", reps)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    qd, bits = {"uint8": (O.UINT8, 8), "quint4x2": (O.UINT4, 4)}[qname]
    for rep in range(reps):
        xs = [np.random.default_rng(100 + r + 1000 * rep).uniform(-1, 1, numel).astype(np.float32) for r in range(world)]
        want = simulate_direct(O, xs, qd, D.ring_chunks(numel, world, bits))
        for r in range(world):
            assert np.array_equal(results[r][rep], want[r]), (rep, r)


def test_quantized_all_reduce_p2p_mesh_grows_with_the_tensors(oracle_mod):
    """One mesh per group and device, grown when a tensor needs larger slots: small, large (the mesh is rebuilt -- a collective every rank enters
    at the same call), small again (served by the large mesh's layout), larger still; every result is the oracle simulation's."""
    import sys

    import torch.multiprocessing as mp

    sys.path.insert(0, os.path.dirname(__file__))
    import piquant.distributed as D
    from ring_sim import simulate_direct

    O = oracle_mod
    world, numels = 2, [40_000, 900_001, 5_000, 2_000_000, 900_001]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ring_gpu_worker, args=(r, world, port, 0, "uint8", q, "direct", "p2p", len(numels), numels)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rep, n in enumerate(numels):
        xs = [np.random.default_rng(100 + r + 1000 * rep).uniform(-1, 1, n).astype(np.float32) for r in range(world)]
        want = simulate_direct(O, xs, O.UINT8, D.ring_chunks(n, world, 8))
        for r in range(world):
            assert np.array_equal(results[r][rep], want[r]), (rep, r)


def _p2p_params_worker(rank, world, port, numel, reps, out_q):
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    for p in (str(root), str(root / "pi-quant_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import piquant.distributed as D

        torch.cuda.set_device(0)
        got = []
        for rep in range(reps):
            whole = np.random.default_rng(500 + rep).normal(size=numel).astype(np.float32) * (1.0 + rep)
            b, e = D.shard_range(numel, rank, world, 8)
            shard = torch.from_numpy(whole[b:e]).cuda()
            got.append((D.compute_quant_params(shard, dtype=torch.quint8, transport="p2p"), D.compute_quant_params(shard, dtype=torch.quint4x2, transport="p2p"),
                        D.compute_quant_params(shard, dtype=torch.quint8)))     # the collective transport (gloo here) beside it
        torch.cuda.synchronize()
        D.release_peer_meshes()
        out_q.put((rank, got))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_params_over_peer_mapped_mailboxes(oracle_mod, world):
    """compute_quant_params(transport='p2p'): the 8-byte MIN all-reduce done by one one-wave kernel per rank over peer-mapped mailboxes (here
    the mailboxes of 2-3 processes on the one GPU) instead of a collective.  Seven exchanges in a row -- both mailbox parities, each reused
    several times -- with the extremes moving between the shards: every rank, every time, the parameters of the whole tensor."""
    import torch.multiprocessing as mp

    O = oracle_mod
    numel, reps = 1_000_003, 7
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_p2p_params_worker, args=(r, world, port, numel, reps, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rep in range(reps):
        whole = np.random.default_rng(500 + rep).normal(size=numel).astype(np.float32) * np.float32(1.0 + rep)
        want8, want4 = O.compute_quant_params(whole, O.F32, O.UINT8), O.compute_quant_params(whole, O.F32, O.UINT4)
        for r in range(world):
            assert results[r][rep] == (want8, want4, want8), (rep, r)


def test_native_rccl_all_reduce_entry_point(oracle_mod):
    """piquant_hip_compute_quant_params_dist: the C-level sharded call that runs ncclAllReduce itself (RCCL resolved from the
    copy already loaded in the process).  One rank here -- the box has one GPU -- which still drives the whole path: scan,
    fold, ncclAllReduce(2 x int32, ncclMin) on the context's stream, epilogue."""
    import ctypes

    import piquant

    O = oracle_mod
    rccl = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]

    rccl.ncclGetUniqueId.argtypes = [ctypes.POINTER(UniqueId)]
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    torch.cuda.set_device(0)
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    try:
        ctx = piquant.Context()
        for n in (1, 1000, 3_000_001):
            x = np.random.default_rng(n).normal(size=n).astype(np.float32)
            xd = torch.from_numpy(x).cuda()
            for dt, odt in ((piquant.DataType.UINT8, O.UINT8), (piquant.DataType.UINT4, O.UINT4)):
                got = ctx.compute_quant_params_dist_ptr(xd.data_ptr(), piquant.DataType.F32, n, dt, comm.value)
                assert got == O.compute_quant_params(x, O.F32, odt), (n, dt)
    finally:
        rccl.ncclCommDestroy(comm)


def _p2p_child_record(stderr: str):
    """bench.py prints its one line first and only then starts the p2p child job (round-4 verdict: nothing in `bench.py --gpus N` may be able to lose
    the headline the day a node appears); the child's record is the stderr line `bench.py p2p_transport_child_job: {json}`."""
    import json

    tag = "bench.py p2p_transport_child_job: "
    recs = [ln[len(tag):] for ln in stderr.splitlines() if ln.startswith(tag)]
    assert len(recs) == 1, stderr[-3000:]
    return json.loads(recs[0])


@pytest.mark.slow
def test_bench_eight_ranks_sharing_the_gpu():
    """The command the driver will run on an 8-GPU node plus --extras (the default prints the line with n1_reference and config 5 only -- what
    test_bench_two_ranks_default_flags_print_the_line_within_a_minute covers), with the eight ranks sharing this box's GPU over gloo (RCCL refuses two ranks on one
    device): 192 buffer sets of 3 408 000 elements per rank allocate, every barrier is reached, ONE JSON line comes out, the shards and the
    config-5 shards have the sizes BASELINE implies.  The numbers mean nothing; the control flow must not be what kills the first real run."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(root / "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5", "--backend", "gloo",
                        "--share-gpu", "--extras"], capture_output=True, text=True, timeout=1500, cwd=str(root),   # --extras: the driver's line + every side measurement
                       env=dict(os.environ, PIQUANT_BENCH_EXTRAS_LIMIT_S="1200"))   # eight ranks on ONE GPU, collectives staged through the host: slow, and not the point
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["numel_total"] == 27_264_000 and d["config"]["numel_per_gpu"] == 3_408_000 and d["config"]["buffer_sets"] == 192
    assert d["roofline"]["algorithmic_bytes_per_launch"] == 5 * 3_408_000 and 0 < d["roofline"]["frac"] < 1
    c5 = d["extras"]["config5_sharded_compute_quant_params"]
    assert c5["result_correct"] and c5["numel_per_gpu"] == 1 << 27 and c5["numel_total"] == 1 << 30 and "gloo (8 ranks)" in c5["note"]
    weak = d["extras"]["weak_scaling_own_tensor_per_gpu"]
    assert weak["scaling"] == "weak" and weak["numel_per_gpu"] == 27_264_000 and weak["GiB/s"] > 0
    assert d["shard_bit_exact"] == [True] * 8 and d["ranks_seen"]["world_size"] == 8 and d["ranks_seen"]["launcher"] == "external"
    assert d["n1_reference"]["bit_exact"] is True
    ar = d["extras"]["all_reduce_109MB"]
    assert ar["quantized_all_reduce_direct_u8"]["within_bound"] and ar["quantized_all_reduce_ring_u8"]["within_bound"]
    child = _p2p_child_record(r.stderr)             # started after the line was out; its record is on stderr
    assert child["ranks"] == 8 and child["p2p_bit_identical_to_collective"] is True


def test_bench_refuses_more_ranks_than_devices():
    """WORLD_SIZE above the number of visible devices (without the test-only --share-gpu) must fail at once with a message, not hang in RCCL."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(root / "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=str(root))
    assert r.returncode != 0
    assert "visible device" in r.stderr, r.stderr[-2000:]


@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("qname,fdtype", [("uint8", torch.float32), ("quint4x2", torch.bfloat16), ("quint2x4", torch.float32)])
def test_quantize_shard_concatenation_is_the_whole_call(oracle_mod, world, qname, fdtype):
    """Strong scaling of the headline tensor: the shards of ONE 27 264 000-element tensor (reference split, src/piquant.cpp:145-157),
    quantized rank by rank with the HIP kernels, are byte for byte the single-call result; dequantize_shard (SET and ADD) likewise."""
    import piquant
    import piquant.distributed as D

    n = 27_264_000 + (3 if qname != "uint8" else 0)     # ragged for the packed types
    qdtype = getattr(torch, qname)
    g = torch.Generator(device="cuda")
    g.manual_seed(17)
    x = torch.empty(n, device="cuda").uniform_(-1, 1, generator=g).to(fdtype)
    scale, zp = piquant.torch.compute_quant_params(x, dtype=qdtype)
    # (the whole calls in the position-independent form too: the plain call's bytes are those of a reference context of num_threads pool threads,
    # whose tails double-round bf16 ADD and, for uint2 -> fp32 ADD, store -- ranks are not pool threads)
    whole = piquant.torch.packed_bytes(piquant.torch.quantize(x, scale=scale, zero_point=zp, dtype=qdtype, uniform=True))
    buf = torch.full_like(whole, 0xAA)
    for r in range(world):
        dst, (b, e) = D.quantize_shard(x, scale=scale, zero_point=zp, dtype=qdtype, out=buf, rank=r, world_size=world)
        assert e > b and dst.numel() == piquant.torch.torch_to_piquant_dtype(qdtype).packed_nbytes(e - b)
    assert torch.equal(buf, whole)
    full = piquant.torch.dequantize(whole, scale=scale, zero_point=zp, dtype=fdtype, quant_dtype=qdtype, shape=(n,), uniform=True)
    out = torch.full((n,), 7.0, device="cuda", dtype=fdtype)
    for r in range(world):
        D.dequantize_shard(buf, numel=n, scale=scale, zero_point=zp, quant_dtype=qdtype, out=out, rank=r, world_size=world)
    assert torch.equal(out.view(torch.int16 if fdtype == torch.bfloat16 else torch.int32), full.view(torch.int16 if fdtype == torch.bfloat16 else torch.int32))
    acc = torch.ones(n, device="cuda", dtype=fdtype)
    want = piquant.torch.dequantize(whole, scale=scale, zero_point=zp, dtype=fdtype, quant_dtype=qdtype, shape=(n,), reduce_op="add",
                                    out=torch.ones(n, device="cuda", dtype=fdtype), uniform=True)
    for r in range(world):
        D.dequantize_shard(buf, numel=n, scale=scale, zero_point=zp, quant_dtype=qdtype, out=acc, reduce_op="add", rank=r, world_size=world)
    assert torch.equal(acc, want)
    # and the whole thing equals the oracle on a window (the full-size oracle comparison lives in test_gpu_parity.py::test_config2)
    O = oracle_mod
    if fdtype == torch.float32 and qname == "uint8":
        lo = D.shard_range(n, world - 1, world, 8)[0] - 1000
        xs = x[lo: lo + 5000].cpu().numpy()
        assert np.array_equal(buf[lo: lo + 5000].cpu().numpy(), O.quantize(xs, O.F32, O.UINT8, scale, zp))


# ---------------------------------------------------------------------------------------------------------------
# bench.py's multi-rank control flow (barriers, max over ranks, the sharded config-5 measurement with its collective)
# on ONE GPU: two ranks share cuda:0 and talk gloo, because RCCL refuses two ranks on one device.  The numbers mean
# nothing; the run must finish, print exactly one JSON line, shard the headline tensor (strong scaling) and get the sharded
# parameters right.
# ---------------------------------------------------------------------------------------------------------------
def test_bench_two_ranks_default_flags_print_the_line_within_a_minute():
    """Round-5 verdict: the first scaling run must not be lost to a side measurement.  `bench.py --gpus 2` with DEFAULT flags is the contract line, its
    N = 1 reference point and config 5 (the one path with a collective) -- no all-reduce schedules, no graph replay, no weak scaling, no
    peer-to-peer child job -- and the line is on stdout within 60 s of the process start (two ranks sharing this box's one GPU)."""
    import json
    import os
    import subprocess
    import sys
    import time
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PIQUANT_BENCH_EXTRAS")}
    t0 = time.perf_counter()
    p = subprocess.Popen([sys.executable, str(root / "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu", "--steps", "20", "--warmup", "5"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=str(root), env=env)
    line = p.stdout.readline()
    t_line = time.perf_counter() - t0
    _, err = p.communicate(timeout=300)
    assert p.returncode == 0, err[-3000:]
    d = json.loads(line)
    assert t_line < 60.0, f"the contract line took {t_line:.1f} s"
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and d["shard_bit_exact"] == [True, True]
    assert "piquant_hip_quantize_uniform" in d["config"]["api"]          # a rank's call covers a shard: the position-independent twin
    assert d["n1_reference"]["bit_exact"] is True
    assert set(d["extras"]) == {"config5_sharded_compute_quant_params"} and d["extras"]["config5_sharded_compute_quant_params"]["result_correct"]
    assert "p2p_transport_child_job" not in err


def test_bench_two_ranks_sharing_the_gpu():
    """`python bench.py --gpus 2 --extras` with NO launcher around it: bench.py starts its own two ranks, one JSON line comes out, and the line
    validates itself -- ranks seen by the process group, every rank's shard bit-exact against the checker, the N = 1 point of the same run, both
    all-reduce schedules and the 8-byte MIN all-reduce in the extras."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu", "--steps", "20", "--warmup", "5", "--extras"],
                       capture_output=True, text=True, timeout=900, cwd=str(root), env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    # the headline is the north-star workload: ONE 27 264 000-element tensor split over the ranks (reference src/piquant.cpp:145-157)
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["numel_total"] == 27_264_000 and d["config"]["numel_per_gpu"] == 13_632_000
    assert d["roofline"]["algorithmic_bytes_per_launch"] == 5 * 13_632_000
    # value counts the whole tensor once per step, not once per rank
    assert abs(d["value"] - 27_264_000 * 4 / 2**30 * 20 / (d["ms_per_step"] * 20e-3)) / d["value"] < 1e-3
    # self-validation
    assert d["shard_bit_exact"] == [True, True] and d["self_check"]["all_bit_exact"] is True
    assert d["self_check"]["shards"] == [[0, 13_632_000], [13_632_000, 27_264_000]]
    seen = d["ranks_seen"]
    assert seen["world_size"] == 2 and seen["backend"] == "gloo" and len(seen["devices"]) == 2 and seen["distinct_devices"] == 1   # --share-gpu
    assert "own" in seen["launcher"]
    n1 = d["n1_reference"]
    assert n1["GiB/s"] > 0 and n1["bit_exact"] is True and 0 < n1["roofline_frac"] < 1
    c5 = d["extras"]["config5_sharded_compute_quant_params"]
    assert c5["result_correct"] and c5["numel_per_gpu"] == (1 << 30) // 2 and "gloo" in c5["note"]
    assert c5["ms_per_call_without_collective"] > 0 and "collective_adds_ms" in c5
    weak = d["extras"]["weak_scaling_own_tensor_per_gpu"]
    assert weak["scaling"] == "weak" and weak["numel_per_gpu"] == 27_264_000 and weak["GiB/s"] > 0
    ar = d["extras"]["all_reduce_109MB"]
    assert ar["ranks"] == 2 and ar["all_reduce_fp32"]["ms"] > 0 and ar["min_all_reduce_8_bytes"]["us_per_call"] > 0
    for algo in ("direct", "ring"):
        rec = ar[f"quantized_all_reduce_{algo}_u8"]
        assert rec["ms"] > 0 and rec["within_bound"] is True and rec["ranks_bit_identical"] is True, rec
    child = _p2p_child_record(r.stderr)            # the peer-to-peer transport, measured by a child job that starts AFTER the line is out
    assert child["ranks"] == 2 and child["p2p_bit_identical_to_collective"] is True, child
    assert child["quantized_all_reduce_direct_u8_p2p"]["within_bound"] is True and child["quantized_all_reduce_direct_u8_p2p"]["ms"] > 0


def test_bench_own_launch_refuses_more_ranks_than_devices():
    """the same loud refusal when bench.py launches its own ranks: no hang in RCCL, no traceback -- a message and a non-zero exit code"""
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1"], capture_output=True, text=True,
                       timeout=300, cwd=str(root), env=env)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "visible device" in r.stderr, r.stderr[-2000:]


def test_bench_headline_survives_side_measurements_that_overrun():
    """N > 1: the side measurements behind the headline contain collectives; when they do not finish in time (a rank that never reaches a
    collective hangs the others for RCCL's timeout) every rank's watchdog ends the run and rank 0 prints the headline without them."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PIQUANT_BENCH_EXTRAS_LIMIT_S="0.0005")   # fires while the first side measurement is being set up
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(root / "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--backend", "gloo",
                        "--share-gpu"], capture_output=True, text=True, timeout=600, cwd=str(root), env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["value"] > 0 and 0 < d["roofline"]["frac"] < 1
    assert "did not finish" in d["extras"]["error"]


# ---------------------------------------------------------------------------------------------------------------
# peer-to-peer transport: first contact with real peers must not fault a queue (round-4 advisor: the two waits trapped after 30 s)
# ---------------------------------------------------------------------------------------------------------------
def test_peer_wait_that_runs_out_reports_the_missing_rank_instead_of_trapping():
    """One GPU, no peers needed: a flag nobody ever signals, a mailbox slot nobody ever fills.  The waiting wave gives up after the limit,
    writes {kind, rank, expected, seen} into the context's pinned record and lets the stream go on; the host reads the record
    (piquant_hip_peer_timeout / Context.peer_timeout) and the GPU is as usable as before -- nothing faulted."""
    import piquant

    torch.cuda.set_device(0)
    ctx = piquant.Context()
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    ctx.set_blocking(False)
    assert ctx.peer_timeout() is None
    flags = torch.zeros(4, dtype=torch.int32, device="cuda")
    flags[0] = 7
    flags[1] = 7
    flags[3] = 7                                            # "rank 2" never signals exchange 7
    torch.cuda.synchronize()
    ctx.wait_flags_ptr(flags.data_ptr(), 4, 7, timeout_us=20_000)
    torch.cuda.synchronize()
    assert ctx.peer_timeout() == ("flags", 2, 7, 0)
    assert ctx.peer_timeout() is None                      # fetched: cleared
    flags[2] = 7
    ctx.wait_flags_ptr(flags.data_ptr(), 4, 7, timeout_us=20_000)   # everybody there: returns at once, reports nothing
    torch.cuda.synchronize()
    assert ctx.peer_timeout() is None
    # the mailbox exchange: rank 0 of a two-rank group whose rank 1 never delivers (its slot in the own mailbox stays empty)
    empty = 0x7fffffff7fffffff
    mailbox = torch.full((2,), empty, dtype=torch.int64, device="cuda")
    elsewhere = torch.full((2,), empty, dtype=torch.int64, device="cuda")      # stands in for rank 1's mailbox
    keys = torch.tensor([123, 456], dtype=torch.int32, device="cuda")
    out = torch.zeros(2, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ctx.exchange_minmax_keys_ptr(keys.data_ptr(), [mailbox.data_ptr(), elsewhere.data_ptr()], mailbox.data_ptr(), out.data_ptr(), timeout_us=20_000)
    torch.cuda.synchronize()
    assert ctx.peer_timeout() == ("keys", 1, 0, 0)
    assert int(elsewhere[0]) == (456 << 32 | 123)           # the own pair did go out
    # and the device still works
    x = torch.empty(1_000_003, device="cuda").uniform_(-1, 1)
    q, rec = piquant.torch.quantize_dynamic(x, dtype=torch.uint8, ctx=ctx)
    s_, z_ = piquant.torch.params_to_host(rec)
    assert piquant.torch.compute_quant_params(x, dtype=torch.quint8) == (s_, z_)


def test_unfetched_peer_timeout_aborts_the_next_peer_call_with_the_rank_named():
    """A C host that never asks: the record left by a wait that ran out makes the context's NEXT peer-to-peer call abort with a message that
    names the rank (subprocess: SIGABRT + the message), instead of carrying on with stale bytes for ever."""
    import signal
    import subprocess
    import sys
    import textwrap
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {str(root / 'pi-quant_amd')!r})
        import torch, piquant
        ctx = piquant.Context()
        ctx.set_blocking(False)
        flags = torch.zeros(3, dtype=torch.int32, device='cuda')
        torch.cuda.synchronize()
        ctx.wait_flags_ptr(flags.data_ptr(), 3, 1, timeout_us=10_000)
        torch.cuda.synchronize()
        ctx.wait_flags_ptr(flags.data_ptr(), 3, 1, timeout_us=10_000)
        print('not reached')
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == -signal.SIGABRT, (r.returncode, r.stderr[-600:])
    assert "rank 0 never signalled exchange 1" in r.stderr and "not reached" not in r.stdout


def _p2p_late_rank_worker(rank, world, port, out_q):
    import sys
    import time
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    for p in (str(root), str(root / "pi-quant_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import piquant.distributed as D

        torch.cuda.set_device(0)
        n = 400_000
        report = {}
        x = torch.from_numpy(np.random.default_rng(300 + rank).uniform(-1, 1, n).astype(np.float32)).cuda()
        D.quantized_all_reduce(x.clone(), transport="p2p")                 # builds the meshes (collective); everybody on time
        on_time = D.compute_quant_params(x, dtype=torch.quint8, transport="p2p")
        assert on_time == D.compute_quant_params(x, dtype=torch.quint8)
        torch.cuda.synchronize()
        dist.barrier()
        # 1. rank 1 is 1.5 s late, the others wait up to 30 s: still one correct all-reduce (round 4's fixed 30 s trap would have been fine here too,
        #    the point is that the timeout is the caller's now)
        if rank == 1:
            time.sleep(1.5)
        y = x.clone()
        D.quantized_all_reduce(y, transport="p2p", timeout=30.0)
        D.check_peer_timeouts()
        report["late_but_in_time"] = y.cpu().numpy()
        dist.barrier()
        # 2. rank 1 is 2 s late and the others give up after 0.3 s: nobody faults, the early ranks raise RuntimeError naming rank 1; a rank that gave up
        #    signals nothing more, so the late rank -- which finds everybody's chunk there, but decodes sums made without its own -- never sees rank 0
        #    finish and raises too, naming rank 0 (round 5: it completed silently with a wrong tensor)
        if rank == 1:
            time.sleep(2.0)
        z = x.clone()
        D.quantized_all_reduce(z, transport="p2p", timeout=0.3)
        try:
            D.check_peer_timeouts()
            report["gave_up"] = None
        except RuntimeError as exc:
            report["gave_up"] = str(exc)
        dist.barrier()
        torch.cuda.synchronize()
        # ... and the group's buffers are refused until everybody has rebuilt them: a late store may still land in a later exchange's parity
        try:
            D.quantized_all_reduce(x.clone(), transport="p2p", timeout=30.0)
            report["refused"] = None
        except RuntimeError as exc:
            report["refused"] = str(exc)
        dist.barrier()
        D.release_peer_meshes()
        v = x.clone()
        D.quantized_all_reduce(v, transport="p2p", timeout=30.0)
        D.check_peer_timeouts()
        report["after_release"] = v.cpu().numpy()
        assert D.compute_quant_params(x, dtype=torch.quint8, transport="p2p") == on_time     # (builds the key mesh again: a collective, everybody on time)
        torch.cuda.synchronize()
        dist.barrier()
        # 3. compute_quant_params(transport='p2p') with a late rank beyond the limit raises from the call itself (it is synchronous)
        if rank == 1:
            time.sleep(1.5)
        try:
            D.compute_quant_params(x, dtype=torch.quint8, transport="p2p", timeout=0.3)
            report["params_gave_up"] = None
        except RuntimeError as exc:
            report["params_gave_up"] = str(exc)
        dist.barrier()
        torch.cuda.synchronize()
        # the GPU and the process are alive: an ordinary collective-transport all-reduce still works
        w = x.clone()
        D.quantized_all_reduce(w, transport="collective")
        torch.cuda.synchronize()
        report["alive"] = bool(torch.isfinite(w).all())
        out_q.put((rank, report))
    finally:
        dist.destroy_process_group()


def test_p2p_transport_with_a_late_rank_raises_instead_of_faulting(oracle_mod):
    """Round-4 advisor: a rank that reaches a p2p exchange later than the (then fixed, 30 s) limit trapped the GPU queue of every peer.  Now the
    limit is an argument (default 10 minutes, PIQUANT_P2P_TIMEOUT_S), a wait that runs out is reported, and the Python layer raises
    RuntimeError naming the late rank -- on the ranks that waited; all processes and the GPU survive."""
    import sys

    import torch.multiprocessing as mp

    sys.path.insert(0, os.path.dirname(__file__))
    import piquant.distributed as D
    from ring_sim import simulate_direct

    O = oracle_mod
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_p2p_late_rank_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    n = 400_000
    xs = [np.random.default_rng(300 + r).uniform(-1, 1, n).astype(np.float32) for r in range(world)]
    want = simulate_direct(O, xs, O.UINT8, D.ring_chunks(n, world, 8))
    for r in range(world):
        assert np.array_equal(results[r]["late_but_in_time"], want[r]), r
        assert results[r]["alive"] is True
    assert results[0]["gave_up"] is not None and "rank 1 did not arrive" in results[0]["gave_up"], results[0]["gave_up"]
    assert results[1]["gave_up"] is not None and "rank 0 did not arrive" in results[1]["gave_up"], results[1]["gave_up"]
    for r in range(world):
        assert results[r]["refused"] is not None and "release_peer_meshes" in results[r]["refused"], results[r]["refused"]
        assert np.array_equal(results[r]["after_release"], want[r]), r
    assert results[0]["params_gave_up"] is not None and "rank 1 did not deliver" in results[0]["params_gave_up"], results[0]["params_gave_up"]


def _p2p_refusal_worker(rank, world, port, out_q):
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    for p in (str(root), str(root / "pi-quant_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PIQUANT_P2P_PRETEND_UNREACHABLE="0-2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import piquant.distributed as D

        torch.cuda.set_device(0)
        x = torch.ones(100_000, device="cuda") * (rank + 1)
        msgs = []
        for call in (lambda: D.quantized_all_reduce(x.clone(), transport="p2p"), lambda: D.compute_quant_params(x, dtype=torch.quint8, transport="p2p")):
            try:
                call()
                msgs.append(None)
            except RuntimeError as exc:
                msgs.append(str(exc))
        y = x.clone()
        D.quantized_all_reduce(y, transport="collective")     # what the message recommends works
        torch.cuda.synchronize()
        out_q.put((rank, msgs, float(y[0])))
    finally:
        dist.destroy_process_group()


def test_p2p_transport_refuses_cleanly_when_a_pair_of_gpus_cannot_reach_each_other():
    """The startup check of the p2p transport (hipDeviceCanAccessPeer for every pair, here with one pair DECLARED unreachable through
    PIQUANT_P2P_PRETEND_UNREACHABLE since the box has one GPU): every rank raises the same RuntimeError naming the pair before any IPC handle is
    opened, nothing is left half-built, and the collective transport carries on."""
    import torch.multiprocessing as mp

    world = 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_p2p_refusal_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, msgs, y0 in results:
        assert len(msgs) == 2 and all(m is not None and "refused" in m and "[(0, 2)]" in m for m in msgs), (rank, msgs)
        assert abs(y0 - 6.0) < 0.1
