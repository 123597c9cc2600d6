"""GPU: what an UNCHANGED caller gets.  No environment variable, no additive setter: `piquant_context_create(num_threads)` and the plain
`piquant_quantize` / `piquant_dequantize` write, byte for byte, what the reference's AVX-512 build writes from a context of the same
`num_threads` -- its partitions (src/piquant.cpp:145-157), each with its scalar head (fp32 -> uint8: kernels_specialized.inl:52-56) and scalar
tail (:178-182, :977-981, :1290-1303; dequantize.inl:72-86), handled by the wave tile they fall into inside the ONE vector launch.

Pins: the 1 100+ golden quantize / dequantize vectors recorded from the reference's own kernels (`ref`, not `uniform`) for a one-thread
context, and the oracle's threaded reference form -- itself pinned to the reference kernels run per partition (tests/test_oracle_vs_ref.py) --
at numel 27 264 000 for T = 1, 7 and 255, on data salted with the values on which the reference's formulas differ.  The additive calls that
shards and one-launch paths are made of stay position-independent in the same contexts.
"""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

from helpers import gpu_dequantize, gpu_quantize, load_golden, same_floats

pytestmark = pytest.mark.gpu

N1 = 27_264_000


@pytest.fixture(scope="module")
def O(oracle_mod):
    return oracle_mod


def _ref_quantize(O, x, dt_in, dt_out, scale, zp, threads, off=0):
    """oracle, reference form, `threads` partitions, output at `off` bytes past a 16-byte boundary (the head of fp32 -> uint8 depends on it)"""
    nbytes = O.packed_numel(x.size, dt_out)
    wbuf = np.zeros(nbytes + 32, dtype=np.uint8)
    base = (-wbuf.ctypes.data) % 16
    return O.quantize(x, dt_in, dt_out, scale, zp, form=O.FORM_REFERENCE, threads=threads, out=wbuf[base + off: base + off + nbytes]).copy()


def test_one_thread_context_writes_the_golden_reference_bytes():
    """Every recorded output of the reference's kernels (one call = one pool thread) from a context created with num_threads = 1 and nothing else."""
    import piquant

    ctx = piquant.Context(1)
    cases, get = load_golden()
    nq = nd = differing = 0
    try:
        for c in cases:
            if c["kind"] == "quantize":
                ctx.set_stochastic_threshold(c["tau"] if c["round_mode"] else None)   # the reference's hidden per-call draw, pinned to the recorded one
                got = gpu_quantize(ctx, get(c["name"], "x"), c["dt_in"], c["dt_out"], c["scale"], c["zp"], c["round_mode"])
                assert np.array_equal(got, get(c["name"], "ref")), c
                differing += int(not np.array_equal(get(c["name"], "ref"), get(c["name"], "uniform")))
                nq += 1
            elif c["kind"] == "dequantize":
                got = gpu_dequantize(ctx, get(c["name"], "q"), c["dt_in"], c["dt_out"], c["numel"], c["scale"], c["zp"], c["op"], prev=get(c["name"], "prev"))
                assert same_floats(got, get(c["name"], "ref")), c
                differing += int(not same_floats(get(c["name"], "ref"), get(c["name"], "uniform")))
                nd += 1
    finally:
        ctx.set_stochastic_threshold(None)
    assert nq > 500 and nd > 400
    assert differing > 20   # cases on which the reference's positions matter are among them


@pytest.mark.parametrize("threads", [1, 7, 255])
def test_default_calls_equal_the_reference_context_of_the_same_thread_count_at_full_size(O, threads):
    """numel 27 264 000, a context created with `threads`: quantize (fp32 -> uint8 aligned and with a misaligned output, bf16 -> uint4 / uint2) and
    dequantize (bf16 SET / ADD with their double-rounding tails, the uint2 -> fp32 ADD tail that stores, uint8 -> fp32 ADD) against the oracle's
    threaded reference form.  The salt sits everywhere and, densely, around every partition boundary."""
    import piquant

    ctx = piquant.Context(threads)
    rng = np.random.default_rng(1000 + threads)
    n = N1 + 3     # ragged for the packed types
    x = rng.uniform(-1.2, 1.2, n).astype(np.float32)
    x[rng.choice(n, n // 50)] = np.float32(0.49999997)
    x[rng.choice(n, n // 60)] = np.float32(-0.49999997)
    x[rng.choice(n, n // 70)] = np.float32(8388609.0)
    edges = np.unique(np.clip(((n * np.arange(0, threads + 1) // threads)[:, None] + np.arange(-300, 40)[None, :]).ravel(), 0, n - 1))
    x[edges] = rng.choice(np.array([0.49999997, -0.49999997, 8388609.0, 0.3], np.float32), edges.size)
    xb = O.f32_to_bf16(x)
    for dt_in, xin, dt_out, off in ((O.F32, x, O.UINT8, 0), (O.F32, x, O.UINT8, 5), (O.F32, x, O.UINT4, 0), (O.BF16, xb, O.UINT8, 0), (O.BF16, xb, O.UINT4, 0),
                                    (O.BF16, xb, O.UINT2, 0)):
        want = _ref_quantize(O, xin, dt_in, dt_out, 1.0, 1, threads, off)
        got = gpu_quantize(ctx, xin, dt_in, dt_out, 1.0, 1, offset_out=off)
        assert np.array_equal(got, want), (threads, dt_in, dt_out, off, np.nonzero(got != want)[0][:8])
        if dt_in == O.F32:
            assert not np.array_equal(want, O.quantize(xin, dt_in, dt_out, 1.0, 1, form=O.FORM_UNIFORM)), "the salt never met a scalar position"
    q8 = rng.integers(0, 256, n, dtype=np.uint8)
    prev = rng.uniform(-3, 3, n).astype(np.float32)
    moved = 0
    for dt_q, dt_f, op in ((O.UINT4, O.BF16, 0), (O.UINT4, O.BF16, 1), (O.UINT8, O.BF16, 1), (O.UINT2, O.BF16, 0), (O.UINT2, O.BF16, 1), (O.UINT2, O.F32, 1),
                           (O.UINT8, O.F32, 1), (O.UINT4, O.F32, 0)):
        q = q8[: O.packed_numel(n, dt_q)]
        pv = prev if dt_f == O.F32 else O.f32_to_bf16(prev)
        want = O.dequantize(q, dt_q, dt_f, n, 0.3, 2, op, form=O.FORM_REFERENCE, threads=threads, out=pv.copy())
        got = gpu_dequantize(ctx, q, dt_q, dt_f, n, 0.3, 2, op, prev=pv.copy())
        assert same_floats(got, want), (threads, dt_q, dt_f, op, np.nonzero(got != want)[0][:8])
        moved += int(not same_floats(want, O.dequantize(q, dt_q, dt_f, n, 0.3, 2, op, form=O.FORM_UNIFORM, out=pv.copy())))
    assert moved >= (3 if threads > 1 else 1), "no dequantize tail differed from the SIMD-body form: nothing was tested"   # one partition: one tail per call


@pytest.mark.parametrize("threads", [1, 255])
def test_large_tensors_take_another_tile_and_write_the_same_bytes(O, threads):
    """Beyond 2^26 (uint4) / 2^25 (uint2) elements the two sub-byte -> bf16 SET dequantizers switch tile and store policy (csrc/tuning.hpp,
    kDequantTuneLarge*): the bytes are the oracle's threaded reference form on both sides of the thresholds, aligned and with the output one element off."""
    import piquant

    ctx = piquant.Context(threads)
    rng = np.random.default_rng(2600 + threads)
    for dt_q, threshold in ((O.UINT4, 1 << 26), (O.UINT2, 1 << 25)):
        for n, off in ((threshold + 4099, 0), (threshold + 4099, 2), (threshold - 1, 0)):   # off: bytes -- one bf16 element, the body then starts inside a packed byte
            q = rng.integers(0, 256, O.packed_numel(n, dt_q), dtype=np.uint8)
            want = O.dequantize(q, dt_q, O.BF16, n, 0.3, 2, 0, form=O.FORM_REFERENCE, threads=threads)
            got = gpu_dequantize(ctx, q, dt_q, O.BF16, n, 0.3, 2, 0, offset_out=off)
            assert same_floats(got, want), (threads, dt_q, n, off, np.nonzero(got != want)[0][:8])


def test_small_tensors_and_more_threads_than_simd_blocks(O):
    """Partitions much smaller than a wave tile -- down to contexts with more pool threads than the tensor has elements -- take the element-by-element
    second look inside the same launch: still the reference's bytes."""
    import piquant

    rng = np.random.default_rng(5)
    for threads in (3, 64, 255, 4096):
        ctx = piquant.Context(threads)
        for n in (1, 17, 255, 256, 1000, 4099, 70_001, 300_007):
            x = rng.uniform(-1.2, 1.2, n).astype(np.float32)
            x[rng.choice(n, max(1, n // 3))] = np.float32(0.49999997)
            x[rng.choice(n, max(1, n // 5))] = np.float32(8388609.0)
            xb = O.f32_to_bf16(x)
            for dt_in, xin, dt_out, off in ((O.F32, x, O.UINT8, 0), (O.F32, x, O.UINT8, 3), (O.F32, x, O.UINT4, 0), (O.BF16, xb, O.UINT2, 0)):
                want = _ref_quantize(O, xin, dt_in, dt_out, 1.0, 1, threads, off)
                got = gpu_quantize(ctx, xin, dt_in, dt_out, 1.0, 1, offset_out=off)
                assert np.array_equal(got, want), (threads, n, dt_in, dt_out, off)
            for dt_q, dt_f, op in ((O.UINT4, O.BF16, 1), (O.UINT8, O.BF16, 1), (O.UINT2, O.BF16, 1), (O.UINT2, O.F32, 1)):
                q = rng.integers(0, 256, O.packed_numel(n, dt_q), dtype=np.uint8)
                pv = rng.uniform(-3, 3, n).astype(np.float32)
                pv = pv if dt_f == O.F32 else O.f32_to_bf16(pv)
                want = O.dequantize(q, dt_q, dt_f, n, 0.3, 2, op, form=O.FORM_REFERENCE, threads=threads, out=pv.copy())
                got = gpu_dequantize(ctx, q, dt_q, dt_f, n, 0.3, 2, op, prev=pv.copy())
                assert same_floats(got, want), (threads, n, dt_q, dt_f, op)


def test_python_default_context_has_the_reference_default_thread_count(O):
    """`piquant.Context()` stands for the reference's default context: cpu_count() - 1 pool threads (python/src/piquant/__init__.py:67-68), and
    piquant.torch's plain calls write that context's bytes."""
    import multiprocessing

    import piquant

    threads = max(multiprocessing.cpu_count() - 1, 1)
    rng = np.random.default_rng(9)
    n = 1_000_003
    x = rng.uniform(-1.2, 1.2, n).astype(np.float32)
    x[rng.choice(n, n // 3)] = np.float32(0.49999997)
    xd = torch.from_numpy(x).cuda()
    q = piquant.torch.quantize(xd, scale=1.0, zero_point=1, dtype=torch.uint8)            # default context of the thread
    want = _ref_quantize(O, x, O.F32, O.UINT8, 1.0, 1, threads, q.data_ptr() % 16)
    assert np.array_equal(q.cpu().numpy(), want)
    acc = O.f32_to_bf16(rng.uniform(-3, 3, n).astype(np.float32))
    q4 = rng.integers(0, 256, (n + 1) // 2, dtype=np.uint8)
    accd = torch.from_numpy(acc.view(np.int16)).cuda().view(torch.bfloat16)
    piquant.torch.dequantize(torch.from_numpy(q4).cuda(), scale=0.3, zero_point=2, dtype=torch.bfloat16, reduce_op="add", out=accd, quant_dtype=torch.quint4x2, shape=(n,))
    want = O.dequantize(q4, O.UINT4, O.BF16, n, 0.3, 2, O.ADD, form=O.FORM_REFERENCE, threads=threads, out=acc.copy())
    assert np.array_equal(accd.view(torch.int16).cpu().numpy().view(np.uint16), want)


def test_additive_calls_stay_position_independent(O):
    """In the same default contexts: `uniform=True`, the device-parameter twins, the one-launch call and the shard functions compute the SIMD-body
    form at every element -- halves concatenate to the whole, whatever num_threads says."""
    import piquant
    import piquant.distributed as D

    rng = np.random.default_rng(13)
    n = 2_000_003
    x = rng.uniform(-1.0, 1.0, n).astype(np.float32)
    x[rng.choice(n, n // 3)] = np.float32(0.49999997)
    xd = torch.from_numpy(x).cuda()
    uniform = O.quantize(x, O.F32, O.UINT8, 1.0, 1, form=O.FORM_UNIFORM)
    for threads in (1, 7):
        ctx = piquant.Context(threads)
        assert not np.array_equal(piquant.torch.quantize(xd, scale=1.0, zero_point=1, dtype=torch.uint8, ctx=ctx).cpu().numpy(), uniform)
        assert np.array_equal(piquant.torch.quantize(xd, scale=1.0, zero_point=1, dtype=torch.uint8, ctx=ctx, uniform=True).cpu().numpy(), uniform)
        whole = torch.zeros(n, dtype=torch.uint8, device="cuda")
        for r in range(3):
            D.quantize_shard(xd, scale=1.0, zero_point=1, dtype=torch.uint8, out=whole, rank=r, world_size=3, ctx=ctx)
        assert np.array_equal(whole.cpu().numpy(), uniform)
        qd, rec = piquant.torch.quantize_dynamic(xd, dtype=torch.uint8, ctx=ctx)
        s, z = piquant.torch.params_to_host(rec)
        assert np.array_equal(qd.cpu().numpy(), O.quantize(x, O.F32, O.UINT8, s, z, form=O.FORM_UNIFORM))
        # bf16 ADD: the plain call double-rounds in the partitions' tails, the uniform one nowhere
        q4 = rng.integers(0, 256, (n + 1) // 2, dtype=np.uint8)
        acc = O.f32_to_bf16(rng.uniform(-3, 3, n).astype(np.float32))
        accd = torch.from_numpy(acc.view(np.int16)).cuda().view(torch.bfloat16)
        piquant.torch.dequantize(torch.from_numpy(q4).cuda(), scale=0.3, zero_point=2, dtype=torch.bfloat16, reduce_op="add", out=accd, quant_dtype=torch.quint4x2,
                                 shape=(n,), ctx=ctx, uniform=True)
        want = O.dequantize(q4, O.UINT4, O.BF16, n, 0.3, 2, O.ADD, form=O.FORM_UNIFORM, out=acc.copy())
        assert np.array_equal(accd.view(torch.int16).cpu().numpy().view(np.uint16), want)


def test_the_environment_can_switch_the_layout_off():
    """PIQUANT_HIP_REFERENCE_LAYOUT=0 at context creation: position-independent output from the plain calls (what rounds 1-5 did by default)."""
    code = textwrap.dedent(f"""
        import sys, numpy as np
        sys.path.insert(0, {str(os.path.join(os.path.dirname(__file__), '..', 'pi-quant_amd'))!r}); sys.path.insert(0, {str(os.path.join(os.path.dirname(__file__), '..'))!r})
        import torch, oracle as O
        from piquant._bootstrap import C_LIB as C
        x = np.full(1000, 0.49999997, dtype=np.float32)
        xd = torch.from_numpy(x).cuda()
        ctx = C.piquant_context_create(3)
        out = torch.zeros(1000, dtype=torch.uint8, device='cuda')
        C.piquant_quantize(ctx, xd.data_ptr(), 0, out.data_ptr(), 4, 1000, 1.0, 0, 0)
        assert np.array_equal(out.cpu().numpy(), O.quantize(x, O.F32, O.UINT8, 1.0, 0, form=O.FORM_UNIFORM)) and bool((out == 1).all())
        print('uniform ok')
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, PIQUANT_HIP_REFERENCE_LAYOUT="0"))
    assert r.returncode == 0 and "uniform ok" in r.stdout, r.stderr[-2000:]
