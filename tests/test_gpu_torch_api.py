"""GPU: the tensor-level API, mirroring the reference's own Python tests (python/tests/test_torch.py:23-53)
with the same seed, value range and tolerances -- but on ROCm device tensors.  torch.quantize_per_tensor
runs on the CPU copy as the independent comparison the reference test uses."""
import math
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TORCH_FLOAT_TYPES = [torch.bfloat16, torch.float32]
TORCH_QUANT_TYPES = [torch.quint8, torch.quint4x2, torch.quint2x4]

random.seed(128)
gen = torch.manual_seed(128)


def shape4():
    while True:   # reference: four dims in [1, 128]; bounded here so the CPU-side torch comparison stays quick
        s = [random.randint(1, 128) for _ in range(4)]
        if s[0] * s[1] * s[2] * s[3] <= 24_000_000:
            return s


@pytest.mark.parametrize("dtype_in", TORCH_FLOAT_TYPES)
@pytest.mark.parametrize("dtype_quantized", TORCH_QUANT_TYPES)
def test_compute_quant_config(dtype_in, dtype_quantized):
    import piquant

    tensor = torch.empty(*shape4(), dtype=dtype_in)
    tensor.uniform_(-1.0, 1.0, generator=gen)
    scale, zero_point = piquant.torch.compute_quant_params(tensor.cuda(), dtype=dtype_quantized)
    assert scale > 0
    assert not math.isnan(scale)
    assert not math.isinf(scale)
    assert isinstance(zero_point, int)


@pytest.mark.parametrize("dtype_in", TORCH_FLOAT_TYPES)
@pytest.mark.parametrize("dtype_quantized", TORCH_QUANT_TYPES)
def test_quantize_roundtrip(dtype_in, dtype_quantized):
    import piquant

    inp = torch.empty(*shape4(), dtype=dtype_in)
    inp.uniform_(-1.0, 1.0, generator=gen)
    dev = inp.cuda()
    scale, zero_point = piquant.torch.compute_quant_params(dev, dtype=dtype_quantized)
    quantized_torch = torch.quantize_per_tensor(inp.float(), scale=scale, zero_point=zero_point, dtype=dtype_quantized)
    quantized_pi = piquant.torch.quantize(dev, zero_point=zero_point, scale=scale, dtype=dtype_quantized)
    assert quantized_pi.is_cuda

    dequantized_torch = quantized_torch.dequantize().to(dtype_in)
    dequantized_pi = piquant.torch.dequantize(quantized_pi, scale=scale, zero_point=zero_point, dtype=dtype_in)
    assert dequantized_pi.is_cuda and dequantized_pi.shape == inp.shape
    dequantized_pi = dequantized_pi.cpu()
    assert dequantized_torch.dtype == dequantized_pi.dtype
    assert dequantized_pi.dtype == inp.dtype
    # reference tolerances (test_torch.py:51-53).  torch rounds x/scale half-to-even with a true division, pi-quant
    # (reference and this build, bit-identically) rounds x*(1/scale) half-away-from-zero; on tensors of millions
    # of elements a handful land on opposite sides of a rounding boundary and differ by exactly one quantum, so
    # the reference's allclose(atol=1e-3) is applied to all but a <= 1e-5 fraction, which may differ by one step.
    diff = (dequantized_torch.float() - dequantized_pi.float()).abs()
    step = scale + (2.0 ** -7 if dtype_in == torch.bfloat16 else 0.0) + 1e-3
    assert float(diff.max()) <= step
    assert float((diff > 1e-3).float().mean()) <= 1e-5
    assert torch.allclose(dequantized_torch, inp, atol=scale * 0.5 + 1e-3)
    assert torch.allclose(dequantized_pi, inp, atol=scale * 0.5 + 1e-3)


@pytest.mark.parametrize("host_path", ["auto", "stage"])
def test_cpu_tensors_round_trip(host_path):
    """A reference user passing CPU tensors keeps working: same call; the host tensors are served where they live by default (the companion
    library) or staged over PCIe through the HIP kernels when the context says so."""
    import piquant

    ctx = piquant.Context()
    ctx.set_host_path(host_path)
    x = torch.empty(257, 33).uniform_(-1, 1, generator=gen)
    scale, zp = piquant.torch.compute_quant_params(x, dtype=torch.quint8, ctx=ctx)
    q = piquant.torch.quantize(x, scale=scale, zero_point=zp, dtype=torch.quint8, ctx=ctx)
    assert q.dtype == torch.quint8 and not q.is_cuda and q.shape == x.shape
    ref = torch.quantize_per_tensor(x, scale=scale, zero_point=zp, dtype=torch.quint8)
    assert torch.equal(q.int_repr(), ref.int_repr()) or (q.int_repr().int() - ref.int_repr().int()).abs().max() <= 1
    back = piquant.torch.dequantize(q, scale=scale, zero_point=zp, dtype=torch.float32, ctx=ctx)
    assert torch.allclose(back, x, atol=scale * 0.5 + 1e-6)


def test_add_store_needs_and_uses_an_accumulator():
    import piquant

    x = torch.empty(10_000, device="cuda").uniform_(-1, 1)
    scale, zp = piquant.torch.compute_quant_params(x, dtype=torch.uint8)
    q = piquant.torch.quantize(x, scale=scale, zero_point=zp, dtype=torch.uint8)
    with pytest.raises(ValueError):
        piquant.torch.dequantize(q, scale=scale, zero_point=zp, dtype=torch.float32, reduce_op="add")
    acc = torch.full_like(x, 5.0)
    piquant.torch.dequantize(q, scale=scale, zero_point=zp, dtype=torch.float32, reduce_op="add", out=acc)
    assert torch.allclose(acc - 5.0, x, atol=scale * 0.5 + 1e-5)


def test_identity_requant_of_constant_tensor():
    """reference test/quant.cpp:198-217: const 42 -> params (1.0, 127) -> quant -> dequant(ADD into zeros) == 42."""
    import piquant

    x = torch.full((8192,), 42.0, device="cuda")
    scale, zp = piquant.torch.compute_quant_params(x, dtype=torch.uint8)
    assert (scale, zp) == (1.0, 127)
    q = piquant.torch.quantize(x, scale=scale, zero_point=zp, dtype=torch.uint8)
    acc = torch.zeros_like(x)
    piquant.torch.dequantize(q, scale=scale, zero_point=zp, dtype=torch.float32, reduce_op="add", out=acc)
    assert float((acc - 42.0).abs().max()) <= 1e-6


def test_non_contiguous_input_is_made_contiguous():
    import piquant

    x = torch.empty(300, 70, device="cuda").uniform_(-1, 1).t()
    assert not x.is_contiguous()
    q = piquant.torch.quantize(x, scale=0.0078431377, zero_point=127, dtype=torch.uint8)
    q2 = piquant.torch.quantize(x.contiguous(), scale=0.0078431377, zero_point=127, dtype=torch.uint8)
    assert torch.equal(q, q2) and q.shape == x.shape


def test_native_front_end_matches_the_ctypes_path():
    """piquant.torch.quantize / dequantize ARE the entry points of csrc/torch_binding.cpp when it is built (CPython vectorcall functions); the Python
    implementation behind them -- with the native allocation path and, with the module switched off, through ctypes -- must give the same tensors
    (dtype, shape, device, bytes), including out=, ADD and non-contiguous input."""
    import piquant
    import piquant.torch as pt

    if pt._native is None:
        pytest.skip("native front end not built")
    native = pt._native
    assert pt.quantize is native.quantize_entry and pt.dequantize is native.dequantize_entry
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.empty(257, 129, device="cuda").uniform_(-2, 3, generator=g)
    as_int = lambda t: t.view(torch.int16 if t.dtype == torch.bfloat16 else torch.int32)   # noqa: E731

    def through(quantize, dequantize, src, scale, zp, qdt):
        q = quantize(src, scale=scale, zero_point=zp, dtype=qdt)
        d = dequantize(q, scale=scale, zero_point=zp, dtype=src.dtype)
        acc = torch.ones_like(d)
        r = dequantize(q, scale=scale, zero_point=zp, dtype=src.dtype, reduce_op="add", out=acc)
        assert r is acc
        return q, d, acc

    try:
        for src in (x, x.to(torch.bfloat16), x.t()):
            for qdt in (torch.quint8, torch.uint8, torch.quint4x2, torch.quint2x4):
                scale, zp = pt.compute_quant_params(src.contiguous(), dtype=qdt)
                pt._native = native
                q_e, d_e, acc_e = through(pt.quantize, pt.dequantize, src, scale, zp, qdt)            # the entry points
                q_n, d_n, acc_n = through(pt._quantize_py, pt._dequantize_py, src, scale, zp, qdt)    # Python wrapper, native call
                pt._native = None
                q_c, d_c, acc_c = through(pt._quantize_py, pt._dequantize_py, src, scale, zp, qdt)    # Python wrapper, ctypes
                for q, d, acc in ((q_e, d_e, acc_e), (q_n, d_n, acc_n)):
                    assert q.dtype == q_c.dtype == qdt and q.shape == q_c.shape == src.shape and q.device == src.device
                    assert torch.equal(pt.packed_bytes(q), pt.packed_bytes(q_c))
                    assert d.dtype == src.dtype and torch.equal(as_int(d), as_int(d_c)) and torch.equal(acc, acc_c)
        # mixing the paths on one context: stream / blocking caches must stay coherent
        pt._native = native
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            a = pt.quantize(x, scale=0.02, zero_point=100, dtype=torch.quint8)
            pt._native = None
            b = pt._quantize_py(x, scale=0.02, zero_point=100, dtype=torch.quint8)
        pt._native = native
        c = pt._quantize_py(x, scale=0.02, zero_point=100, dtype=torch.quint8)
        torch.cuda.synchronize()
        assert torch.equal(pt.packed_bytes(a), pt.packed_bytes(b)) and torch.equal(pt.packed_bytes(a), pt.packed_bytes(c))
    finally:
        pt._native = native


def test_entry_points_take_what_the_python_implementation_takes_and_raise_what_it_raises():
    """The vectorcall entry points serve the common call themselves and hand everything else, arguments untouched, to the Python implementation:
    numpy and tensor scalars, ctx=, out= identity, uniform=, and every refusal with the exception type of the Python path."""
    import numpy as np

    import piquant
    import piquant.torch as pt

    if pt._native is None:
        pytest.skip("native front end not built")
    x = torch.linspace(-1, 1, 4099, device="cuda")
    want = pt.packed_bytes(pt._quantize_py(x, scale=0.01, zero_point=100, dtype=torch.quint8))
    for scale, zp in ((0.01, 100), (np.float32(0.01).item(), np.int64(100)), (np.float64(0.01), 100), (torch.tensor(0.01, dtype=torch.float64), torch.tensor(100))):
        got = pt.quantize(x, scale=scale, zero_point=zp, dtype=torch.quint8)
        assert got.dtype == torch.quint8 and torch.equal(pt.packed_bytes(got), want), (type(scale), type(zp))
    assert torch.equal(pt.packed_bytes(pt.quantize(x, scale=1, zero_point=100, dtype=torch.quint8)),      # an int for scale
                       pt.packed_bytes(pt._quantize_py(x, scale=1, zero_point=100, dtype=torch.quint8)))
    out = torch.empty(x.shape, dtype=torch.quint8, device="cuda")
    assert pt.quantize(x, scale=0.01, zero_point=100, dtype=torch.quint8, out=out) is out and torch.equal(pt.packed_bytes(out), want)
    assert torch.equal(pt.packed_bytes(pt.quantize(x, scale=0.01, zero_point=100, dtype=torch.quint8, uniform=True)), want)   # ordinary data: same bytes
    ctx = piquant.Context(num_threads=3)
    assert torch.equal(pt.packed_bytes(pt.quantize(x, scale=0.01, zero_point=100, dtype=torch.quint8, ctx=ctx)), want)
    assert torch.equal(pt.packed_bytes(pt.quantize(x, scale=0.01, zero_point=100, dtype=torch.quint8, round_mode="nearest", ctx=None, out=None)), want)
    q = pt.quantize(x, scale=0.01, zero_point=100, dtype=torch.quint8)
    d = pt.dequantize(q, scale=0.01, zero_point=100, dtype=torch.float32, reduce_op="set")
    assert torch.equal(d, pt._dequantize_py(q, scale=0.01, zero_point=100, dtype=torch.float32))
    raw = pt.packed_bytes(q)
    assert torch.equal(pt.dequantize(raw, scale=0.01, zero_point=100, dtype=torch.float32, quant_dtype=torch.quint8, shape=x.shape), d)
    # refusals: the same exception type from the entry point and from the Python implementation
    cases = [
        (pt.quantize, pt._quantize_py, (x,), dict(scale=0.01, zero_point=100, dtype=torch.float32)),                     # not a quantized dtype
        (pt.quantize, pt._quantize_py, (x,), dict(scale=0.01, zero_point=100, dtype=torch.quint8, round_mode="up")),      # unknown mode name
        (pt.quantize, pt._quantize_py, (x,), dict(scale=0.01, zero_point=100, dtype=torch.quint8, bogus=1)),              # unknown keyword
        (pt.quantize, pt._quantize_py, (x,), dict(scale=0.01, dtype=torch.quint8)),                                      # missing keyword
        (pt.quantize, pt._quantize_py, (x, 0.01), dict(zero_point=100, dtype=torch.quint8)),                             # positional scale
        (pt.quantize, pt._quantize_py, (x.to(torch.int32),), dict(scale=0.01, zero_point=100, dtype=torch.quint8)),       # not a float tensor
        (pt.dequantize, pt._dequantize_py, (q,), dict(scale=0.01, zero_point=100, dtype=torch.quint8)),                  # not a float dtype
        (pt.dequantize, pt._dequantize_py, (q,), dict(scale=0.01, zero_point=100, dtype=torch.float32, reduce_op="add")),  # ADD without an accumulator
        (pt.dequantize, pt._dequantize_py, (q,), dict(scale=0.01, zero_point=100, dtype=torch.float32, reduce_op="mul")),  # unknown operator name
        (pt.dequantize, pt._dequantize_py, (q,), dict(scale=0.01, zero_point=100, dtype=torch.float32, out=torch.empty(3, device="cuda"))),   # out= of the wrong size
    ]
    for entry, py, args, kw in cases:
        with pytest.raises(Exception) as e_py:
            py(*args, **kw)
        with pytest.raises(Exception) as e_entry:
            entry(*args, **kw)
        assert type(e_entry.value) is type(e_py.value), (kw, e_entry.value, e_py.value)


def test_reference_style_benchmark_harness_runs(tmp_path):
    """the reference's two benchmark scripts re-stated on device tensors (python/benchmark/benchmark.py:16-72,
    throughput_avg.py:9-42) must keep running: short run, schema of the JSON document, the scripts' own agreement check with
    torch.quantize_per_tensor, and the bar chart when matplotlib is there."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    png = tmp_path / "quant_benchmark.png"
    r = subprocess.run([sys.executable, str(root / "tools" / "reference_style_benchmarks.py"), "--gib", "0.25", "--runs", "50", "--plot", str(png)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout)
    assert d["runs"] == 50 and d["device"]
    bars = d["benchmark_py (NUMEL=1e6, 50 runs)"]
    assert [b["dtype"] for b in bars] == ["quint8", "quint4x2", "quint2x4"]
    for b in bars:
        # ties (torch: half to even, the reference and this library: half away from zero) may put a handful of the 10^6 elements one step apart
        assert b["elements_beyond_1e-1"] < 100 and b["torch_s_per_50"] > 0 and b["piquant_s_per_50"] > 0
    assert bars[0]["results_allclose_1e-1"] or bars[0]["elements_beyond_1e-1"] == 0   # a quint8 step (1/255) is below the tolerance
    head = d["README headline (numel=27264000, seconds per 50 runs)"]
    assert head["piquant.torch.quantize (device)"] > 0 and head["elements_differing_from_torch"] < 27_264_000 // 1000
    thr = d["throughput_avg_py (0.25 GiB float tensor, 10 iterations, allocation + sync inside the timed call)"]
    assert len(thr) == 3 and all(t["quantize_GiB/s_of_float_input"] > 0 and t["dequantize_GiB/s_of_float_output"] > 0 for t in thr)
    try:
        import matplotlib  # noqa: F401
    except ImportError:
        assert d["plot"].startswith("matplotlib is not importable")
    else:
        assert d["plot"] == str(png) and png.stat().st_size > 5000


def test_default_context_is_per_thread_and_device():
    """the shared default context carried the stream of the call being made; two threads on two streams could swap it between
    set_stream and the call.  The default context is now per (thread, device)."""
    import threading

    import piquant

    main = piquant.Context.get()
    assert piquant.Context.get() is main
    seen = {}

    def work(i):
        torch.cuda.set_device(0)
        a = piquant.Context.get()
        b = piquant.Context.get(0)
        x = torch.full((1000,), float(i + 1), device="cuda")
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            q = piquant.torch.quantize(x, scale=0.5, zero_point=3, dtype=torch.uint8)
        s.synchronize()
        seen[i] = (a, b, int(q[0]))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert seen[0][0] is seen[0][1] and seen[1][0] is seen[1][1]
    assert seen[0][0] is not seen[1][0] and seen[0][0] is not main and seen[1][0] is not main
    assert (seen[0][2], seen[1][2]) == (5, 7)      # 1 / 0.5 + 3, 2 / 0.5 + 3


def test_out_buffers_are_checked_by_the_tensor_not_by_its_storage():
    """a short out= slice of a large buffer used to pass (the storage was large enough) and the kernel wrote past the slice."""
    import piquant

    x = torch.empty(1000, device="cuda").uniform_(-1, 1)
    big = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    with pytest.raises((ValueError, RuntimeError)):
        piquant.torch.quantize(x, scale=0.01, zero_point=128, dtype=torch.uint8, out=big[:999])
    with pytest.raises((ValueError, RuntimeError)):
        piquant.torch.quantize(x, scale=0.01, zero_point=8, dtype=torch.quint4x2, out=big[:499])
    piquant.torch.quantize(x, scale=0.01, zero_point=8, dtype=torch.quint4x2, out=big[:500])           # exactly enough
    rec = torch.empty(16, dtype=torch.uint8, device="cuda")
    for bad in (dict(out=big[:999]), dict(params=rec[:8]), dict(params=torch.empty(16, dtype=torch.uint8)), dict(out=torch.zeros(1000, dtype=torch.uint8))):
        with pytest.raises(ValueError):
            piquant.torch.quantize_dynamic(x, dtype=torch.uint8, **bad)
    q, r = piquant.torch.quantize_dynamic(x, dtype=torch.uint8)
    with pytest.raises(ValueError):
        piquant.torch.dequantize_dynamic(q, r, dtype=torch.float32, out=torch.empty(999, device="cuda"))
    with pytest.raises(ValueError):
        piquant.torch.dequantize_sum([q, q[:10]], [r, r], dtype=torch.float32)
    with pytest.raises(ValueError):
        piquant.torch.reduce_quantize_dynamic(x.clone(), [q[:999]], [r], dtype=torch.uint8)
    assert bool((big[500:] == 0).all())


def test_blocking_call_in_a_capture_aborts_with_a_message_instead_of_hanging():
    import subprocess
    import sys
    import textwrap
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    code = textwrap.dedent(f"""
        import sys; sys.path.insert(0, {str(root / 'pi-quant_amd')!r})
        import torch, piquant
        x = torch.zeros(4096, device='cuda'); q = torch.zeros(4096, dtype=torch.uint8, device='cuda')
        ctx = piquant.Context(); s = torch.cuda.Stream(); ctx.set_stream(s.cuda_stream); ctx.set_blocking(True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            ctx.quantize_ptr(x.data_ptr(), piquant.DataType.F32, q.data_ptr(), piquant.DataType.UINT8, 4096, 1.0, 0, piquant.RoundMode.NEAREST, _device_ptrs=True)
        print('survived')
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == -6 and "cannot be captured" in r.stderr and "survived" not in r.stdout, (r.returncode, r.stderr[-400:])
