"""GPU: the reference's own C++ test suites re-stated against the HIP path, with the reference's data distribution,
sizes and tolerances (the bit-exact oracle comparisons live in test_gpu_parity.py; these show a reference user that the
checks they know still hold):

  test/quant.cpp:29-122   quantize vs the naive scalar formula: |dq| <= 1 for uint8, exact nibbles for uint4 nearest
  test/dequant.cpp:18-89  params -> quantize -> dequantize round trip, 24 cases, abs err 0.05 / 0.2 / 2.0
  test/quant_config.cpp   scale > 0 and finite, quantize runs, 12 cases
  test/quant.cpp:198-217  identity requant of a constant tensor
"""
import numpy as np
import pytest

from helpers import gpu_dequantize, gpu_quantize

pytestmark = pytest.mark.gpu

ITERS = 10


@pytest.fixture(scope="module")
def ctx():
    import piquant

    return piquant.Context()


def naive_quant(x, scale, zp, qmax, O):
    """test/naive.hpp:52-96, nearest: clamp(int64(std::round(x * (1/scale))) + zp, 0, qmax)."""
    inv = np.float32(1.0) / np.float32(scale)
    p = x.astype(np.float32) * inv
    r = np.where(p >= 0, np.floor(p + np.float32(0.5)), np.ceil(p - np.float32(0.5)))   # round half away from zero on exact fp32 values
    exact_round = np.sign(p) * np.floor(np.abs(p.astype(np.float64)) + 0.5)
    return np.clip(exact_round.astype(np.int64) + zp, 0, qmax)


@pytest.mark.parametrize("dt_in", [0, 1], ids=["f32", "bf16"])
@pytest.mark.parametrize("dt_out,qmax,zlo,zhi", [(4, 255, -128, 127), (3, 15, -8, 7)], ids=["u8", "u4"])
@pytest.mark.parametrize("rm", [0, 1], ids=["nearest", "stochastic"])
def test_quantize_vs_naive(ctx, oracle_mod, dt_in, dt_out, qmax, zlo, zhi, rm):
    O = oracle_mod
    rng = np.random.default_rng(0x9032002)
    for _ in range(ITERS):
        scale = float(np.float32(rng.uniform(0.1, 1.0)))
        zp = int(rng.integers(zlo, zhi + 1))
        n = int(rng.integers(5000, 15001))
        x = rng.uniform(-1, 1, n).astype(np.float32)
        xin = x if dt_in == 0 else O.f32_to_bf16(x)
        xf = x if dt_in == 0 else O.bf16_to_f32(xin)
        ctx.set_stochastic_threshold(None)
        got = gpu_quantize(ctx, xin, dt_in, dt_out, scale, zp, rm)
        if dt_out == 3:
            got = np.stack([got & 15, got >> 4], axis=1).reshape(-1)[:n]
        want = naive_quant(xf, scale, zp, qmax, O)
        d = np.abs(got.astype(np.int64) - want)
        if rm == 0 and dt_out == 3:
            assert d.max() == 0                      # test/quant.cpp:95-96: exact nibble equality
        else:
            assert d.max() <= 1                      # test/quant.cpp:16,52-53: stochastic_epsilon


@pytest.mark.parametrize("dt_f", [0, 1], ids=["f32", "bf16"])
@pytest.mark.parametrize("dt_q,eps", [(2, 2.0), (3, 0.2), (4, 0.05)], ids=["u2", "u4", "u8"])
@pytest.mark.parametrize("rm", [0, 1], ids=["nearest", "stochastic"])
@pytest.mark.parametrize("op", [0, 1], ids=["set", "add"])
def test_dequantize_round_trip(ctx, oracle_mod, dt_f, dt_q, eps, rm, op):
    import piquant
    import torch

    O = oracle_mod
    rng = np.random.default_rng(0x9032002)
    tq = {2: torch.quint2x4, 3: torch.quint4x2, 4: torch.quint8}[dt_q]
    for _ in range(ITERS):
        n = int(rng.integers(5000, 15001))
        x = rng.uniform(-1, 1, n).astype(np.float32)
        xin = x if dt_f == 0 else O.f32_to_bf16(x)
        xf = x if dt_f == 0 else O.bf16_to_f32(xin)
        xd = torch.from_numpy(xin).cuda() if dt_f == 0 else torch.from_numpy(xin.view(np.int16)).cuda().view(torch.bfloat16)
        scale, zp = piquant.torch.compute_quant_params(xd, dtype=tq)
        ctx.set_stochastic_threshold(None)
        q = gpu_quantize(ctx, xin, dt_f, dt_q, scale, zp, rm)
        prev_v = np.float32(rng.uniform(-1, 1)) if op else np.float32(0.0)
        prev = np.full(n, prev_v, np.float32)
        prev_in = prev if dt_f == 0 else O.f32_to_bf16(prev)
        back = gpu_dequantize(ctx, q, dt_q, dt_f, n, scale, zp, op, prev=prev_in)
        backf = back if dt_f == 0 else O.bf16_to_f32(back)
        prevf = prev_v if dt_f == 0 else O.bf16_to_f32(prev_in[:1])[0]
        assert np.abs(xf - (backf - prevf)).max() <= eps


@pytest.mark.parametrize("dt_f", [0, 1], ids=["f32", "bf16"])
@pytest.mark.parametrize("tq_name", ["quint2x4", "quint4x2", "quint8"])
@pytest.mark.parametrize("rm", ["nearest", "stochastic"])
def test_quantize_range(dt_f, tq_name, rm):
    import piquant
    import torch

    rng = np.random.default_rng()
    for _ in range(ITERS):
        n = int(rng.integers(5000, 15001))
        x = torch.from_numpy(rng.uniform(-1, 1, n).astype(np.float32)).cuda()
        if dt_f:
            x = x.to(torch.bfloat16)
        scale, zp = piquant.torch.compute_quant_params(x, dtype=getattr(torch, tq_name))
        assert scale > 0.0 and np.isfinite(scale)
        q = piquant.torch.quantize(x, scale=scale, zero_point=zp, dtype=getattr(torch, tq_name), round_mode=rm)
        assert q.shape == x.shape and q.is_cuda
    torch.cuda.synchronize()


def test_concurrent_contexts_from_threads(oracle_mod):
    """Contexts are independent: four host threads, each with its own context and stream, hammer the library at once."""
    import threading

    import piquant
    import torch

    O = oracle_mod
    errors = []

    def worker(seed):
        try:
            rng = np.random.default_rng(seed)
            c = piquant.Context()
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(20):
                    n = int(rng.integers(100_000, 400_000))
                    x = rng.uniform(-1, 1, n).astype(np.float32)
                    xd = torch.from_numpy(x).cuda()
                    scale, zp = piquant.torch.compute_quant_params(xd, dtype=torch.quint8, ctx=c)
                    q = piquant.torch.quantize(xd, scale=scale, zero_point=zp, dtype=torch.uint8, ctx=c)
                    s.synchronize()
                    assert (scale, zp) == O.compute_quant_params(x, 0, 4)
                    assert np.array_equal(q.cpu().numpy(), O.quantize(x, 0, 4, scale, zp))
        except Exception as exc:   # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
