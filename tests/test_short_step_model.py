"""CPU: the arithmetic behind the kernels' SHORT quantization steps (csrc/quant_kernels.hpp quantize_vec_bounded and
quantize_vec_bounded_stochastic), restated in numpy float32 -- IEEE single precision, the same roundings the gfx950 instructions
perform (v_pk_mul_f32, v_pk_add_f32, v_trunc_f32, v_med3_f32, v_cvt_i32_f32, v_cvt_pk_u8_f32 = round-to-nearest-even + saturate,
NaN -> 0) -- against the oracle's restatement of the reference's int32 / int64 steps (kernels_specialized.inl:62-77, quantize.inl:8-26).

The short steps are only taken where the kernels' range condition holds (0 <= zero point <= qmax and max|x| * |1/scale| < 1e9), so
that is where the two must give the same integers; the GPU suite checks the instructions, this file checks the mathematics, on every
machine."""
import numpy as np
import pytest

QMAX = {4: 255, 3: 15, 2: 3}   # oracle dtype code -> qmax


def _pack(q, dt_out):
    """reference packing: element i of a byte in bits [i * bits, (i + 1) * bits) (kernels_specialized.inl:648-651, quantize.inl:73-99)"""
    bits = {4: 8, 3: 4, 2: 2}[dt_out]
    per = 8 // bits
    q = np.concatenate([q.astype(np.uint8), np.zeros((-q.size) % per, dtype=np.uint8)]).reshape(-1, per)
    out = np.zeros(q.shape[0], dtype=np.uint8)
    for k in range(per):
        out |= q[:, k] << np.uint8(k * bits)
    return out


def _finish(s, zp, dt_out):
    """s: integer-valued float32 (or NaN).  8-bit: sat_u8(rne(s + float(zp))) as v_cvt_pk_u8_f32; 4/2-bit: int(med3(s, -zp, qmax - zp)) + zp."""
    qmax = QMAX[dt_out]
    with np.errstate(invalid="ignore", over="ignore"):
        if dt_out == 4:
            t = (s + np.float32(zp)).astype(np.float32)
            q = np.where(np.isnan(t), np.float32(0), np.clip(np.rint(t), 0, 255))
            return q.astype(np.uint8)
        lo, hi = np.float32(-zp), np.float32(qmax - zp)
        c = np.where(np.isnan(s), lo, np.minimum(np.maximum(s, lo), hi))     # v_med3_f32: a NaN operand gives min3 of the other two
        return (c.astype(np.int32) + zp).astype(np.uint8)


def short_nearest(x, inv, zp, dt_out, generic=False):
    with np.errstate(invalid="ignore", over="ignore"):
        p = (x * np.float32(inv)).astype(np.float32)
        if generic:   # std::round: half away from zero, exact
            adj = (np.sign(p) * np.floor(np.abs(p).astype(np.float64) + 0.5)).astype(np.float32)
            adj = np.where(np.isnan(p), p, adj)
        else:
            adj = np.trunc((p + np.copysign(np.float32(0.5), p)).astype(np.float32))
    return _pack(_finish(adj.astype(np.float32), zp, dt_out), dt_out)


def short_stochastic(x, inv, zp, dt_out, tau):
    with np.errstate(invalid="ignore", over="ignore"):
        r = (x * np.float32(inv)).astype(np.float32)
        tr = np.trunc(r)
        d = (r - tr).astype(np.float32)
        adj = np.where(np.float32(tau) < np.abs(d), np.copysign(np.float32(1.0), r), np.float32(0.0)).astype(np.float32)
        s = (tr + adj).astype(np.float32)
    return _pack(_finish(s, zp, dt_out), dt_out)


def _values(rng, n, reach):
    """floats whose products with 1/scale stay below `reach`, with the corners the equivalence argument leans on"""
    x = rng.uniform(-300, 300, n).astype(np.float32)
    k = n // 10
    x[:k] = (np.trunc(x[:k]) + np.copysign(np.float32(0.5), x[:k])).astype(np.float32)          # ties
    x[k:2 * k] = np.float32(0.49999997) * np.sign(x[k:2 * k])                                    # p + 0.5 rounds up to 1.0
    x[2 * k:3 * k] = rng.choice(np.array([0.0, -0.0, 16777216.0, -16777217.0, 8388609.0, reach * 0.999, -reach * 0.999, np.nan], dtype=np.float32), k)
    x[3 * k:4 * k] = rng.standard_normal(k).astype(np.float32) * np.float32(1e-3)
    x[4 * k:5 * k] = (rng.integers(-400, 400, k) + rng.choice([0.25, 0.375, 0.75], k)).astype(np.float32)   # exact fractions
    with np.errstate(invalid="ignore"):
        x[np.abs(x) >= np.float32(reach)] = np.float32(reach * 0.5)   # the kernels send such a tile through the long step: outside this file's claim
    return x


@pytest.mark.parametrize("dt_out", [4, 3, 2])
def test_short_nearest_step_equals_the_reference_step_inside_its_range(oracle_mod, dt_out):
    O = oracle_mod
    rng = np.random.default_rng(2026 + dt_out)
    qmax = QMAX[dt_out]
    for scale in (1.0, 0.5, 3.0, 0.0078431377, 1.0e-3):
        inv = np.float32(1.0) / np.float32(scale)
        x = _values(rng, 200_000, 0.9e9 * scale)
        xb = O.bf16_to_f32(O.f32_to_bf16(x))
        for zp in (0, qmax // 2, qmax):
            for dt_in, xin, xf in ((O.F32, x, x), (O.BF16, O.f32_to_bf16(x), xb)):
                want = O.quantize(xin, dt_in, dt_out, scale, zp)
                generic = dt_in == O.F32 and dt_out == 2     # the one nearest pair that takes the generic std::round step (quantize.inl:105-127)
                got = short_nearest(xf, inv, zp, dt_out, generic=generic)
                assert np.array_equal(got, want), (dt_out, scale, zp, dt_in, np.nonzero(got != want)[0][:5])


@pytest.mark.parametrize("dt_out", [4, 3, 2])
def test_short_stochastic_step_equals_the_reference_step_inside_its_range(oracle_mod, dt_out):
    O = oracle_mod
    rng = np.random.default_rng(77 + dt_out)
    qmax = QMAX[dt_out]
    for scale in (1.0, 0.5, 3.0, 0.0078431377):
        inv = np.float32(1.0) / np.float32(scale)
        x = _values(rng, 200_000, 0.9e9 * scale)
        xb = O.bf16_to_f32(O.f32_to_bf16(x))
        for zp in (0, qmax // 2, qmax):
            for tau in (0.0, 0.25, 0.375, 0.37499997, 0.5, 0.99999994):
                for dt_in, xin, xf in ((O.F32, x, x), (O.BF16, O.f32_to_bf16(x), xb)):
                    want = O.quantize(xin, dt_in, dt_out, scale, zp, O.STOCHASTIC, tau)
                    got = short_stochastic(xf, inv, zp, dt_out, tau)
                    assert np.array_equal(got, want), (dt_out, scale, zp, tau, dt_in, np.nonzero(got != want)[0][:5])


def test_oracle_treats_signaling_nans_like_quiet_ones(oracle_mod):
    """The checker's side of tests/test_gpu_parity.py::test_signaling_nans_do_not_poison_range_tests_or_scans: in the oracle a NaN is a NaN --
    skipped by the min/max restatement wherever it sits, quantized to what the reference's indefinite conversion clamps to."""
    O = oracle_mod
    x = np.array([-5.0, 7.0, 1.0, 2.0, 3.0, 0.5, -0.5, 6.0], dtype=np.float32)
    for nan_bits in (0x7f800001, 0xff800001, 0x7fc00000, 0x7fa00000):
        for pos in (2, 5, 7):
            y = x.copy()
            y.view(np.uint32)[pos] = nan_bits
            assert O.minmax(y, O.F32) == (-5.0, 7.0)
            q = O.quantize(y, O.F32, O.UINT8, 1.0, 10)
            assert q[pos] == 0 and q[0] == 5 and q[1] == 17
            q = O.quantize(y, O.F32, O.UINT8, 1.0, 10, O.STOCHASTIC, 0.3)
            assert q[pos] == 0
    yb = O.f32_to_bf16(x)
    yb.view(np.uint16)[3] = 0x7f81          # a signaling NaN in bf16
    assert O.minmax(yb, O.BF16) == (-5.0, 7.0)
    assert O.quantize(yb, O.BF16, O.UINT4, 1.0, 3)[1] & 0xf0 == 0   # element 3 = high nibble of byte 1
