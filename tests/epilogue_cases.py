"""(min, max) pairs for pinning the (min,max) -> (scale, zero_point) epilogue (reference src/piquant.cpp:245-258) and its exact model.

``exact_epilogue`` evaluates the reference's lines step by step in exact rational arithmetic, rounding to IEEE double exactly where the
C++ does (each `double` operation is the correctly rounded value of the exact result; ``float(Fraction)`` is correctly rounded):

    double scale      = (r_max - r_min) / (q_max - q_min);                                    // :253
    double zero_point = q_min - r_min / scale;                                               // :254
    zero_point = max(min((double)(int64)round(zero_point), q_max), q_min);                   // :255  (round = half away from zero)
    return {(float)scale, (int64)zero_point};                                                // :256  (double -> float: nearest even)

This is still not reference execution (src/piquant.cpp cannot be built here: it includes an un-vendored thread-pool header); it removes
"three restatements share one author's reading of IEEE arithmetic" as a failure mode: the model uses no floating-point operation at all.
"""
from fractions import Fraction

import numpy as np


def exact_epilogue(r_min: float, r_max: float, bits: int):
    qmax = (1 << bits) - 1
    if r_max == r_min:                                      # :249-252
        return 1.0, qmax >> 1
    d = float(Fraction(r_max) - Fraction(r_min))            # both operands are exact doubles (fp32 widened)
    s = float(Fraction(d) / qmax)                           # q_max - q_min == qmax exactly
    t = float(Fraction(r_min) / Fraction(s))                # r_min / scale
    zp = -t                                                 # 0.0 - t: exact
    mag = Fraction(abs(zp))
    r = (mag.numerator * 2 + mag.denominator) // (2 * mag.denominator)    # floor(|zp| + 1/2): round half away from zero
    zi = r if zp >= 0 else -r
    assert abs(zi) < 2 ** 63                                # the int64 cast is always in range (|r_min| / scale <= 2^24 * qmax)
    zi = max(min(zi, qmax), 0)
    return float(np.float32(s)), zi


def pairs(seed: int, n_random: int):
    """float32 arrays (lo, hi), lo <= hi, finite: random bit patterns, log-uniform magnitudes, and the adversarial families."""
    rng = np.random.default_rng(seed)
    out = []
    # (1) any two finite bit patterns
    bits = rng.integers(0, 2 ** 32, size=(n_random // 2, 2), dtype=np.uint64).astype(np.uint32)
    f = bits.view(np.float32)
    f = f[np.isfinite(f).all(axis=1)]
    out.append(np.sort(f, axis=1))
    # (2) log-uniform magnitudes over 60 decades, mixed signs, including narrow ranges around a centre
    m = n_random // 4
    a = (10.0 ** rng.uniform(-30, 30, m) * rng.choice([-1.0, 1.0], m)).astype(np.float32)
    b = (10.0 ** rng.uniform(-30, 30, m) * rng.choice([-1.0, 1.0], m)).astype(np.float32)
    out.append(np.sort(np.stack([a, b], axis=1), axis=1))
    c = (10.0 ** rng.uniform(-20, 20, m) * rng.choice([-1.0, 1.0], m)).astype(np.float32)
    w = (np.abs(c) * 10.0 ** rng.uniform(-7, 1, m)).astype(np.float32)
    out.append(np.sort(np.stack([c - w, c + w], axis=1).astype(np.float32), axis=1))
    # (3) exact ties: scale = 2^e, -min/scale = k + 1/2 for every k and quantized width
    ties = []
    for qmax in (255, 15, 3):
        for e in (-40, -10, -1, 0, 3, 20):
            for k in range(0, qmax + 1):
                lo = -(k + 0.5) * 2.0 ** e
                ties.append((lo, lo + qmax * 2.0 ** e))
                ties.append((np.nextafter(np.float32(lo), np.float32(0)), lo + qmax * 2.0 ** e))
                ties.append((lo, np.nextafter(np.float32(lo + qmax * 2.0 ** e), np.float32(np.inf))))
    out.append(np.array(ties, dtype=np.float32))
    # (4) neighbours, denormals, extremes, one-sided ranges (the zero point clamps), degenerate ranges
    fmax, tiny, den = np.float32(3.4028235e38), np.float32(1.1754944e-38), np.float32(1e-45)
    base = rng.integers(0, 2 ** 32, size=20000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    base = base[np.isfinite(base) & (np.abs(base) < fmax)]
    out.append(np.sort(np.stack([base, np.nextafter(base, np.float32(np.inf))], axis=1), axis=1))
    special = [(-fmax, fmax), (-fmax, -fmax), (fmax, fmax), (-fmax, 0), (0, fmax), (0, 0), (-0.0, 0.0), (den, 2 * den), (-den, den), (0, den),
               (-tiny, tiny), (tiny, np.nextafter(tiny, np.float32(1))), (-1, 1), (2, 6), (-6, -2), (-1, 3), (0, 1), (-0.5, 1.5), (42, 42),
               (1e30, 1.0000001e30), (-1e-30, 1e30), (-1e30, 1e-30), (np.nextafter(-fmax, np.float32(0)), fmax)]
    out.append(np.array(special, dtype=np.float32))
    p = np.concatenate(out).astype(np.float32)
    p = p[p[:, 0] <= p[:, 1]]
    return np.ascontiguousarray(p[:, 0]), np.ascontiguousarray(p[:, 1])
