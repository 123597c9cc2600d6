"""The (min,max) -> (scale, zero_point) epilogue against an exact-rational model of reference src/piquant.cpp:245-258:
more than 10^6 random and adversarial pairs; the oracle's C restatement and the product's host function (libpiquant.so, pure host code)
must both equal the model bit for bit.  The device epilogue is checked against the same pairs in tests/test_gpu_epilogue.py."""
import ctypes
import multiprocessing as mp
import os

import numpy as np

from epilogue_cases import exact_epilogue, pairs

BITS = {8: 4, 4: 3, 2: 2}   # quantized width -> dtype code (include/piquant.h)


def _model_chunk(args):
    lo, hi, bits = args
    scales = np.empty(lo.size, dtype=np.float32)
    zps = np.empty(lo.size, dtype=np.int64)
    for i in range(lo.size):
        s, z = exact_epilogue(float(lo[i]), float(hi[i]), bits)
        scales[i], zps[i] = s, z
    return scales, zps


def _model(lo, hi, bits, pool):
    n = max(1, lo.size // (8 * (pool._processes if pool else 1)) + 1)   # noqa: SLF001
    chunks = [(lo[i: i + n], hi[i: i + n], bits) for i in range(0, lo.size, n)]
    res = pool.map(_model_chunk, chunks) if pool else [_model_chunk(c) for c in chunks]
    return np.concatenate([r[0] for r in res]), np.concatenate([r[1] for r in res])


def _c_epilogue(fn, lo, hi, code, first_arg_double):
    """per-pair ctypes calls of a C epilogue (the functions take scalars)"""
    s, z = ctypes.c_float(), ctypes.c_int64()
    scales = np.empty(lo.size, dtype=np.float32)
    zps = np.empty(lo.size, dtype=np.int64)
    conv = (ctypes.c_double if first_arg_double else ctypes.c_float)
    for i in range(lo.size):
        fn(conv(float(lo[i])), conv(float(hi[i])), code, ctypes.byref(s), ctypes.byref(z))
        scales[i], zps[i] = s.value, z.value
    return scales, zps


def test_model_reproduces_the_known_answers_of_the_reference():
    # SURVEY.md section 8 a11 [probe]: values the real reference printed
    known = [((-1.0, 3.0), 8, (0.015686275, 64)), ((-1.0, 3.0), 4, (0.26666668, 4)), ((-1.0, 3.0), 2, (1.3333334, 1)), ((42.0, 42.0), 8, (1.0, 127)),
             ((42.0, 42.0), 4, (1.0, 7)), ((42.0, 42.0), 2, (1.0, 1)), ((0.0, 1.0), 8, (0.0039215689, 0)), ((-1.0, 1.0), 8, (0.00784313772, 128)),
             ((-0.5, 1.5), 4, (0.13333334, 4)), ((2.0, 6.0), 8, (0.0156862754, 0)), ((-6.0, -2.0), 8, (0.0156862754, 255))]
    for (lo, hi), bits, (scale, zp) in known:
        s, z = exact_epilogue(lo, hi, bits)
        assert (np.float32(s), z) == (np.float32(scale), zp), ((lo, hi), bits, (s, z))


def test_oracle_and_product_host_epilogue_equal_the_exact_model(oracle_mod):
    import oracle.oracle as OM
    from piquant._bootstrap import C_LIB

    lo, hi = pairs(seed=2026, n_random=1_100_000)
    assert lo.size >= 1_000_000
    orc = OM._load().orc_quant_params_from_minmax   # noqa: SLF001
    orc.restype = None
    orc.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64)]
    host = C_LIB.piquant_hip_quant_params_from_minmax
    procs = min(32, os.cpu_count() or 1)
    with mp.get_context("fork").Pool(procs) as pool:
        for bits, code in BITS.items():
            # all widths on the adversarial families and a slice of the random ones; the 8-bit width on everything
            sel = slice(None) if bits == 8 else np.r_[0:lo.size:7, lo.size - 60000:lo.size]
            l, h = lo[sel], hi[sel]
            ms, mz = _model(l, h, bits, pool)
            os_, oz = _c_epilogue(orc, l, h, code, True)
            hs, hz = _c_epilogue(host, l, h, code, False)
            for name, s, z in (("oracle", os_, oz), ("product host", hs, hz)):
                bad = np.flatnonzero((s.view(np.uint32) != ms.view(np.uint32)) | (z != mz))
                assert bad.size == 0, (name, bits, bad.size, [(float(l[i]), float(h[i]), float(s[i]), int(z[i]), float(ms[i]), int(mz[i])) for i in bad[:5]])
