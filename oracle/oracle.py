"""TEST INFRASTRUCTURE ONLY -- ctypes/numpy front end of the C oracle and of oracle/_ref.

Arrays: f32 tensors are numpy float32, bf16 tensors are numpy uint16 bit patterns, quantized tensors
are numpy uint8 holding `packed_numel` bytes (low bits = lower element index, reference
src/kernels/quantize.inl:36-50).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / "liboracle.so"
REF_PATH = HERE / "_ref" / "libpiquant_ref.so"

# include/piquant.h:33-40 (reference) -- same codes as the product ABI
F32, BF16, UINT2, UINT4, UINT8 = 0, 1, 2, 3, 4
NEAREST, STOCHASTIC = 0, 1
SET, ADD = 0, 1
FORM_REFERENCE, FORM_UNIFORM = 0, 1

_BITS = {F32: 32, BF16: 16, UINT2: 2, UINT4: 4, UINT8: 8}
_NP_OF = {F32: np.float32, BF16: np.uint16}


def build(with_ref: bool | None = None) -> None:
    """Compile liboracle.so (always) and oracle/_ref (only where /root/reference is mounted)."""
    subprocess.run(["make", "-s", "-C", str(HERE)], check=True)
    if with_ref is None:
        with_ref = Path("/root/reference/src/kernels/kernels.inl").exists()
    if with_ref:
        subprocess.run(["make", "-s", "-C", str(HERE), "ref"], check=True)


_lib = None


def _load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            build(with_ref=False)
        lib = C.CDLL(str(LIB_PATH))
        i64, f32, vp, ci = C.c_int64, C.c_float, C.c_void_p, C.c_int
        lib.orc_packed_numel.restype = i64
        lib.orc_packed_numel.argtypes = [i64, ci]
        lib.orc_quantize.argtypes = [vp, ci, vp, ci, i64, f32, i64, ci, f32, ci]
        lib.orc_dequantize.argtypes = [vp, ci, vp, ci, i64, f32, i64, ci, ci]
        lib.orc_quantize_threads.argtypes = [vp, ci, vp, ci, i64, f32, i64, ci, f32, ci, ci]
        lib.orc_dequantize_threads.argtypes = [vp, ci, vp, ci, i64, f32, i64, ci, ci, ci]
        lib.orc_requantize.argtypes = [vp, ci, vp, ci, i64, f32, i64, ci, f32, ci]
        lib.orc_partition.restype = ci
        lib.orc_partition.argtypes = [i64, i64, i64, ci, C.POINTER(i64), C.POINTER(i64)]
        lib.orc_minmax_f32.argtypes = [vp, i64, C.POINTER(f32)]
        lib.orc_minmax_bf16.argtypes = [vp, i64, C.POINTER(f32)]
        lib.orc_quant_params_from_minmax.argtypes = [C.c_double, C.c_double, ci, C.POINTER(f32), C.POINTER(i64)]
        lib.orc_element_threshold.restype = f32
        lib.orc_element_threshold.argtypes = [C.c_uint64, C.c_uint64]
        lib.orc_quantize_per_element.argtypes = [vp, ci, vp, ci, i64, f32, i64, C.c_uint64, C.c_uint64]
        lib.orc_f32_to_bf16.restype = C.c_uint16
        lib.orc_f32_to_bf16.argtypes = [f32]
        _lib = lib
    return _lib


def packed_numel(numel: int, dtype: int) -> int:
    per = 8 // _BITS[dtype]
    return (numel + per - 1) // per


def _check_in(x: np.ndarray, dt: int) -> np.ndarray:
    want = _NP_OF[dt]
    assert x.dtype == want, f"dtype code {dt} needs numpy {want}, got {x.dtype}"
    return np.ascontiguousarray(x).reshape(-1)


def f32_to_bf16(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even with NaN quieting (reference include/piquant.hpp:86-90)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    nan = (u & 0x7FFFFFFF) > 0x7F800000
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF
    r = np.where(nan, (u >> 16) | 64, r)
    return r.astype(np.uint16)


def bf16_to_f32(b: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def quantize(x, dt_in, dt_out, scale, zero_point, round_mode=NEAREST, rnd_threshold=0.0,
             form=FORM_UNIFORM, threads=1, out=None) -> np.ndarray:
    lib = _load()
    x = _check_in(x, dt_in)
    n = x.size
    if out is None:
        out = np.zeros(packed_numel(n, dt_out), dtype=np.uint8)
    assert out.dtype == np.uint8 and out.size == packed_numel(n, dt_out)
    if threads == 1:
        lib.orc_quantize(x.ctypes.data, dt_in, out.ctypes.data, dt_out, n, scale, int(zero_point), round_mode,
                         rnd_threshold, form)
    else:
        lib.orc_quantize_threads(x.ctypes.data, dt_in, out.ctypes.data, dt_out, n, scale, int(zero_point),
                                 round_mode, rnd_threshold, form, threads)
    return out


def dequantize(q, dt_in, dt_out, numel, scale, zero_point, reduce_op=SET, form=FORM_UNIFORM, threads=1,
               out=None) -> np.ndarray:
    """`out` (required for ADD) is accumulated into / overwritten in place and returned."""
    lib = _load()
    q = np.ascontiguousarray(q, dtype=np.uint8).reshape(-1)
    assert q.size == packed_numel(numel, dt_in), (q.size, numel, dt_in)
    if out is None:
        assert reduce_op == SET, "ADD needs an accumulator"
        out = np.zeros(numel, dtype=_NP_OF[dt_out])
    assert out.dtype == _NP_OF[dt_out] and out.size == numel and out.flags.c_contiguous
    if threads == 1:
        lib.orc_dequantize(q.ctypes.data, dt_in, out.ctypes.data, dt_out, numel, scale, int(zero_point), reduce_op, form)
    else:
        lib.orc_dequantize_threads(q.ctypes.data, dt_in, out.ctypes.data, dt_out, numel, scale, int(zero_point),
                                   reduce_op, form, threads)
    return out


def requantize(x, dt_inout, quant_dtype, scale, zero_point, round_mode=NEAREST, rnd_threshold=0.0, reduce_op=SET, out=None) -> np.ndarray:
    """Fused quantize->dequantize; `out` (same float dtype) is required for ADD and returned."""
    lib = _load()
    x = _check_in(x, dt_inout)
    if out is None:
        assert reduce_op == SET, "ADD needs an accumulator"
        out = np.zeros(x.size, dtype=_NP_OF[dt_inout])
    assert out.dtype == _NP_OF[dt_inout] and out.size == x.size and out.flags.c_contiguous
    lib.orc_requantize(x.ctypes.data, dt_inout, out.ctypes.data, quant_dtype, x.size, scale, int(zero_point), round_mode,
                       rnd_threshold, reduce_op)
    return out


def partition(numel: int, ti: int, tc: int, packed_bits: int):
    b, n = C.c_int64(), C.c_int64()
    ok = _load().orc_partition(numel, ti, tc, packed_bits, C.byref(b), C.byref(n))
    return (b.value, n.value) if ok else None


def minmax(x: np.ndarray, dt: int):
    lib = _load()
    x = _check_in(x, dt)
    mm = (C.c_float * 2)()
    (lib.orc_minmax_f32 if dt == F32 else lib.orc_minmax_bf16)(x.ctypes.data, x.size, mm)
    return float(mm[0]), float(mm[1])


def quant_params_from_minmax(r_min: float, r_max: float, quant_dtype: int):
    s, z = C.c_float(), C.c_int64()
    _load().orc_quant_params_from_minmax(r_min, r_max, quant_dtype, C.byref(s), C.byref(z))
    return float(s.value), int(z.value)


def compute_quant_params(x: np.ndarray, dt: int, quant_dtype: int):
    lo, hi = minmax(x, dt)
    return quant_params_from_minmax(lo, hi, quant_dtype)


def element_threshold(seed: int, index: int) -> float:
    return float(_load().orc_element_threshold(seed, index))


def quantize_per_element(x, dt_in, dt_out, scale, zero_point, seed, index_base=0) -> np.ndarray:
    lib = _load()
    x = _check_in(x, dt_in)
    out = np.zeros(packed_numel(x.size, dt_out), dtype=np.uint8)
    lib.orc_quantize_per_element(x.ctypes.data, dt_in, out.ctypes.data, dt_out, x.size, scale, int(zero_point),
                                 seed, index_base)
    return out


# ---------------------------------------------------------------------------------------------------
# oracle/_ref : the reference's own kernel units (built only in the container that has /root/reference;
# the prebuilt .so travels to the GPU box inside the repo snapshot).
# ---------------------------------------------------------------------------------------------------
def ref_available() -> bool:
    return REF_PATH.exists()


class Ref:
    """Calls into the reference kernels through oracle/ref_driver.cpp."""

    GENERIC, SSE42, AVX2, AVX512F, AVX512F_BF16 = range(5)

    def __init__(self) -> None:
        if not REF_PATH.exists():
            raise FileNotFoundError(f"{REF_PATH} not built (make -C oracle ref, needs /root/reference)")
        lib = C.CDLL(str(REF_PATH))
        i64, f32, vp, ci = C.c_longlong, C.c_float, C.c_void_p, C.c_int
        lib.ref_isa_name.restype = C.c_char_p
        lib.ref_quantize.argtypes = [ci, vp, ci, vp, ci, i64, f32, i64, ci, f32, ci]
        lib.ref_dequantize.argtypes = [ci, vp, ci, vp, ci, i64, f32, i64, ci, ci]
        lib.ref_requantize.argtypes = [ci, vp, ci, vp, ci, i64, f32, i64, ci, f32, ci]
        lib.ref_minmax_f32.argtypes = [ci, vp, i64, ci, C.POINTER(f32)]
        lib.ref_minmax_bf16.argtypes = [ci, vp, i64, C.POINTER(f32)]
        lib.ref_set_pinning.argtypes = [C.POINTER(ci), ci]
        lib.ref_partition_copy.argtypes = [vp, vp, i64, ci, ci]
        self.lib = lib

    def set_pinning(self, cpus) -> None:
        """pool thread t runs on logical CPU cpus[t] (timing runs; an empty list removes the pinning)"""
        arr = (C.c_int * max(len(cpus), 1))(*cpus)
        self.lib.ref_set_pinning(arr, len(cpus))

    def partition_copy(self, src: np.ndarray, dst: np.ndarray, threads: int) -> np.ndarray:
        """dst[:] = src, each pool thread copying (and thereby first-touching) the partition it will later process"""
        assert src.flags.c_contiguous and dst.flags.c_contiguous and src.dtype == dst.dtype and src.size == dst.size
        self.lib.ref_partition_copy(src.ctypes.data, dst.ctypes.data, src.size, src.dtype.itemsize, threads)
        return dst

    def isa_name(self, isa: int) -> str:
        return self.lib.ref_isa_name(isa).decode()

    def supported(self, isa: int) -> bool:
        return bool(self.lib.ref_isa_supported(isa))

    def best_isa(self) -> int:
        return int(self.lib.ref_isa_best())

    def quantize(self, x, dt_in, dt_out, scale, zero_point, round_mode=NEAREST, rnd_threshold=0.0, isa=None,
                 threads=1, out=None) -> np.ndarray:
        isa = self.best_isa() if isa is None else isa
        x = _check_in(x, dt_in)
        if out is None:
            out = np.zeros(packed_numel(x.size, dt_out), dtype=np.uint8)
        self.lib.ref_quantize(isa, x.ctypes.data, dt_in, out.ctypes.data, dt_out, x.size, scale, int(zero_point),
                              round_mode, rnd_threshold, threads)
        return out

    def dequantize(self, q, dt_in, dt_out, numel, scale, zero_point, reduce_op=SET, isa=None, threads=1,
                   out=None) -> np.ndarray:
        isa = self.best_isa() if isa is None else isa
        q = np.ascontiguousarray(q, dtype=np.uint8).reshape(-1)
        # the generic uint2 kernel reads one byte past the end when numel % 4 == 0
        # (reference src/kernels/dequantize.inl:72) -- give it a padded copy.
        qpad = np.concatenate([q, np.zeros(8, dtype=np.uint8)])
        if out is None:
            assert reduce_op == SET
            out = np.zeros(numel, dtype=_NP_OF[dt_out])
        self.lib.ref_dequantize(isa, qpad.ctypes.data, dt_in, out.ctypes.data, dt_out, numel, scale, int(zero_point),
                                reduce_op, threads)
        return out

    def requantize(self, x, dt_inout, quant_dtype, scale, zero_point, round_mode=NEAREST, rnd_threshold=0.0, reduce_op=SET,
                   isa=None, out=None) -> np.ndarray:
        isa = self.best_isa() if isa is None else isa
        x = _check_in(x, dt_inout)
        if out is None:
            assert reduce_op == SET
            out = np.zeros(x.size, dtype=_NP_OF[dt_inout])
        self.lib.ref_requantize(isa, x.ctypes.data, dt_inout, out.ctypes.data, quant_dtype, x.size, scale, int(zero_point),
                                round_mode, rnd_threshold, reduce_op)
        return out

    def minmax(self, x, dt, isa=None, threads=1):
        isa = self.best_isa() if isa is None else isa
        x = _check_in(x, dt)
        mm = (C.c_float * 2)()
        if dt == F32:
            self.lib.ref_minmax_f32(isa, x.ctypes.data, x.size, threads, mm)
        else:
            self.lib.ref_minmax_bf16(isa, x.ctypes.data, x.size, mm)
        return float(mm[0]), float(mm[1])


if os.environ.get("PIQUANT_ORACLE_BUILD"):
    build()
