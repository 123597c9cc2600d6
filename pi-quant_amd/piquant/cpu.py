"""ctypes binding of libpiquant_cpu.so (include/piquant_cpu.h): the host-memory companion of the MI355X library.

The same arithmetic as the HIP kernels on AVX-512 host cores, for buffers that live in host memory -- where `piquant.Context` sends calls on
pageable host buffers by default (`Context.set_host_path`: 'auto'); device tensors always run the HIP kernels, and a missing HIP extension is
never papered over by this library.  bench.py times it as the reproducible CPU baseline.
"""
import ctypes as C
from pathlib import Path
from typing import Optional, Sequence, Tuple

_LIB_PATH = Path(__file__).resolve().parent / "libpiquant_cpu.so"
_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            raise FileNotFoundError(f"{_LIB_PATH} is missing: run `make -C pi-quant_amd/csrc/cpu` (or __graft_entry__.build())")
        L = C.CDLL(str(_LIB_PATH))
        L.piquant_cpu_context_create.restype = C.c_void_p
        L.piquant_cpu_context_create.argtypes = [C.c_size_t]
        L.piquant_cpu_context_destroy.argtypes = [C.c_void_p]
        L.piquant_cpu_num_threads.restype = C.c_size_t
        L.piquant_cpu_num_threads.argtypes = [C.c_void_p]
        L.piquant_cpu_set_active_threads.argtypes = [C.c_void_p, C.c_size_t]
        L.piquant_cpu_set_affinity.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_size_t]
        L.piquant_cpu_has_avx512.restype = C.c_int
        L.piquant_cpu_use_avx512.restype = C.c_int
        L.piquant_cpu_use_avx512.argtypes = [C.c_int]
        L.piquant_cpu_quantize.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_size_t, C.c_float, C.c_int64, C.c_int, C.c_float]
        L.piquant_cpu_dequantize.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_size_t, C.c_float, C.c_int64, C.c_int]
        L.piquant_cpu_quantize_reference_layout.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_size_t, C.c_float, C.c_int64, C.c_int, C.c_float,
                                                            C.c_size_t]
        L.piquant_cpu_dequantize_reference_layout.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_size_t, C.c_float, C.c_int64, C.c_int, C.c_size_t]
        L.piquant_cpu_minmax.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.piquant_cpu_compute_quant_params.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int64)]
        L.piquant_cpu_partition_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t]
        _lib = L
    return _lib


class CpuContext:
    """A pool of `num_threads` workers (0 = one per usable physical core); calls take raw host addresses, dtype / mode codes are those of piquant.h.
    quantize / dequantize hand every worker's share out in 256 KiB chunks (own share first, then the others'): a busy core costs a chunk, not the call."""

    def __init__(self, num_threads: int = 0):
        self._lib = lib()
        self._ctx = self._lib.piquant_cpu_context_create(num_threads)

    def close(self) -> None:
        if self._ctx:
            self._lib.piquant_cpu_context_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def num_threads(self) -> int:
        return int(self._lib.piquant_cpu_num_threads(self._ctx))

    def set_active_threads(self, n: int) -> None:
        self._lib.piquant_cpu_set_active_threads(self._ctx, n)

    def set_affinity(self, cpus: Sequence[int]) -> None:
        arr = (C.c_int * len(cpus))(*cpus)
        self._lib.piquant_cpu_set_affinity(self._ctx, arr, len(cpus))

    def quantize_ptr(self, ptr_in: int, dt_in: int, ptr_out: int, dt_out: int, numel: int, scale: float, zero_point: int, round_mode: int = 0,
                     threshold: float = 0.0, reference_threads: int = 0) -> None:
        """reference_threads > 0: reference-layout mode -- the bytes of the reference's AVX-512 context of that many pool threads (include/piquant_cpu.h)"""
        if reference_threads > 0:
            self._lib.piquant_cpu_quantize_reference_layout(self._ctx, ptr_in, dt_in, ptr_out, dt_out, numel, scale, zero_point, round_mode, threshold, reference_threads)
        else:
            self._lib.piquant_cpu_quantize(self._ctx, ptr_in, dt_in, ptr_out, dt_out, numel, scale, zero_point, round_mode, threshold)

    def dequantize_ptr(self, ptr_in: int, dt_in: int, ptr_out: int, dt_out: int, numel: int, scale: float, zero_point: int, reduce_op: int = 0,
                       reference_threads: int = 0) -> None:
        if reference_threads > 0:
            self._lib.piquant_cpu_dequantize_reference_layout(self._ctx, ptr_in, dt_in, ptr_out, dt_out, numel, scale, zero_point, reduce_op, reference_threads)
        else:
            self._lib.piquant_cpu_dequantize(self._ctx, ptr_in, dt_in, ptr_out, dt_out, numel, scale, zero_point, reduce_op)

    def minmax_ptr(self, ptr: int, dt: int, numel: int) -> Tuple[float, float]:
        lo, hi = C.c_float(), C.c_float()
        self._lib.piquant_cpu_minmax(self._ctx, ptr, dt, numel, C.byref(lo), C.byref(hi))
        return lo.value, hi.value

    def compute_quant_params_ptr(self, ptr: int, dt: int, numel: int, target: int) -> Tuple[float, int]:
        s, z = C.c_float(), C.c_int64()
        self._lib.piquant_cpu_compute_quant_params(self._ctx, ptr, dt, numel, target, C.byref(s), C.byref(z))
        return s.value, z.value

    def partition_copy_ptr(self, src: int, dst: int, dt: int, numel: int) -> None:
        self._lib.piquant_cpu_partition_copy(self._ctx, src, dst, dt, numel)


def has_avx512() -> bool:
    return bool(lib().piquant_cpu_has_avx512())


def use_avx512(enable: bool) -> bool:
    return bool(lib().piquant_cpu_use_avx512(1 if enable else 0))
