"""Loads the MI355X-native ``libpiquant.so`` that sits next to this package.

Mirrors the role of the reference's ``python/src/piquant/_bootstrap.py`` (which dlopens the library
through cffi in ABI mode, lines 85-101).  cffi is optional here: the exported symbols are the same six C
functions, so either binding works; ctypes is used because it is always available.

There is deliberately no fallback: if the HIP library is missing the import fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from pathlib import Path

_LIB_NAME = 'libpiquant.so'

# C declarations (include/piquant.h + include/piquant_hip.h), as (name, restype, argtypes)
_vp, _sz, _f32, _i64, _int = C.c_void_p, C.c_size_t, C.c_float, C.c_int64, C.c_int
_DECLS = [
    # --- piquant.h: the reference ABI (reference include/piquant.h:42-85) -----------------------------
    ('piquant_context_create', _vp, [_sz]),
    ('piquant_context_destroy', None, [_vp]),
    ('piquant_quantize', None, [_vp, _vp, _int, _vp, _int, _sz, _f32, _i64, _int]),
    ('piquant_dequantize', None, [_vp, _vp, _int, _vp, _int, _sz, _f32, _i64, _int]),
    ('piquant_compute_quant_params_float32', None, [_vp, _vp, _sz, _int, C.POINTER(_f32), C.POINTER(_i64)]),
    ('piquant_compute_quant_params_bfloat16', None, [_vp, _vp, _sz, _int, C.POINTER(_f32), C.POINTER(_i64)]),
    # --- piquant_hip.h: additive extensions ---------------------------------------------------------
    ('piquant_hip_set_stream', None, [_vp, _vp]),
    ('piquant_hip_reset_stream', None, [_vp]),
    ('piquant_hip_set_blocking', None, [_vp, _int]),
    ('piquant_hip_set_blocking_wait', None, [_vp, _int]),
    ('piquant_hip_assume_device_pointers', None, [_vp, _int]),
    ('piquant_hip_set_stochastic_threshold', None, [_vp, _f32]),
    ('piquant_hip_set_stochastic_seed', None, [_vp, C.c_uint64]),
    ('piquant_hip_set_stochastic_per_element', None, [_vp, _int, C.c_uint64, C.c_uint64]),
    ('piquant_hip_set_reference_layout', None, [_vp, _int]),
    ('piquant_hip_set_reference_threads', None, [_vp, _int]),
    ('piquant_hip_quantize_uniform', None, [_vp, _vp, _int, _vp, _int, _sz, _f32, _i64, _int]),
    ('piquant_hip_dequantize_uniform', None, [_vp, _vp, _int, _vp, _int, _sz, _f32, _i64, _int]),
    ('piquant_hip_quantize_dequantize', None, [_vp, _vp, _int, _vp, _int, _sz, _f32, _i64, _int, _int]),
    ('piquant_hip_compute_quant_params_device', None, [_vp, _vp, _int, _sz, _int, _vp]),
    ('piquant_hip_quantize_dp', None, [_vp, _vp, _int, _vp, _int, _sz, _vp, _int]),
    ('piquant_hip_quantize_dynamic', None, [_vp, _vp, _int, _vp, _int, _sz, _vp, _int]),
    ('piquant_hip_set_fusion', None, [_vp, _int]),
    ('piquant_hip_set_independent_calls', None, [_vp, _int]),
    ('piquant_hip_set_host_path', None, [_vp, _int]),
    ('piquant_hip_host_path_in_effect', _int, [_vp]),
    ('piquant_hip_set_barrier_timeout_us', None, [_vp, C.c_uint32]),
    ('piquant_hip_barrier_bailouts', C.c_uint64, [_vp]),
    ('piquant_hip_quantize_dynamic_batch', None, [_vp, C.POINTER(C.c_void_p), _int, C.POINTER(C.c_void_p), _int, C.POINTER(_sz), C.POINTER(C.c_void_p), _sz, _int]),
    ('piquant_hip_dequantize_dp_batch', None, [_vp, C.POINTER(C.c_void_p), _int, C.POINTER(C.c_void_p), _int, C.POINTER(_sz), C.POINTER(C.c_void_p), _sz, _int]),
    ('piquant_hip_reduce_quantize_dynamic', None, [_vp, _vp, _int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _sz, _vp, _int, _sz, _vp, _int]),
    ('piquant_hip_dequantize_sum', None, [_vp, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _sz, _int, _vp, _int, _sz, _int]),
    ('piquant_hip_dequantize_dp', None, [_vp, _vp, _int, _vp, _int, _sz, _vp, _int]),
    ('piquant_hip_compute_quant_params_dist', None, [_vp, _vp, _int, _sz, _int, _vp, C.POINTER(_f32), C.POINTER(_i64)]),
    ('piquant_hip_minmax_keys', None, [_vp, _vp, _int, _sz, _vp, _int]),
    ('piquant_hip_peer_alloc', _vp, [_vp, _sz, _int, C.c_uint32, _vp]),
    ('piquant_hip_peer_open', _vp, [_vp, _vp]),
    ('piquant_hip_peer_close', None, [_vp, _vp]),
    ('piquant_hip_peer_free', None, [_vp, _vp]),
    ('piquant_hip_signal_flags', None, [_vp, C.POINTER(C.c_void_p), _sz, C.c_uint32]),
    ('piquant_hip_wait_flags', None, [_vp, _vp, _sz, C.c_uint32, C.c_uint32]),
    ('piquant_hip_exchange_minmax_keys', None, [_vp, _vp, C.POINTER(C.c_void_p), _vp, _sz, _vp, C.c_uint32]),
    ('piquant_hip_peer_timeout', _int, [_vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ('piquant_hip_decode_minmax_keys', None, [C.POINTER(C.c_int32), C.POINTER(_f32), C.POINTER(_f32)]),
    ('piquant_hip_quant_params_from_minmax', None, [_f32, _f32, _int, C.POINTER(_f32), C.POINTER(_i64)]),
    ('piquant_hip_device', _int, [_vp]),
    ('piquant_hip_version', C.c_char_p, []),
]

# enum values of include/piquant.h
PIQUANT_NEAREST, PIQUANT_STOCHASTIC = 0, 1
PIQUANT_REDUCE_OP_SET, PIQUANT_REDUCE_OP_ADD = 0, 1
PIQUANT_DTYPE_F32, PIQUANT_DTYPE_BF16, PIQUANT_DTYPE_UINT2, PIQUANT_DTYPE_UINT4, PIQUANT_DTYPE_UINT8 = range(5)


def library_path() -> Path:
    """libpiquant.so next to this package (where the reference's loader looks too).  PIQUANT_HIP_LIBRARY names another build of the same library:
    interleaved A/B runs of two builds (tools/, profiles/*_ab.*), nothing else."""
    override = os.environ.get('PIQUANT_HIP_LIBRARY')
    return Path(override) if override else Path(__file__).resolve().parent / _LIB_NAME


def _load_native_module() -> C.CDLL:
    # One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64 (torch/lib, found through an
    # $ORIGIN rpath under the unversioned name), libpiquant.so needs `libamdhip64.so.7`.  If PyTorch is loaded first the
    # dynamic loader satisfies our dependency with PyTorch's already-loaded runtime (same SONAME); in the other order
    # the process would end up with two runtimes that cannot see each other's devices and allocations.  So: torch first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    assert sys.platform.startswith('linux'), f'libpiquant.so for MI355X / ROCm exists for Linux only, this is {sys.platform}'
    lib_path = library_path()
    if not lib_path.exists():
        raise ImportError(
            f'piquant HIP library not found: {lib_path}. Build it with `make -C pi-quant_amd/csrc` '
            f'(or `python -c "import __graft_entry__ as g; g.build()"`). There is no CPU fallback.'
        )
    lib = C.CDLL(str(lib_path))
    for name, restype, argtypes in _DECLS:
        fn = getattr(lib, name)   # AttributeError here == the library does not export the ABI
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


C_LIB = _load_native_module()
