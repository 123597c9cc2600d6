"""piquant -- MI355X-native drop-in for pi-quant's Python package.

Same public names as the reference package (reference ``python/src/piquant/__init__.py:20-142``):
``RoundMode``, ``ReduceOp``, ``DataType``, ``Context`` with ``quantize_ptr`` / ``dequantize_ptr`` /
``compute_quant_params_ptr_float32`` / ``compute_quant_params_ptr_bfloat16``, and the ``piquant.torch``
module.  Behind it every call on device, pinned or managed memory lands in hand-written HIP kernels through the C ABI of ``libpiquant.so``.
Pointers may be device pointers (PyTorch-ROCm ``tensor.data_ptr()``) or pageable host pointers (served by the companion library
``libpiquant_cpu.so`` where they live -- the default -- or staged over PCIe to the same HIP kernels: ``piquant_hip_set_host_path``).
"""
from __future__ import annotations

__version__ = '0.1.0'

import ctypes as _C
import importlib.util
import threading
import weakref
from enum import Enum, unique
from typing import Dict, Optional, Tuple, Union

from . import _bootstrap as _b
from ._bootstrap import C_LIB as C


@unique
class RoundMode(Enum):
    NEAREST = _b.PIQUANT_NEAREST
    STOCHASTIC = _b.PIQUANT_STOCHASTIC


@unique
class ReduceOp(Enum):
    SET = _b.PIQUANT_REDUCE_OP_SET
    ADD = _b.PIQUANT_REDUCE_OP_ADD


@unique
class DataType(Enum):
    F32 = _b.PIQUANT_DTYPE_F32
    BF16 = _b.PIQUANT_DTYPE_BF16
    UINT2 = _b.PIQUANT_DTYPE_UINT2
    UINT4 = _b.PIQUANT_DTYPE_UINT4
    UINT8 = _b.PIQUANT_DTYPE_UINT8

    @property
    def bit_size(self) -> int:
        return {DataType.F32: 32, DataType.BF16: 16, DataType.UINT2: 2, DataType.UINT4: 4, DataType.UINT8: 8}[self]

    @property
    def is_quantized(self) -> bool:
        return self in (DataType.UINT2, DataType.UINT4, DataType.UINT8)

    @property
    def is_dequantized(self) -> bool:
        return self in (DataType.F32, DataType.BF16)

    @property
    def stride(self) -> int:
        """Bytes per storage unit (packed types live in bytes)."""
        return max(8, self.bit_size) >> 3

    def packed_nbytes(self, numel: int) -> int:
        """Bytes that hold ``numel`` elements (reference src/piquant_internal.hpp:41-44)."""
        if self.bit_size >= 8:
            return numel * (self.bit_size >> 3)
        per = 8 // self.bit_size
        return (numel + per - 1) // per


class Context:
    """Owns one native context, bound to the HIP device that is current at construction.

    ``num_threads`` sized the reference's CPU thread pool (``__init__.py:65-70``; default there and here: CPUs - 1).  The GPU grid does
    the partitioning, but the number still means something: ``quantize_ptr`` / ``dequantize_ptr`` write, byte for byte, what a reference
    context of that many pool threads writes -- its partitions' scalar heads and tails included (``set_reference_layout``).
    """

    # Default contexts are per (thread, device): a context carries the stream and the blocking / pointer modes of the call being
    # made, so two threads driving different streams through one shared default context could swap each other's stream between
    # "set the stream" and "make the call".  Contexts are cheap (a private stream, ~60 KiB of device state, two pinned words).
    _tls = threading.local()
    # Policy applied to a DEFAULT context -- fusion, barrier timeout, blocking-wait mode, stochastic seed / threshold, reference layout,
    # host path -- is process-wide intent ("fusion off", "seeded run"), but default contexts are per thread: without this record the
    # default context an autograd or DDP comm-hook thread gets on first use would silently lack what the main thread configured.  Setters
    # called on any default context are recorded here and replayed into every default context created afterwards (explicit contexts are
    # nobody's template and inherit nothing).
    _default_policy: Dict[str, tuple] = {}
    _default_policy_lock = threading.Lock()

    def _record_policy(self, name: str, *args) -> None:
        if getattr(self, '_is_default', False):
            with Context._default_policy_lock:
                Context._default_policy[name] = args

    def __init__(self, num_threads: Union[int, None] = None) -> None:
        if num_threads is None:   # reference __init__.py:67-68
            import multiprocessing

            num_threads = max(multiprocessing.cpu_count() - 1, 1)
        self._num_threads = int(num_threads)
        _require_device()   # a Python exception instead of the native abort when there is no GPU
        self._ctx = C.piquant_context_create(self._num_threads)
        assert self._ctx, 'piquant_context_create returned NULL'
        self._finalizer = weakref.finalize(self, C.piquant_context_destroy, self._ctx)
        self._device = int(C.piquant_hip_device(self._ctx))
        # last values pushed to the native context: the setters below only cross the FFI when something changes
        self._stream: object = 'own'
        self._blocking = True
        self._assume_device = False
        self._native_managed = False     # piquant.torch's C++ front end drives this context too: the cache above cannot be trusted, always push

    @staticmethod
    def get(device_index: Optional[int] = None) -> 'Context':
        """Default context of the calling thread (one per thread and HIP device, created on first use)."""
        if device_index is None:
            device_index = _current_device()
        defaults = Context._thread_defaults()
        ctx = defaults.get(device_index)
        if ctx is None:
            ctx = _make_on_device(device_index)
            with Context._default_policy_lock:
                policy = dict(Context._default_policy)
            for name, args in policy.items():      # what earlier default contexts were told (see _default_policy)
                getattr(ctx, name)(*args)
            ctx._is_default = True
            defaults[device_index] = ctx
        return ctx

    @staticmethod
    def _thread_defaults() -> Dict[int, 'Context']:
        d = getattr(Context._tls, 'defaults', None)
        if d is None:
            d = Context._tls.defaults = {}
        return d

    @property
    def device(self) -> int:
        return self._device

    # ---- reference surface (python/src/piquant/__init__.py:82-142) ---------------------------------
    def quantize_ptr(self, ptr_in: int, dtype_in: DataType, ptr_out: int, dtype_out: DataType, numel: int, scale: float,
                     zero_point: int, round_mode: RoundMode, _device_ptrs: bool = False, uniform: bool = False) -> None:
        """``uniform=True`` (additive): the position-independent form -- the SIMD-body formula at every element, whatever the context's
        layout mode: what a shard of a larger tensor is computed with (``piquant_hip_quantize_uniform``)."""
        assert dtype_in.is_dequantized, f'Input dtype must be a dequantized type, but is: {dtype_in}'
        assert dtype_out.is_quantized, f'Output dtype must be a quantized type, but is: {dtype_out}'
        assert numel == 0 or ptr_in != 0, 'Input arr pointer must not be NULL'
        assert numel == 0 or ptr_out != 0, 'Output arr pointer must not be NULL'
        self.assume_device_pointers(_device_ptrs)
        call = C.piquant_hip_quantize_uniform if uniform else C.piquant_quantize
        call(self._ctx, ptr_in, dtype_in.value, ptr_out, dtype_out.value, numel, scale, zero_point, round_mode.value)

    def dequantize_ptr(self, ptr_in: int, dtype_in: DataType, ptr_out: int, dtype_out: DataType, numel: int, scale: float,
                       zero_point: int, reduce_op: ReduceOp, _device_ptrs: bool = False, uniform: bool = False) -> None:
        assert dtype_in.is_quantized, f'Input dtype must be a quantized type, but is: {dtype_in}'
        assert dtype_out.is_dequantized, f'Output dtype must be a dequantized type, but is: {dtype_out}'
        assert numel == 0 or ptr_in != 0, 'Input arr pointer must not be NULL'
        assert numel == 0 or ptr_out != 0, 'Output arr pointer must not be NULL'
        self.assume_device_pointers(_device_ptrs)
        call = C.piquant_hip_dequantize_uniform if uniform else C.piquant_dequantize
        call(self._ctx, ptr_in, dtype_in.value, ptr_out, dtype_out.value, numel, scale, zero_point, reduce_op.value)

    def compute_quant_params_ptr_float32(self, ptr: int, target_quant_dtype: DataType, numel: int, _device_ptrs: bool = False) -> Tuple[float, int]:
        assert target_quant_dtype.is_quantized, f'Target dtype must be a quantized type, but is: {target_quant_dtype}'
        assert ptr != 0, 'Input arr pointer must not be NULL'
        scale, zero_point = _C.c_float(), _C.c_int64()
        self.assume_device_pointers(_device_ptrs)
        C.piquant_compute_quant_params_float32(self._ctx, ptr, numel, target_quant_dtype.value, _C.byref(scale), _C.byref(zero_point))
        return scale.value, zero_point.value

    def compute_quant_params_ptr_bfloat16(self, ptr: int, target_quant_dtype: DataType, numel: int, _device_ptrs: bool = False) -> Tuple[float, int]:
        assert target_quant_dtype.is_quantized, f'Target dtype must be a quantized type, but is: {target_quant_dtype}'
        assert ptr != 0, 'Input arr pointer must not be NULL'
        scale, zero_point = _C.c_float(), _C.c_int64()
        self.assume_device_pointers(_device_ptrs)
        C.piquant_compute_quant_params_bfloat16(self._ctx, ptr, numel, target_quant_dtype.value, _C.byref(scale), _C.byref(zero_point))
        return scale.value, zero_point.value

    # ---- additive GPU controls (include/piquant_hip.h) ----------------------------------------------
    def set_stream(self, hip_stream: int) -> None:
        """Enqueue on this hipStream_t, e.g. ``torch.cuda.current_stream().cuda_stream``.  0 is HIP's legacy default
        stream (PyTorch's default stream), not "no stream"; see ``reset_stream``."""
        if self._stream != hip_stream or self._native_managed:
            C.piquant_hip_set_stream(self._ctx, hip_stream or None)
            self._stream = hip_stream

    def _native_call(self) -> int:
        """For the C++ front end (csrc/torch_binding.cpp), which pushes stream / non-blocking / device-pointer mode to the native
        context itself: returns the native handle and brings the cache of pushed settings in line (the stream as unknown, so a
        later ctypes call pushes it again)."""
        self._stream = None
        self._blocking = False
        self._assume_device = True
        return self._ctx

    def reset_stream(self) -> None:
        """Back to the context's private non-blocking stream (the state of a new context)."""
        if self._stream != 'own' or self._native_managed:
            C.piquant_hip_reset_stream(self._ctx)
            self._stream = 'own'

    def set_blocking(self, blocking: bool) -> None:
        """True (native default): calls return after completion, like the reference.  False: stream-ordered."""
        if self._blocking != bool(blocking) or self._native_managed:
            C.piquant_hip_set_blocking(self._ctx, 1 if blocking else 0)
            self._blocking = bool(blocking)

    def set_blocking_wait(self, mode: str) -> None:
        """How a blocking call waits for the GPU: 'sync' (hipStreamSynchronize), 'write32' (the command processor writes a pinned host
        word behind the kernel, the host spins on it), 'kernel' (a one-thread kernel writes it) or 'event' (the work kernel carries a stop
        event -- its own completion signal -- and the host polls hipEventQuery); include/piquant_hip.h."""
        self._record_policy('set_blocking_wait', mode)
        C.piquant_hip_set_blocking_wait(self._ctx, {'sync': 0, 'write32': 1, 'kernel': 2, 'event': 3}[mode])

    def assume_device_pointers(self, assume: bool) -> None:
        """Skip the native pointer classification for the calls that follow (all buffers are device or pinned memory).
        The ``*_ptr`` methods set it per call from their ``_device_ptrs`` argument (False unless the torch binding knows
        better), so raw-pointer users always get the classifying, host-pointer-safe behaviour."""
        if self._assume_device != bool(assume) or self._native_managed:
            C.piquant_hip_assume_device_pointers(self._ctx, 1 if assume else 0)
            self._assume_device = bool(assume)

    def set_stochastic_threshold(self, threshold: Optional[float]) -> None:
        """Pin the per-call stochastic threshold in [0,1) (None: draw a fresh one per call, the default)."""
        self._record_policy('set_stochastic_threshold', threshold)
        C.piquant_hip_set_stochastic_threshold(self._ctx, -1.0 if threshold is None else float(threshold))

    def set_stochastic_seed(self, seed: int) -> None:
        self._record_policy('set_stochastic_seed', seed)
        C.piquant_hip_set_stochastic_seed(self._ctx, seed & 0xFFFFFFFFFFFFFFFF)

    def set_stochastic_per_element(self, enabled: bool, seed: int = 0, index_base: int = 0) -> None:
        """Opt-in: an independent threshold per element (counter hash of seed and global element index)."""
        self._record_policy('set_stochastic_per_element', enabled, seed, index_base)
        C.piquant_hip_set_stochastic_per_element(self._ctx, 1 if enabled else 0, seed & 0xFFFFFFFFFFFFFFFF, index_base)

    def set_reference_layout(self, enabled: bool, threads: Optional[int] = None) -> None:
        """On (the default): ``quantize_ptr`` / ``dequantize_ptr`` reproduce the reference's scalar head / tail formulas at the positions where
        its AVX-512 build uses them (include/piquant_hip.h), for a reference context with ``threads`` pool threads (each partition has its
        own head and tail; None: the ``num_threads`` this context was created with).  Off: every element takes the SIMD-body formula."""
        threads = self._num_threads if threads is None else int(threads)
        self._record_policy('set_reference_layout', enabled, threads)
        C.piquant_hip_set_reference_threads(self._ctx, max(threads, 1))
        C.piquant_hip_set_reference_layout(self._ctx, 1 if enabled else 0)

    def quantize_dequantize_ptr(self, ptr_in: int, dtype_in_out: DataType, ptr_out: int, quant_dtype: DataType, numel: int, scale: float,
                                zero_point: int, round_mode: RoundMode, reduce_op: ReduceOp, _device_ptrs: bool = False) -> None:
        """Fused quantize->dequantize (the reference's C++-only ``quantize_dequantize_fused``, piquant.hpp:276-285)."""
        assert dtype_in_out.is_dequantized and quant_dtype.is_quantized
        self.assume_device_pointers(_device_ptrs)
        C.piquant_hip_quantize_dequantize(self._ctx, ptr_in, dtype_in_out.value, ptr_out, quant_dtype.value, numel, scale, zero_point,
                                          round_mode.value, reduce_op.value)

    # device-resident parameters: a 16-byte record {float scale, float 1/scale, int64 zero_point} in device memory
    def compute_quant_params_device_ptr(self, ptr: int, dtype: DataType, numel: int, target_quant_dtype: DataType, params_ptr: int, _device_ptrs: bool = False) -> None:
        assert dtype.is_dequantized and target_quant_dtype.is_quantized and params_ptr != 0
        self.assume_device_pointers(_device_ptrs)
        C.piquant_hip_compute_quant_params_device(self._ctx, ptr, dtype.value, numel, target_quant_dtype.value, params_ptr)

    def quantize_dp_ptr(self, ptr_in: int, dtype_in: DataType, ptr_out: int, dtype_out: DataType, numel: int, params_ptr: int,
                        round_mode: RoundMode, _device_ptrs: bool = False) -> None:
        assert dtype_in.is_dequantized and dtype_out.is_quantized and params_ptr != 0
        self.assume_device_pointers(_device_ptrs)
        C.piquant_hip_quantize_dp(self._ctx, ptr_in, dtype_in.value, ptr_out, dtype_out.value, numel, params_ptr, round_mode.value)

    def quantize_dynamic_ptr(self, ptr_in: int, dtype_in: DataType, ptr_out: int, dtype_out: DataType, numel: int, params_ptr: int,
                             round_mode: RoundMode, _device_ptrs: bool = False) -> None:
        """compute_quant_params + quantize as one stream-ordered call; one kernel launch reading the tensor once when it fits on
        the chip (include/piquant_hip.h, piquant_hip_quantize_dynamic).  The parameter record is written to ``params_ptr``."""
        assert dtype_in.is_dequantized and dtype_out.is_quantized and params_ptr != 0
        self.assume_device_pointers(_device_ptrs)
        C.piquant_hip_quantize_dynamic(self._ctx, ptr_in, dtype_in.value, ptr_out, dtype_out.value, numel, params_ptr, round_mode.value)

    def quantize_dynamic_batch_ptr(self, ptrs_in, dtype_in: DataType, ptrs_out, dtype_out: DataType, numels, params_ptrs, round_mode: RoundMode,
                                   _device_ptrs: bool = False) -> None:
        """``quantize_dynamic_ptr`` for several independent tensors (own parameters and record each); up to 16 of them share one
        kernel launch (include/piquant_hip.h, piquant_hip_quantize_dynamic_batch)."""
        n = len(ptrs_in)
        assert dtype_in.is_dequantized and dtype_out.is_quantized and n == len(ptrs_out) == len(numels) == len(params_ptrs)
        if n == 0:
            return
        self.assume_device_pointers(_device_ptrs)
        C.piquant_hip_quantize_dynamic_batch(self._ctx, (_C.c_void_p * n)(*ptrs_in), dtype_in.value, (_C.c_void_p * n)(*ptrs_out), dtype_out.value,
                                             (_C.c_size_t * n)(*numels), (_C.c_void_p * n)(*params_ptrs), n, round_mode.value)

    def dequantize_dp_batch_ptr(self, ptrs_in, dtype_in: DataType, ptrs_out, dtype_out: DataType, numels, params_ptrs, reduce_op: ReduceOp,
                                _device_ptrs: bool = False) -> None:
        """``dequantize_dp_ptr`` for several independent tensors in one launch per 16 (piquant_hip_dequantize_dp_batch)."""
        n = len(ptrs_in)
        assert dtype_in.is_quantized and dtype_out.is_dequantized and n == len(ptrs_out) == len(numels) == len(params_ptrs)
        if n == 0:
            return
        self.assume_device_pointers(_device_ptrs)
        C.piquant_hip_dequantize_dp_batch(self._ctx, (_C.c_void_p * n)(*ptrs_in), dtype_in.value, (_C.c_void_p * n)(*ptrs_out), dtype_out.value,
                                          (_C.c_size_t * n)(*numels), (_C.c_void_p * n)(*params_ptrs), n, reduce_op.value)

    def reduce_quantize_dynamic_ptr(self, ptr_acc: int, dtype_acc: DataType, ptrs_in, params_in, ptr_out: int, dtype_out: DataType, numel: int,
                                    params_ptr: int, round_mode: RoundMode, _device_ptrs: bool = False) -> None:
        """out = quantize(acc + sum_i dequantize(input i)) with parameters from that sum, left in ``params_ptr`` (include/piquant_hip.h,
        piquant_hip_reduce_quantize_dynamic); the inputs have type ``dtype_out``.  ``acc`` is unspecified afterwards."""
        assert dtype_acc.is_dequantized and dtype_out.is_quantized and len(ptrs_in) == len(params_in) and params_ptr != 0
        n = len(ptrs_in)
        self.assume_device_pointers(_device_ptrs)
        C.piquant_hip_reduce_quantize_dynamic(self._ctx, ptr_acc, dtype_acc.value, (_C.c_void_p * max(n, 1))(*ptrs_in), (_C.c_void_p * max(n, 1))(*params_in), n,
                                              ptr_out, dtype_out.value, numel, params_ptr, round_mode.value)

    def dequantize_sum_ptr(self, ptrs_in, params_ptrs, dtype_in: DataType, ptr_out: int, dtype_out: DataType, numel: int, reduce_op: ReduceOp,
                           _device_ptrs: bool = False) -> None:
        """out (op)= sum of dequantize(input i) over several quantized buffers, each with its own device parameter record, in one
        pass (include/piquant_hip.h, piquant_hip_dequantize_sum); bit-identical to the calls made one after the other."""
        assert dtype_in.is_quantized and dtype_out.is_dequantized and len(ptrs_in) == len(params_ptrs)
        n = len(ptrs_in)
        if n == 0:
            return
        self.assume_device_pointers(_device_ptrs)
        arr_in = (_C.c_void_p * n)(*ptrs_in)
        arr_p = (_C.c_void_p * n)(*params_ptrs)
        C.piquant_hip_dequantize_sum(self._ctx, arr_in, arr_p, n, dtype_in.value, ptr_out, dtype_out.value, numel, reduce_op.value)

    def peer_alloc(self, nbytes: int, fine_grained: bool, fill_word: int = 0):
        """-> (device address, 64-byte IPC handle) of a fresh allocation other GPUs / processes of the node may map (include/piquant_hip.h)."""
        handle = _C.create_string_buffer(64)
        ptr = C.piquant_hip_peer_alloc(self._ctx, nbytes, 1 if fine_grained else 0, fill_word & 0xFFFFFFFF, handle)
        return int(ptr), bytes(handle.raw)

    def peer_open(self, handle: bytes) -> int:
        """Local address of a peer's allocation (its 64-byte IPC handle from ``peer_alloc`` on the peer)."""
        assert len(handle) == 64
        return int(C.piquant_hip_peer_open(self._ctx, _C.create_string_buffer(handle, 64)))

    def peer_close(self, ptr: int) -> None:
        C.piquant_hip_peer_close(self._ctx, ptr)

    def peer_free(self, ptr: int) -> None:
        C.piquant_hip_peer_free(self._ctx, ptr)

    def signal_flags_ptr(self, flag_ptrs, value: int) -> None:
        """Stream-ordered: store ``value`` into every flag address (peers' memory) once the work enqueued so far has completed
        (include/piquant_hip.h, piquant_hip_signal_flags)."""
        n = len(flag_ptrs)
        if n:
            C.piquant_hip_signal_flags(self._ctx, (_C.c_void_p * n)(*flag_ptrs), n, value & 0xFFFFFFFF)

    def wait_flags_ptr(self, flags_ptr: int, count: int, value: int, timeout_us: int = 0) -> None:
        """Stream-ordered: hold the stream until ``count`` uint32 flags at ``flags_ptr`` (this device's memory) have reached ``value``."""
        if count:
            C.piquant_hip_wait_flags(self._ctx, flags_ptr, count, value & 0xFFFFFFFF, timeout_us)

    def exchange_minmax_keys_ptr(self, keys_ptr: int, peer_slot_ptrs, my_slots_ptr: int, out_keys_ptr: int, timeout_us: int = 0) -> None:
        """Stream-ordered MIN all-reduce of this rank's int32[2] key pair over peer-mapped mailboxes (include/piquant_hip.h,
        piquant_hip_exchange_minmax_keys); ``peer_slot_ptrs[j]`` is this rank's slot in rank j's mailbox."""
        n = len(peer_slot_ptrs)
        C.piquant_hip_exchange_minmax_keys(self._ctx, keys_ptr, (_C.c_void_p * n)(*peer_slot_ptrs), my_slots_ptr, n, out_keys_ptr, timeout_us)

    def peer_timeout(self):
        """None, or ``(kind, rank, expected, seen)`` of a peer-to-peer wait of this context that ran out since the last query (``kind``:
        'flags' = ``wait_flags_ptr``, 'keys' = ``exchange_minmax_keys_ptr``); clears the record (include/piquant_hip.h, piquant_hip_peer_timeout)."""
        rank, expected, seen = _C.c_uint32(0), _C.c_uint32(0), _C.c_uint32(0)
        kind = C.piquant_hip_peer_timeout(self._ctx, _C.byref(rank), _C.byref(expected), _C.byref(seen))
        if kind == 0:
            return None
        return ('flags' if kind == 1 else 'keys', int(rank.value), int(expected.value), int(seen.value))

    def set_host_path(self, path: str) -> None:
        """Who serves calls on pageable HOST buffers: 'auto' (default: the companion libpiquant_cpu.so -- the same arithmetic in AVX-512 on the
        host cores, as the reference does with host tensors -- when it is present and the host has AVX-512, PCIe staging otherwise), 'stage'
        (always PCIe staging through the HIP kernels) or 'cpu' (always the companion; abort if it is missing).  Device and pinned buffers
        always run the HIP kernels (include/piquant_hip.h)."""
        self._record_policy('set_host_path', path)
        C.piquant_hip_set_host_path(self._ctx, {'stage': 0, 'cpu': 1, 'auto': 2}[path])

    def host_path_in_effect(self) -> str:
        """'stage' or 'cpu': what a call on pageable host buffers gets from this context right now ('auto' resolved)."""
        return ('stage', 'cpu')[C.piquant_hip_host_path_in_effect(self._ctx)]

    def set_fusion(self, enabled: bool) -> None:
        """False: ``quantize_dynamic`` always runs the scan (with its parameter epilogue) and the quantize kernel as two launches (for A/B timing)."""
        self._record_policy('set_fusion', enabled)
        C.piquant_hip_set_fusion(self._ctx, 1 if enabled else 0)

    def set_independent_calls(self, enabled: bool) -> None:
        """Opt-in: the caller promises that every stream-ordered quantize / dequantize call on this context depends on nothing still in flight on
        the stream (tensor after tensor of a gradient bucket); the launches then go out without the barrier bit and a call's ramp runs under the
        previous one's drain -- 22.9 -> 21.6 us per fp32 -> uint8 call at numel 27 264 000 (include/piquant_hip.h, piquant_hip_set_independent_calls)."""
        self._record_policy('set_independent_calls', enabled)
        C.piquant_hip_set_independent_calls(self._ctx, 1 if enabled else 0)

    def independent_calls(self):
        """``with ctx.independent_calls(): ...`` -- the mode on for the calls inside the block (a gradient bucket quantized tensor after tensor), off again
        behind it, so that whatever follows is ordered behind all of them as usual."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            self.set_independent_calls(True)
            try:
                yield self
            finally:
                self.set_independent_calls(False)

        return scope()

    #: ``set_barrier_timeout_us(Context.HAND_OVER_ALWAYS)``: every block but the last of each tensor hands its share over without waiting
    #: (PIQUANT_HIP_BARRIER_HAND_OVER_ALWAYS; the deterministic form of the hand-over path, for tests)
    HAND_OVER_ALWAYS = 0xffffffff

    def set_barrier_timeout_us(self, microseconds: int) -> None:
        """Longest wait of a block at the fused kernel's grid barrier before it hands its share over and frees its CU
        (include/piquant_hip.h); 0 = default (1 ms)."""
        self._record_policy('set_barrier_timeout_us', microseconds)
        C.piquant_hip_set_barrier_timeout_us(self._ctx, int(microseconds))

    def barrier_bailouts(self) -> int:
        """Blocks of this context's fused launches that ever left their barrier early (0 in normal operation; synchronises)."""
        return int(C.piquant_hip_barrier_bailouts(self._ctx))

    def dequantize_dp_ptr(self, ptr_in: int, dtype_in: DataType, ptr_out: int, dtype_out: DataType, numel: int, params_ptr: int,
                          reduce_op: ReduceOp, _device_ptrs: bool = False) -> None:
        assert dtype_in.is_quantized and dtype_out.is_dequantized and params_ptr != 0
        self.assume_device_pointers(_device_ptrs)
        C.piquant_hip_dequantize_dp(self._ctx, ptr_in, dtype_in.value, ptr_out, dtype_out.value, numel, params_ptr, reduce_op.value)

    def compute_quant_params_dist_ptr(self, ptr: int, dtype: DataType, numel: int, target_quant_dtype: DataType, nccl_comm: int,
                                      _device_ptrs: bool = False) -> Tuple[float, int]:
        """Sharded ``compute_quant_params`` with the all-reduce done natively: ``nccl_comm`` is an ``ncclComm_t`` (RCCL)."""
        assert dtype.is_dequantized and target_quant_dtype.is_quantized and nccl_comm
        scale, zero_point = _C.c_float(), _C.c_int64()
        self.assume_device_pointers(_device_ptrs)
        C.piquant_hip_compute_quant_params_dist(self._ctx, ptr, dtype.value, numel, target_quant_dtype.value, nccl_comm, _C.byref(scale),
                                                _C.byref(zero_point))
        return scale.value, zero_point.value

    def minmax_keys_ptr(self, ptr: int, dtype: DataType, numel: int, device_keys_ptr: int, init: bool = True, _device_ptrs: bool = False) -> None:
        """Asynchronously fold {min, -max} of the buffer into two int32 keys in device memory (atomic MIN)."""
        assert dtype.is_dequantized
        self.assume_device_pointers(_device_ptrs)
        C.piquant_hip_minmax_keys(self._ctx, ptr, dtype.value, numel, device_keys_ptr, 1 if init else 0)


def decode_minmax_keys(k_min: int, k_negmax: int) -> Tuple[float, float]:
    keys = (_C.c_int32 * 2)(k_min, k_negmax)
    lo, hi = _C.c_float(), _C.c_float()
    C.piquant_hip_decode_minmax_keys(keys, _C.byref(lo), _C.byref(hi))
    return lo.value, hi.value


def quant_params_from_minmax(r_min: float, r_max: float, target_quant_dtype: DataType) -> Tuple[float, int]:
    """The reference's (min,max) -> (scale, zero_point) epilogue (src/piquant.cpp:245-258), host side."""
    assert target_quant_dtype.is_quantized
    scale, zero_point = _C.c_float(), _C.c_int64()
    C.piquant_hip_quant_params_from_minmax(r_min, r_max, target_quant_dtype.value, _C.byref(scale), _C.byref(zero_point))
    return scale.value, zero_point.value


def _require_device() -> None:
    if importlib.util.find_spec('torch') is None:
        return   # no torch to ask: the native library reports a missing device itself (message + abort)
    import torch as _torch

    if not _torch.cuda.is_available():
        raise RuntimeError('piquant needs a HIP device: torch.cuda.is_available() is False and there is no CPU path')


def _current_device() -> int:
    import torch as _torch

    _require_device()
    return _torch.cuda.current_device()


def _make_on_device(device_index: int) -> Context:
    import torch as _torch

    with _torch.cuda.device(device_index):
        return Context()


if importlib.util.find_spec('torch') is not None:
    from . import torch  # noqa: E402,F401
