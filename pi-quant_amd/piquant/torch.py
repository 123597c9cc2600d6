"""``piquant.torch`` -- the reference's tensor-level API on PyTorch-ROCm tensors.

Same functions and keyword arguments as the reference module (``python/src/piquant/torch.py:40-129``):
``torch_to_piquant_dtype``, ``piquant_to_torch_dtype``, ``compute_quant_params``, ``quantize``,
``dequantize``.  Differences, all additive:

* tensors may live on a ROCm device; the output is allocated on ``tensor.device`` and the kernels are
  enqueued on the current PyTorch stream (ordinary PyTorch stream semantics, no host sync);
* CPU tensors still work: their host buffers are served where they live by the companion library ``libpiquant_cpu.so`` (the default since
  round 4, synchronous on host threads) or, without it / with ``PIQUANT_HIP_HOST_PATH=stage``, staged through the GPU over PCIe;
* ``out=`` lets ``reduce_op='add'`` accumulate into an existing tensor -- the reference allocates a fresh
  uninitialised output (``torch.py:117``), which makes ADD unusable through its tensor API;
* ``ctx=None`` resolves to the default context of the tensor's device at call time (the reference evaluates
  ``Context.get()`` once, at import).

Quantized results are real ``torch.quint8`` / ``torch.quint4x2`` / ``torch.quint2x4`` tensors of the input's shape,
exactly as in the reference; PyTorch-ROCm can allocate them on the device (``torch.empty(shape, dtype=torch.quint4x2,
device='cuda')``) even though it implements hardly any operator on them -- ``packed_bytes`` gives the raw bytes.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import Context, DataType, ReduceOp, RoundMode

_TORCH_DTYPE_MAP: dict[torch.dtype, DataType] = {
    torch.float32: DataType.F32,
    torch.bfloat16: DataType.BF16,
    torch.quint2x4: DataType.UINT2,
    torch.quint4x2: DataType.UINT4,
    torch.quint8: DataType.UINT8,
    torch.uint8: DataType.UINT8,
}

# Native front end (csrc/torch_binding.cpp): checks, output allocation, current stream and the C ABI call in one C++ function.
# Optional -- it only removes Python/ctypes overhead (8.8 -> ~4.5 us per call on a 10^6-element tensor); without it the same C
# ABI entry points are reached through ctypes below.
try:
    from . import _piquant_torch as _native
except ImportError:
    _native = None

_QUANT_TYPES: set[torch.dtype] = {torch.quint2x4, torch.quint4x2, torch.quint8, torch.uint8}
_DEQUANT_TYPES: set[torch.dtype] = {torch.float32, torch.bfloat16}
_ROUND_MODES: dict[str, RoundMode] = {'nearest': RoundMode.NEAREST, 'stochastic': RoundMode.STOCHASTIC}
_REDUCE_OPS: dict[str, ReduceOp] = {'set': ReduceOp.SET, 'add': ReduceOp.ADD}
_ROUND_MODE_CODES: dict[str, int] = {k: v.value for k, v in _ROUND_MODES.items()}   # plain ints for the native front end
_REDUCE_OP_CODES: dict[str, int] = {k: v.value for k, v in _REDUCE_OPS.items()}


def torch_to_piquant_dtype(dtype: torch.dtype) -> DataType:
    if dtype not in _TORCH_DTYPE_MAP:
        raise ValueError(f'Unsupported quant_dtype: {dtype} (float32, bfloat16, uint8 / quint8, quint4x2 and quint2x4 have a piquant counterpart)')
    return _TORCH_DTYPE_MAP[dtype]


def piquant_to_torch_dtype(dtype: DataType) -> torch.dtype:
    """First torch dtype mapped to ``dtype`` (the intent of reference ``torch.py:46-50``)."""
    for torch_dtype, piquant_dtype in _TORCH_DTYPE_MAP.items():
        if piquant_dtype == dtype:
            return torch_dtype
    raise ValueError(f'Unsupported quantized dtype: {dtype} (no torch dtype is mapped to it)')


def packed_bytes(tensor: torch.Tensor) -> torch.Tensor:
    """1-D uint8 view of the bytes that hold a quantized tensor: ``ceil(numel * bits / 8)`` of them, lower element
    index in the lower bits (PyTorch's own packing for quint4x2 / quint2x4 and the reference's)."""
    dt = torch_to_piquant_dtype(tensor.dtype)
    raw = torch.empty(0, dtype=torch.uint8, device=tensor.device).set_(tensor.untyped_storage())
    first = tensor.storage_offset() * max(dt.bit_size, 8) // 8
    return raw[first: first + dt.packed_nbytes(tensor.numel())]


# torch.cuda.current_stream(i).cuda_stream builds a Stream object per call (~1.2 us); the raw getter is ~5x cheaper
_current_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None) or (lambda index: torch.cuda.current_stream(index).cuda_stream)


def _ctx_for(tensor: torch.Tensor, ctx: Optional[Context]) -> Context:
    """Default context of the tensor's device, with the current PyTorch stream attached for device tensors."""
    if tensor.is_cuda:
        index = tensor.device.index if tensor.device.index is not None else torch.cuda.current_device()
        if ctx is None:
            ctx = Context.get(index)
        elif ctx.device != index:
            raise ValueError(f'context is bound to device {ctx.device} but the tensor lives on device {index}')
        ctx.set_stream(_current_raw_stream(index))
        ctx.set_blocking(False)   # stream-ordered, like every other PyTorch device op
    else:
        if ctx is None:
            ctx = Context.get()
        ctx.reset_stream()        # host buffers: staged on the context's own streams, complete on return
        ctx.set_blocking(True)
    return ctx


def _resolve_default_handle(index: int) -> int:
    """Called by the native front end once per (thread, device): the native handle of the thread's default context of that device.  The front end
    pushes stream / non-blocking / device-pointer mode to the native context itself on every call, so the Python-side cache of pushed settings is
    switched off for this context (``Context._native_managed``): a later ctypes call pushes what it needs whatever the cache says."""
    ctx = Context._thread_defaults().get(index) or Context.get(index)
    ctx._native_managed = True
    return ctx._native_call()


if _native is not None:
    _native.set_default_resolver(_resolve_default_handle)


def _native_handle(tensor: torch.Tensor, ctx: Optional[Context]) -> int:
    """Native context handle for a call through the C++ front end (default context of the tensor's device unless one is given)."""
    if ctx is None:
        index = tensor.device.index
        ctx = Context._thread_defaults().get(index) or Context.get(index)
    return ctx._native_call()


def _quant_meta(tensor: torch.Tensor, quant_dtype: Optional[torch.dtype], shape) -> Tuple[DataType, torch.Size]:
    """Quantized dtype and logical shape of a dequantize input: from the tensor itself, or -- for a raw uint8 buffer of
    packed bytes -- from the ``quant_dtype=`` / ``shape=`` keywords."""
    if quant_dtype is not None and quant_dtype != tensor.dtype:
        _require(tensor.dtype == torch.uint8, 'quant_dtype= reinterprets a raw uint8 byte buffer')
        _require(shape is not None, 'shape= is required together with quant_dtype= for raw packed buffers')
        return torch_to_piquant_dtype(quant_dtype), torch.Size(shape)
    return torch_to_piquant_dtype(tensor.dtype), tensor.shape


# Argument checks of the additive entry points.  They raise (ValueError) rather than assert: a short, misplaced or non-contiguous
# buffer handed to a kernel by raw pointer is an out-of-bounds device write, and `python -O` strips asserts.
def _require(cond: bool, msg: str) -> None:
    if not cond:
        raise ValueError(msg)


def _check_float_input(t: torch.Tensor, what: str = 'tensor') -> None:
    _require(isinstance(t, torch.Tensor) and t.is_cuda and t.dtype in _DEQUANT_TYPES, f'{what} must be a float32 or bfloat16 ROCm device tensor')


def _check_packed_out(out: torch.Tensor, dt: DataType, numel: int, device: torch.device, what: str = 'out') -> None:
    """`out` receives `numel` quantized elements: a raw uint8 buffer of at least packed_nbytes bytes, or a quantized torch tensor of numel elements."""
    _require(isinstance(out, torch.Tensor) and out.device == device and out.is_contiguous(), f'{what} must be a contiguous tensor on {device}')
    if out.dtype == torch.uint8:
        _require(out.numel() >= dt.packed_nbytes(numel), f'{what} holds {out.numel()} bytes, {dt.packed_nbytes(numel)} are needed for {numel} {dt.name} elements')
    else:
        _require(out.dtype in _QUANT_TYPES and torch_to_piquant_dtype(out.dtype) == dt and out.numel() == numel,
                 f'{what} must be a {dt.name} tensor of {numel} elements (or a uint8 buffer of the packed bytes)')


def _check_packed_in(t: torch.Tensor, dt: DataType, numel: int, device: torch.device, what: str) -> None:
    """`t` supplies `numel` quantized elements: a raw uint8 buffer of at least packed_nbytes bytes, or a quantized torch tensor of that dtype."""
    _require(isinstance(t, torch.Tensor) and t.device == device and t.is_contiguous(), f'{what} must be a contiguous tensor on {device}')
    if t.dtype == torch.uint8:
        _require(t.numel() >= dt.packed_nbytes(numel), f'{what} holds {t.numel()} bytes, {dt.packed_nbytes(numel)} are needed for {numel} {dt.name} elements')
    else:
        _require(t.dtype in _QUANT_TYPES and torch_to_piquant_dtype(t.dtype) == dt and t.numel() >= numel, f'{what} does not hold {numel} {dt.name} elements')


def _check_float_out(out: torch.Tensor, dtype: torch.dtype, numel: int, device: torch.device, what: str = 'out') -> None:
    _require(isinstance(out, torch.Tensor) and out.dtype == dtype and out.is_contiguous() and out.device == device and out.numel() == numel,
             f'{what} must be a contiguous {dtype} tensor of {numel} elements on {device}')


def _check_params(params: torch.Tensor, device: torch.device, what: str = 'params') -> None:
    _require(isinstance(params, torch.Tensor) and params.dtype == torch.uint8 and params.device == device and params.is_contiguous() and
             params.numel() >= PARAMS_NBYTES and params.data_ptr() % 8 == 0,
             f'{what} must be a contiguous uint8 tensor of at least {PARAMS_NBYTES} bytes on {device}, 8-byte aligned (the device parameter record)')


def _numel_of(shape) -> int:
    n = 1
    for s_ in shape:
        n *= int(s_)
    return n


def compute_quant_params(tensor: torch.Tensor, *, dtype: torch.dtype, ctx: Optional[Context] = None) -> Tuple[float, int]:
    """(scale, zero_point) from the tensor's min/max (reference ``torch.py:53-67``)."""
    assert dtype in _QUANT_TYPES, f'Unsupported quantized dtype: {dtype}; choose from {[str(t) for t in _QUANT_TYPES]}'
    if not tensor.is_contiguous():
        tensor = tensor.contiguous()
    ctx = _ctx_for(tensor, ctx)
    if tensor.dtype == torch.bfloat16:
        return ctx.compute_quant_params_ptr_bfloat16(tensor.data_ptr(), torch_to_piquant_dtype(dtype), tensor.numel(), _device_ptrs=tensor.is_cuda)
    assert tensor.dtype == torch.float32, f'compute_quant_params needs float32 or bfloat16, got {tensor.dtype}'
    return ctx.compute_quant_params_ptr_float32(tensor.data_ptr(), torch_to_piquant_dtype(dtype), tensor.numel(), _device_ptrs=tensor.is_cuda)


def quantize(
    tensor: torch.Tensor,
    *,
    scale: float,
    zero_point: int,
    dtype: torch.dtype,
    round_mode: str = 'nearest',
    ctx: Optional[Context] = None,
    out: Optional[torch.Tensor] = None,
    uniform: bool = False,
) -> torch.Tensor:
    """Reference ``torch.py:70-99``; the result lives on ``tensor.device``.  The bytes are those of a reference context with the
    context's ``num_threads`` (``Context.set_reference_layout``); ``uniform=True`` (additive) asks for the position-independent form
    instead -- what shards of one logical tensor must be computed with (``piquant.distributed``)."""
    if _native is not None and ctx is None and tensor.is_cuda:      # the whole call in C++: checks, default context, output, stream, the C ABI call
        return _native.quantize_default(tensor, scale, zero_point, dtype, round_mode, out, uniform)
    assert dtype in _QUANT_TYPES, f'Unsupported quantized dtype: {dtype}; choose from {[str(t) for t in _QUANT_TYPES]}'
    if _native is not None and tensor.is_cuda and tensor.dtype in _DEQUANT_TYPES:
        return _native.quantize(_native_handle(tensor, ctx), tensor, scale, zero_point, dtype, _ROUND_MODE_CODES[round_mode], out, uniform)
    if not tensor.is_contiguous():
        tensor = tensor.contiguous()
    dtype_in = torch_to_piquant_dtype(tensor.dtype)
    dtype_out = torch_to_piquant_dtype(dtype)
    if out is None:
        out = torch.empty(tensor.shape, dtype=dtype, device=tensor.device)   # reference torch.py:87, plus the device
    else:
        _check_packed_out(out, dtype_out, tensor.numel(), tensor.device)
    ctx = _ctx_for(tensor, ctx)
    ctx.quantize_ptr(
        tensor.data_ptr(),
        dtype_in,
        out.data_ptr(),
        dtype_out,
        numel=tensor.numel(),
        scale=scale,
        zero_point=zero_point,
        round_mode=_ROUND_MODES[round_mode],
        _device_ptrs=tensor.is_cuda,
        uniform=uniform,
    )
    return out


def dequantize(
    tensor: torch.Tensor,
    *,
    scale: float,
    zero_point: int,
    dtype: torch.dtype,
    reduce_op: str = 'set',
    ctx: Optional[Context] = None,
    out: Optional[torch.Tensor] = None,
    quant_dtype: Optional[torch.dtype] = None,
    shape=None,
    uniform: bool = False,
) -> torch.Tensor:
    """Reference ``torch.py:102-129``.  ``out=`` (same shape, ``dtype``) is the accumulator for ``reduce_op='add'``; ``uniform``: see ``quantize``."""
    if dtype not in _DEQUANT_TYPES:
        raise ValueError(f'Unsupported dequantized dtype: {dtype}; choose from {[str(t) for t in _DEQUANT_TYPES]}')
    if _native is not None and ctx is None and quant_dtype is None and shape is None and tensor.is_cuda and tensor.dtype in _QUANT_TYPES:
        return _native.dequantize_default(tensor, scale, zero_point, dtype, reduce_op, out, uniform)
    if _native is not None and tensor.is_cuda and quant_dtype is None and shape is None and tensor.dtype in _QUANT_TYPES:
        if out is None and reduce_op == 'add':
            raise ValueError("reduce_op='add' accumulates into out=; pass the accumulator tensor")
        return _native.dequantize(_native_handle(tensor, ctx), tensor, scale, zero_point, dtype, _REDUCE_OP_CODES[reduce_op], out, uniform)
    if not tensor.is_contiguous():
        tensor = tensor.contiguous()
    dtype_in, logical_shape = _quant_meta(tensor, quant_dtype, shape)
    numel = 1
    for s in logical_shape:
        numel *= int(s)
    if out is None:
        if reduce_op == 'add':
            raise ValueError("reduce_op='add' accumulates into out=; pass the accumulator tensor")
        out = torch.empty(logical_shape, dtype=dtype, device=tensor.device)
    else:
        _check_float_out(out, dtype, numel, tensor.device)
    ctx = _ctx_for(tensor, ctx)
    ctx.dequantize_ptr(
        tensor.data_ptr(),
        dtype_in,
        out.data_ptr(),
        torch_to_piquant_dtype(out.dtype),
        numel=numel,
        scale=scale,
        zero_point=zero_point,
        reduce_op=_REDUCE_OPS[reduce_op],
        _device_ptrs=tensor.is_cuda,
        uniform=uniform,
    )
    return out


# With the native front end built, the two functions above ARE its entry points: keyword arguments are parsed in C++, a device tensor with the default
# context never enters a Python frame, and everything else comes back to the implementations above (kept under their own names for that).
_quantize_py, _dequantize_py = quantize, dequantize
if _native is not None:
    _native.set_python_implementations(_quantize_py, _dequantize_py)
    quantize, dequantize = _native.quantize_entry, _native.dequantize_entry


def quantize_dequantize(
    tensor: torch.Tensor,
    *,
    scale: float,
    zero_point: int,
    quant_dtype: torch.dtype,
    round_mode: str = 'nearest',
    reduce_op: str = 'set',
    ctx: Optional[Context] = None,
    out: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """out (op)= dequantize(quantize(tensor)) in one pass over HBM -- the reference's C++-only
    ``context::quantize_dequantize_fused`` (``include/piquant.hpp:276-285``); ``out`` may be ``tensor`` (in place)."""
    _require(quant_dtype in _QUANT_TYPES, f'{quant_dtype} is not a quantized dtype')
    _check_float_input(tensor)
    if not tensor.is_contiguous():
        tensor = tensor.contiguous()
    if out is None:
        if reduce_op == 'add':
            raise ValueError("reduce_op='add' accumulates into out=; pass the accumulator tensor")
        out = torch.empty_like(tensor)
    else:
        _check_float_out(out, tensor.dtype, tensor.numel(), tensor.device)
    ctx = _ctx_for(tensor, ctx)
    ctx.quantize_dequantize_ptr(tensor.data_ptr(), torch_to_piquant_dtype(tensor.dtype), out.data_ptr(), torch_to_piquant_dtype(quant_dtype),
                                tensor.numel(), scale, zero_point, _ROUND_MODES[round_mode], _REDUCE_OPS[reduce_op], _device_ptrs=True)
    return out


# -----------------------------------------------------------------------------------------------------------------
# Device-resident parameters (additive): no host round trip between the min/max scan and its consumers.
# -----------------------------------------------------------------------------------------------------------------
PARAMS_NBYTES = 16   # piquant_hip_params_t: float scale, float 1/scale, int64 zero_point


def params_to_host(params: torch.Tensor) -> Tuple[float, int]:
    """(scale, zero_point) of a device parameter record (synchronises)."""
    import struct

    scale, _inv, zp = struct.unpack('<ffq', bytes(params[:PARAMS_NBYTES].cpu().numpy().tobytes()))
    return scale, zp


def compute_quant_params_device(tensor: torch.Tensor, *, dtype: torch.dtype, ctx: Optional[Context] = None,
                                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Like ``compute_quant_params`` but asynchronous: the result is a 16-byte uint8 device tensor (the parameter record)."""
    _require(dtype in _QUANT_TYPES, f'{dtype} is not a quantized dtype')
    _check_float_input(tensor)
    if not tensor.is_contiguous():
        tensor = tensor.contiguous()
    if out is None:
        out = torch.empty(PARAMS_NBYTES, dtype=torch.uint8, device=tensor.device)
    _check_params(out, tensor.device, 'out')
    ctx = _ctx_for(tensor, ctx)
    ctx.compute_quant_params_device_ptr(tensor.data_ptr(), torch_to_piquant_dtype(tensor.dtype), tensor.numel(), torch_to_piquant_dtype(dtype),
                                        out.data_ptr(), _device_ptrs=True)
    return out


def quantize_dynamic(tensor: torch.Tensor, *, dtype: torch.dtype, round_mode: str = 'nearest', ctx: Optional[Context] = None,
                     out: Optional[torch.Tensor] = None, params: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """``compute_quant_params`` + ``quantize`` in one asynchronous call, parameters computed and kept on the device.  Returns
    (quantized, parameter record).  A tensor that fits on the chip (up to ~113 MB on an MI355X) is read from HBM once, by a single
    kernel that keeps it in registers / LDS between the min/max pass and the quantization; larger ones take two launches (scan with the parameter epilogue, then quantize)."""
    _require(dtype in _QUANT_TYPES, f'{dtype} is not a quantized dtype')
    _check_float_input(tensor)
    if not tensor.is_contiguous():
        tensor = tensor.contiguous()
    if params is None:
        params = torch.empty(PARAMS_NBYTES, dtype=torch.uint8, device=tensor.device)
    _check_params(params, tensor.device)
    if out is None:
        out = torch.empty(tensor.shape, dtype=dtype, device=tensor.device)
    else:
        _check_packed_out(out, torch_to_piquant_dtype(dtype), tensor.numel(), tensor.device)
    ctx = _ctx_for(tensor, ctx)
    ctx.quantize_dynamic_ptr(tensor.data_ptr(), torch_to_piquant_dtype(tensor.dtype), out.data_ptr(), torch_to_piquant_dtype(dtype), tensor.numel(),
                             params.data_ptr(), _ROUND_MODES[round_mode], _device_ptrs=True)
    return out, params


def dequantize_dynamic(tensor: torch.Tensor, params: torch.Tensor, *, dtype: torch.dtype, reduce_op: str = 'set',
                       ctx: Optional[Context] = None, out: Optional[torch.Tensor] = None, quant_dtype: Optional[torch.dtype] = None,
                       shape=None) -> torch.Tensor:
    """``dequantize`` with (scale, zero_point) read from a device parameter record."""
    _require(dtype in _DEQUANT_TYPES, f'{dtype} is not a float dtype to dequantize into')
    _require(isinstance(tensor, torch.Tensor) and tensor.is_cuda, 'dequantize_dynamic needs a ROCm device tensor')
    if not tensor.is_contiguous():
        tensor = tensor.contiguous()
    dtype_in, logical_shape = _quant_meta(tensor, quant_dtype, shape)
    numel = _numel_of(logical_shape)
    _check_packed_in(tensor, dtype_in, numel, tensor.device, 'tensor')
    _check_params(params, tensor.device)
    if out is None:
        if reduce_op == 'add':
            raise ValueError("reduce_op='add' accumulates into out=; pass the accumulator tensor")
        out = torch.empty(logical_shape, dtype=dtype, device=tensor.device)
    else:
        _check_float_out(out, dtype, numel, tensor.device)
    ctx = _ctx_for(tensor, ctx)
    ctx.dequantize_dp_ptr(tensor.data_ptr(), dtype_in, out.data_ptr(), torch_to_piquant_dtype(out.dtype), numel, params.data_ptr(),
                          _REDUCE_OPS[reduce_op], _device_ptrs=True)
    return out


def dequantize_sum(tensors, params, *, dtype: torch.dtype, reduce_op: str = 'set', ctx: Optional[Context] = None,
                   out: Optional[torch.Tensor] = None, quant_dtype: Optional[torch.dtype] = None, shape=None) -> torch.Tensor:
    """out (op)= sum_i dequantize(tensors[i]) with (scale, zero_point) of input i read from the device record ``params[i]``: one pass
    over the accumulator instead of ``len(tensors)``; the result equals ``dequantize_dynamic`` applied in order (first with
    ``reduce_op``, the rest with 'add') bit for bit.  The reduction step of ``piquant.distributed.quantized_all_reduce``."""
    _require(dtype in _DEQUANT_TYPES, f'{dtype} is not a float dtype to dequantize into')
    _require(len(tensors) == len(params) and len(tensors) > 0, 'dequantize_sum needs as many parameter records as tensors, and at least one')
    first = tensors[0]
    _require(isinstance(first, torch.Tensor) and first.is_cuda, 'dequantize_sum needs ROCm device tensors')
    dtype_in, logical_shape = _quant_meta(first, quant_dtype, shape)
    numel = _numel_of(logical_shape)
    for i, (t, p) in enumerate(zip(tensors, params)):
        _check_packed_in(t, dtype_in, numel, first.device, f'tensors[{i}]')
        _check_params(p, first.device, f'params[{i}]')
    if out is None:
        if reduce_op == 'add':
            raise ValueError("reduce_op='add' accumulates into out=; pass the accumulator tensor")
        out = torch.empty(logical_shape, dtype=dtype, device=first.device)
    else:
        _check_float_out(out, dtype, numel, first.device)
    ctx = _ctx_for(first, ctx)
    ctx.dequantize_sum_ptr([t.data_ptr() for t in tensors], [p.data_ptr() for p in params], dtype_in, out.data_ptr(), torch_to_piquant_dtype(out.dtype),
                           numel, _REDUCE_OPS[reduce_op], _device_ptrs=True)
    return out


def quantize_dynamic_batch(tensors, *, dtype: torch.dtype, round_mode: str = 'nearest', ctx: Optional[Context] = None, outs=None, params=None):
    """``quantize_dynamic`` for a list of independent tensors of one float dtype: each gets its own (scale, zero_point) and record,
    up to 16 of them are processed by ONE kernel launch.  Returns (list of quantized tensors, list of parameter records)."""
    _require(dtype in _QUANT_TYPES, f'{dtype} is not a quantized dtype')
    _require(len(tensors) > 0, 'quantize_dynamic_batch needs at least one tensor')
    fdt = tensors[0].dtype
    for i, t in enumerate(tensors):
        _check_float_input(t, f'tensors[{i}]')
        _require(t.dtype == fdt and t.device == tensors[0].device, 'all tensors of a batch share one float dtype and one device')
    tensors = [t if t.is_contiguous() else t.contiguous() for t in tensors]
    if outs is None:
        outs = [torch.empty(t.shape, dtype=dtype, device=t.device) for t in tensors]
    if params is None:
        block = torch.empty(len(tensors) * PARAMS_NBYTES, dtype=torch.uint8, device=tensors[0].device)
        params = [block[i * PARAMS_NBYTES: (i + 1) * PARAMS_NBYTES] for i in range(len(tensors))]
    _require(len(outs) == len(params) == len(tensors), 'outs= and params= must have one entry per tensor')
    qdt = torch_to_piquant_dtype(dtype)
    for i, (t, o, p) in enumerate(zip(tensors, outs, params)):
        _check_packed_out(o, qdt, t.numel(), t.device, f'outs[{i}]')
        _check_params(p, t.device, f'params[{i}]')
    ctx = _ctx_for(tensors[0], ctx)
    ctx.quantize_dynamic_batch_ptr([t.data_ptr() for t in tensors], torch_to_piquant_dtype(fdt), [o.data_ptr() for o in outs], torch_to_piquant_dtype(dtype),
                                   [t.numel() for t in tensors], [p.data_ptr() for p in params], _ROUND_MODES[round_mode], _device_ptrs=True)
    return outs, params


def dequantize_dynamic_batch(tensors, params, *, dtype: torch.dtype, reduce_op: str = 'set', ctx: Optional[Context] = None, outs=None,
                             quant_dtype: Optional[torch.dtype] = None, shapes=None):
    """``dequantize_dynamic`` for a list of independent quantized tensors (raw uint8 buffers with ``quant_dtype=`` and ``shapes=``, or
    quantized torch tensors) in one launch per 16; ``outs`` are required for ``reduce_op='add'``."""
    _require(dtype in _DEQUANT_TYPES, f'{dtype} is not a float dtype to dequantize into')
    _require(len(tensors) == len(params) and len(tensors) > 0, 'dequantize_dynamic_batch needs as many parameter records as tensors, and at least one')
    _require(all(isinstance(t, torch.Tensor) and t.is_cuda for t in tensors), 'dequantize_dynamic_batch needs ROCm device tensors')
    metas = [_quant_meta(t, quant_dtype, None if shapes is None else shapes[i]) for i, t in enumerate(tensors)]
    dtype_in = metas[0][0]
    _require(all(m[0] == dtype_in for m in metas), 'all tensors of a batch share one quantized dtype')
    numels = [_numel_of(shp) for _dt, shp in metas]
    device = tensors[0].device
    for i, (t, p, n) in enumerate(zip(tensors, params, numels)):
        _check_packed_in(t, dtype_in, n, device, f'tensors[{i}]')
        _check_params(p, device, f'params[{i}]')
    if outs is None:
        if reduce_op == 'add':
            raise ValueError("reduce_op='add' accumulates into outs=; pass the accumulator tensors")
        outs = [torch.empty(m[1], dtype=dtype, device=t.device) for m, t in zip(metas, tensors)]
    _require(len(outs) == len(tensors), 'outs= must have one entry per tensor')
    for i, (o, n) in enumerate(zip(outs, numels)):
        _check_float_out(o, dtype, n, device, f'outs[{i}]')
    ctx = _ctx_for(tensors[0], ctx)
    ctx.dequantize_dp_batch_ptr([t.data_ptr() for t in tensors], dtype_in, [o.data_ptr() for o in outs], torch_to_piquant_dtype(dtype), numels,
                                [p.data_ptr() for p in params], _REDUCE_OPS[reduce_op], _device_ptrs=True)
    return outs


def reduce_quantize_dynamic(acc: torch.Tensor, tensors, params, *, dtype: torch.dtype, round_mode: str = 'nearest', ctx: Optional[Context] = None,
                            out: Optional[torch.Tensor] = None, out_params: Optional[torch.Tensor] = None):
    """(quantize(acc + sum_i dequantize(tensors[i])), record): the owner's step of a mesh all-reduce as one call -- one kernel launch
    that never writes the sum to memory when it stays on chip.  ``tensors`` are raw uint8 buffers of packed ``dtype`` values with
    ``acc.numel()`` elements each, ``params`` their device records.  The contents of ``acc`` afterwards are unspecified."""
    _require(dtype in _QUANT_TYPES, f'{dtype} is not a quantized dtype')
    _check_float_input(acc, 'acc')
    _require(acc.is_contiguous(), 'acc must be contiguous')
    _require(len(tensors) == len(params), 'reduce_quantize_dynamic needs as many parameter records as tensors')
    qdt = torch_to_piquant_dtype(dtype)
    for i, (t, p) in enumerate(zip(tensors, params)):
        _check_packed_in(t, qdt, acc.numel(), acc.device, f'tensors[{i}]')
        _check_params(p, acc.device, f'params[{i}]')
    if out is None:
        out = torch.empty(acc.shape, dtype=dtype, device=acc.device)
    else:
        _check_packed_out(out, qdt, acc.numel(), acc.device)
    if out_params is None:
        out_params = torch.empty(PARAMS_NBYTES, dtype=torch.uint8, device=acc.device)
    _check_params(out_params, acc.device, 'out_params')
    ctx = _ctx_for(acc, ctx)
    ctx.reduce_quantize_dynamic_ptr(acc.data_ptr(), torch_to_piquant_dtype(acc.dtype), [t.data_ptr() for t in tensors], [p.data_ptr() for p in params],
                                    out.data_ptr(), torch_to_piquant_dtype(dtype), acc.numel(), out_params.data_ptr(), _ROUND_MODES[round_mode],
                                    _device_ptrs=True)
    return out, out_params
