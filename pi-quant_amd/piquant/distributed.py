"""Multi-GPU use of the path: one process per GPU, ``torch.distributed`` (backend ``nccl`` == RCCL over xGMI).

The reference is a single-process CPU library; its only "parallelism" is a static split of one flat array
over pool threads (``src/piquant.cpp:132-176``).  Here the same split rule shards a tensor over ranks:

* ``quantize`` / ``dequantize`` are element-local -> every rank processes its own shard, **no collective**;
* ``compute_quant_params`` needs the global min/max -> each rank scans its shard on its GPU into two int32
  keys {key(min), key(-max)}; **one** 8-byte ``all_reduce(MIN)`` combines them; every rank then runs the same
  double-precision epilogue and obtains identical ``(scale, zero_point)``.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist

from . import Context, DataType, decode_minmax_keys, quant_params_from_minmax
from .torch import _QUANT_TYPES, _ctx_for, torch_to_piquant_dtype


def shard_range(numel: int, rank: int, world_size: int, packed_bits: int = 8) -> Tuple[int, int]:
    """[begin, end) of ``rank``'s shard: the reference's range split (``src/piquant.cpp:145-157``) -- boundaries are
    aligned down to a whole packed byte (2 elements for uint4, 4 for uint2); the last rank keeps the ragged end."""
    world_size = max(1, world_size)
    pack = 8 // packed_bits if packed_bits < 8 else 1
    begin = numel * rank // world_size
    end = numel * (rank + 1) // world_size
    if pack > 1:
        begin -= begin % pack
        if rank + 1 != world_size:
            end -= end % pack
    return begin, max(begin, end)


def local_minmax_keys(tensor: torch.Tensor, ctx: Optional[Context] = None) -> torch.Tensor:
    """int32[2] tensor on ``tensor.device`` holding {key(min), key(-max)} of the local shard (HIP scan, async)."""
    if not tensor.is_cuda:
        raise RuntimeError('local_minmax_keys needs a ROCm device tensor: the min/max scan is a HIP kernel, there is no CPU path')
    if not tensor.is_contiguous():
        tensor = tensor.contiguous()
    ctx = _ctx_for(tensor, ctx)
    keys = torch.empty(2, dtype=torch.int32, device=tensor.device)
    ctx.minmax_keys_ptr(tensor.data_ptr(), torch_to_piquant_dtype(tensor.dtype), tensor.numel(), keys.data_ptr(), init=True)
    return keys


def compute_quant_params(
    local_shard: torch.Tensor,
    *,
    dtype: torch.dtype,
    group: Optional[dist.ProcessGroup] = None,
    ctx: Optional[Context] = None,
    _scan=local_minmax_keys,
) -> Tuple[float, int]:
    """Quantization parameters of the tensor whose shards are spread over ``group`` (identical on every rank)."""
    assert dtype in _QUANT_TYPES, f'Unsupported quantized dtype: {dtype}'
    keys = _scan(local_shard, ctx)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(keys, op=dist.ReduceOp.MIN, group=group)   # the path's only collective: 8 bytes
    k = keys.cpu()
    r_min, r_max = decode_minmax_keys(int(k[0]), int(k[1]))
    return quant_params_from_minmax(r_min, r_max, torch_to_piquant_dtype(dtype))
