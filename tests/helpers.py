"""Shared test helpers: golden-vector loader and thin wrappers that drive the product THROUGH THE C ABI."""
import json
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"

_cache = None


def load_golden():
    """-> (cases, get) where get(case_name, field) returns the stored numpy array."""
    global _cache
    if _cache is None:
        meta = json.loads((GOLDEN / "ref_vectors.json").read_text())
        blob = np.load(GOLDEN / "ref_vectors.npz")["blob"]
        index = meta["index"]

        def get(name, field):
            dtype, size, off = index[f"{name}.{field}"]
            dt = np.dtype(dtype)
            return blob[off: off + size * dt.itemsize].view(dt).copy()

        _cache = (meta["cases"], get)
    return _cache


def np_dtype_of(code):
    return {0: np.float32, 1: np.uint16}[code]


# ---------------------------------------------------------------------------------------------------
# GPU side: numpy in, numpy out, everything through libpiquant.so's C ABI on device pointers.
# ---------------------------------------------------------------------------------------------------
def _torch():
    import torch

    return torch


def to_device(a: np.ndarray, offset_bytes: int = 0):
    """Copy a numpy array into HBM; returns (tensor_keepalive, data_ptr).  offset_bytes mis-aligns the buffer."""
    torch = _torch()
    raw = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
    buf = torch.empty(raw.size + offset_bytes + 64, dtype=torch.uint8, device="cuda")
    view = buf[offset_bytes: offset_bytes + raw.size]
    if raw.size:
        view.copy_(torch.from_numpy(raw))
    return view, view.data_ptr()


def gpu_quantize(ctx, x, dt_in, dt_out, scale, zp, round_mode=0, offset_in=0, offset_out=0, host=False):
    import piquant

    torch = _torch()
    n = x.size
    nbytes = piquant.DataType(dt_out).packed_nbytes(n)
    if host:
        out = np.full(nbytes, 0xAA, dtype=np.uint8)
        xin = np.ascontiguousarray(x)
        ctx.reset_stream()
        ctx.set_blocking(True)
        if n:
            ctx.quantize_ptr(xin.ctypes.data, piquant.DataType(dt_in), out.ctypes.data, piquant.DataType(dt_out), n, scale, zp,
                             piquant.RoundMode(round_mode))
        return out
    xin, pin = to_device(x, offset_in)
    obuf = torch.full((nbytes + offset_out + 64,), 0xAA, dtype=torch.uint8, device="cuda")
    out = obuf[offset_out: offset_out + nbytes]
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_blocking(False)
    ctx.quantize_ptr(pin if n else 0, piquant.DataType(dt_in), out.data_ptr() if n else 0, piquant.DataType(dt_out), n, scale, zp,
                     piquant.RoundMode(round_mode))
    torch.cuda.synchronize()
    # guard bytes after the output must be untouched
    assert bool((obuf[offset_out + nbytes:] == 0xAA).all()), "kernel wrote past the end of the output"
    assert bool((obuf[:offset_out] == 0xAA).all()), "kernel wrote before the start of the output"
    return out.cpu().numpy()


def gpu_dequantize(ctx, q, dt_in, dt_out, numel, scale, zp, op=0, prev=None, offset_in=0, offset_out=0, host=False):
    import piquant

    torch = _torch()
    odt = np_dtype_of(dt_out)
    if prev is None:
        prev = np.zeros(numel, dtype=odt)
    assert prev.dtype == odt and prev.size == numel
    if host:
        out = prev.copy()
        qin = np.ascontiguousarray(q)
        ctx.reset_stream()
        ctx.set_blocking(True)
        if numel:
            ctx.dequantize_ptr(qin.ctypes.data, piquant.DataType(dt_in), out.ctypes.data, piquant.DataType(dt_out), numel, scale, zp,
                               piquant.ReduceOp(op))
        return out
    qin, pin = to_device(q, offset_in)
    nbytes = prev.nbytes
    obuf = torch.full((nbytes + offset_out + 64,), 0xAA, dtype=torch.uint8, device="cuda")
    out = obuf[offset_out: offset_out + nbytes]
    if nbytes:
        out.copy_(torch.from_numpy(prev.view(np.uint8).reshape(-1)))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_blocking(False)
    ctx.dequantize_ptr(pin if numel else 0, piquant.DataType(dt_in), out.data_ptr() if numel else 0, piquant.DataType(dt_out), numel,
                       scale, zp, piquant.ReduceOp(op))
    torch.cuda.synchronize()
    assert bool((obuf[offset_out + nbytes:] == 0xAA).all()), "kernel wrote past the end of the output"
    return out.cpu().numpy().view(odt).copy()


def same_floats(a: np.ndarray, b: np.ndarray) -> bool:
    """Bit equality, except that any NaN matches any NaN (payloads are outside the contract)."""
    if a.dtype == np.uint16:   # bf16 bit patterns
        na = (a & 0x7FFF) > 0x7F80
        nb = (b & 0x7FFF) > 0x7F80
    else:
        na, nb = np.isnan(a), np.isnan(b)
    if not np.array_equal(na, nb):
        return False
    au = a.view(np.uint32) if a.dtype == np.float32 else a
    bu = b.view(np.uint32) if b.dtype == np.float32 else b
    return bool(np.array_equal(au[~na], bu[~nb]))
