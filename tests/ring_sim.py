"""Reference simulation (numpy + oracle) of piquant.distributed.quantized_all_reduce for G ranks, run in one process.
Used by the CPU (gloo, oracle-backed ops) and GPU (gloo between processes sharing the GPU, real HIP ops) ring tests."""
import numpy as np


def simulate(O, xs, qd, chunks, round_mode=0):
    """xs: list of G float32 arrays (one per rank).  Returns the list of final arrays (all equal)."""
    G = len(xs)
    xs = [x.copy() for x in xs]

    def seg(r, idx):
        b, e = chunks[idx]
        return xs[r][b:e]

    for step in range(G - 1):
        msgs = []
        for r in range(G):
            s = seg(r, (r - step) % G)
            if s.size:
                scale, zp = O.compute_quant_params(s, O.F32, qd)
                msgs.append((O.quantize(s, O.F32, qd, scale, zp, round_mode), scale, zp))
            else:
                msgs.append((None, 1.0, 0))
        for r in range(G):
            q, scale, zp = msgs[(r - 1) % G]
            d = seg(r, (r - step - 1) % G)
            if d.size:
                d[:] = O.dequantize(q, qd, O.F32, d.size, scale, zp, O.ADD, out=d.copy())
    wire = []
    for r in range(G):
        s = seg(r, (r + 1) % G)
        if s.size:
            scale, zp = O.compute_quant_params(s, O.F32, qd)
            q = O.quantize(s, O.F32, qd, scale, zp, round_mode)
            s[:] = O.dequantize(q, qd, O.F32, s.size, scale, zp)
            wire.append((q, scale, zp))
        else:
            wire.append((None, 1.0, 0))
    for step in range(G - 1):
        new = []
        for r in range(G):
            q, scale, zp = wire[(r - 1) % G]
            d = seg(r, (r - step) % G)
            if d.size:
                d[:] = O.dequantize(q, qd, O.F32, d.size, scale, zp)
            new.append((q, scale, zp))
        wire = new
    return xs


def simulate_direct(O, xs, qd, chunks, round_mode=0):
    """piquant.distributed.quantized_all_reduce_direct for G ranks: all-to-all of quantized chunks, the owner adds them to its own
    values in increasing rank order, one more quantization, all-gather."""
    G = len(xs)
    xs = [x.copy() for x in xs]
    final = []
    for owner in range(G):
        b, e = chunks[owner]
        if e == b:
            final.append(None)
            continue
        acc = xs[owner][b:e].copy()
        for src in range(G):
            if src == owner:
                continue
            s = xs[src][b:e]
            scale, zp = O.compute_quant_params(s, O.F32, qd)
            q = O.quantize(s, O.F32, qd, scale, zp, round_mode)
            acc = O.dequantize(q, qd, O.F32, acc.size, scale, zp, O.ADD, out=acc)
        scale, zp = O.compute_quant_params(acc, O.F32, qd)
        q = O.quantize(acc, O.F32, qd, scale, zp, round_mode)
        final.append(O.dequantize(q, qd, O.F32, acc.size, scale, zp))
    for r in range(G):
        for owner in range(G):
            b, e = chunks[owner]
            if e > b:
                xs[r][b:e] = final[owner]
    return xs


class OracleOps:
    """Wire encode / decode on CPU torch tensors through the oracle: stands in for the HIP ops where there is no GPU.
    Same wire format: 16-byte header {float scale, float 1/scale, int64 zero_point} + packed bytes."""

    def __init__(self, O):
        self.O = O

    def _qd(self, qdtype):
        import torch

        return {torch.uint8: 4, torch.quint8: 4, torch.quint4x2: 3, torch.quint2x4: 2}[qdtype]

    def encode(self, x, buf, qdtype, round_mode):
        import struct

        import torch

        scale, zp = self.O.compute_quant_params(x.numpy(), self.O.F32, self._qd(qdtype))
        inv = float(np.float32(1.0) / np.float32(scale))
        buf[:16].copy_(torch.frombuffer(bytearray(struct.pack("<ffq", scale, inv, zp)), dtype=torch.uint8))
        buf[16:].copy_(torch.from_numpy(self.O.quantize(x.numpy(), self.O.F32, self._qd(qdtype), scale, zp, 0)))

    def decode(self, buf, out, qdtype, reduce_op):
        import struct

        import torch

        scale, _inv, zp = struct.unpack("<ffq", buf[:16].numpy().tobytes())
        res = self.O.dequantize(buf[16:].numpy(), self._qd(qdtype), self.O.F32, out.numel(), scale, zp, 1 if reduce_op == "add" else 0,
                                out=out.numpy().copy())
        out.copy_(torch.from_numpy(res))

    def decode_sum(self, bufs, out, qdtype):
        for buf in bufs:
            self.decode(buf, out, qdtype, "add")

    def encode_batch(self, xs, bufs, qdtype, round_mode):
        for x, buf in zip(xs, bufs):
            self.encode(x, buf, qdtype, round_mode)

    def decode_batch(self, bufs, outs, qdtype, reduce_op):
        for buf, out in zip(bufs, outs):
            self.decode(buf, out, qdtype, reduce_op)

    def reduce_encode(self, bufs, acc, buf, qdtype, round_mode):
        self.decode_sum(bufs, acc, qdtype)
        self.encode(acc, buf, qdtype, round_mode)
