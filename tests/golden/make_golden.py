"""Generates tests/golden/*.npz by RUNNING THE REFERENCE'S OWN KERNELS (oracle/_ref, compiled from
/root/reference by oracle/Makefile) on small seeded inputs.  Only inputs and the reference's outputs are
stored -- data, no reference source.  Re-run in a container that has /root/reference:

    make -C oracle ref && python tests/golden/make_golden.py

Each case stores the reference output for ONE contiguous range (== a reference context with one thread,
AVX-512F unit) plus `uniform`: the position-independent "SIMD-body formula everywhere" result the HIP
kernels implement (oracle FORM_UNIFORM).  `uniform` and `ref` differ only at the documented self-
inconsistencies of the reference (scalar head/tail vs SIMD body); the CPU tests assert exactly that.
"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import oracle as O  # noqa: E402

OUT = Path(__file__).resolve().parent
LENGTHS = [0, 1, 2, 3, 4, 5, 15, 16, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1000, 4099]
F32_NAMES = {O.F32: "f32", O.BF16: "bf16"}
Q_NAMES = {O.UINT8: "u8", O.UINT4: "u4", O.UINT2: "u2"}


def edge_values(scale):
    """Inputs x such that x/scale hits ties, the 0.49999997 corner, clamp edges and non-finite values."""
    p = np.array([0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 3.5, 254.5, 255.5, 300, -300, np.nan, np.inf, -np.inf,
                  0.49999997, -0.49999997, 0.0, -0.0, 14.5, 15.5, 2.49999, 1e9, -1e9, 3e9, -3e9, 1e20, -1e20,
                  8388609.0, 127.5, -128.5, 3.0, 1e-40, -1e-40], dtype=np.float32)
    return (p * np.float32(scale)).astype(np.float32)


def main():
    R = O.Ref()
    isa = O.Ref.AVX512F
    assert R.supported(isa), "golden vectors are defined by the AVX-512F unit"
    rng = np.random.default_rng(20250912)
    arrays, manifest = {}, []

    def add(kind, name, meta, **arrs):
        for k, v in arrs.items():
            arrays[f"{name}.{k}"] = v
        manifest.append({"kind": kind, "name": name, **meta})

    # ---------------- quantize ----------------
    params = {O.UINT8: [(0.0078431377, 127), (0.05, -3)], O.UINT4: [(0.13333334, 7), (0.3, -2)], O.UINT2: [(0.6666667, 1), (1.0, 0)]}
    for dt_in in (O.F32, O.BF16):
        for dt_out in (O.UINT8, O.UINT4, O.UINT2):
            for rm, taus in ((O.NEAREST, [0.0]), (O.STOCHASTIC, [0.0, 0.25, 0.5, 0.999])):
                for (scale, zp) in params[dt_out]:
                    for n in LENGTHS:
                        for tau in taus:
                            if rm == O.STOCHASTIC and n not in (0, 1, 5, 17, 65, 257, 1000):
                                continue
                            x = rng.uniform(-1.2, 1.2, n).astype(np.float32)
                            if n >= 64:
                                ev = edge_values(scale)
                                pos = rng.choice(n, size=min(len(ev), n), replace=False)
                                x[pos] = ev[: len(pos)]
                            xin = x if dt_in == O.F32 else O.f32_to_bf16(x)
                            ref = R.quantize(xin, dt_in, dt_out, scale, zp, rm, tau, isa=isa)
                            uni = O.quantize(xin, dt_in, dt_out, scale, zp, rm, tau, form=O.FORM_UNIFORM)
                            name = f"q_{F32_NAMES[dt_in]}_{Q_NAMES[dt_out]}_{'st' if rm else 'nr'}_{len(manifest)}"
                            add("quantize", name,
                                dict(dt_in=dt_in, dt_out=dt_out, round_mode=rm, tau=tau, scale=scale, zp=zp, numel=n),
                                x=xin, ref=ref, uniform=uni)

    # the reference's own known answers (re-executed here through _ref)
    x = np.array([.5, 1.5, 2.5, -.5, -1.5, -2.5, 3.5, 254.5, 255.5, 300, -300, np.nan, np.inf, -np.inf], dtype=np.float32)
    ref = R.quantize(x, O.F32, O.UINT8, 1.0, 10, isa=isa)
    add("quantize", "q_known_ties", dict(dt_in=O.F32, dt_out=O.UINT8, round_mode=0, tau=0.0, scale=1.0, zp=10, numel=len(x)),
        x=x, ref=ref, uniform=O.quantize(x, O.F32, O.UINT8, 1.0, 10))
    x = np.arange(9, dtype=np.float32)
    for dt_out, zp in ((O.UINT4, 1), (O.UINT2, 0)):
        ref = R.quantize(x, O.F32, dt_out, 1.0, zp, isa=isa)
        add("quantize", f"q_known_pack_{Q_NAMES[dt_out]}", dict(dt_in=O.F32, dt_out=dt_out, round_mode=0, tau=0.0, scale=1.0, zp=zp, numel=9),
            x=x, ref=ref, uniform=O.quantize(x, O.F32, dt_out, 1.0, zp))

    # the reference's self-inconsistent corner: scalar head/tail rounds 0.49999997 to 0, the SIMD
    # body to 1; and odd integers >= 2^23, where p + 0.5 rounds up to the next even float in the SIMD body
    for dt_out in (O.UINT8, O.UINT4):
        for val, zp in ((0.49999997, 0), (-0.49999997, 5), (8388609.0, -8388600)):
            x = np.full(203, val, dtype=np.float32)
            ref = R.quantize(x, O.F32, dt_out, 1.0, zp, isa=isa)
            add("quantize", f"q_corner_{Q_NAMES[dt_out]}_{len(manifest)}",
                dict(dt_in=O.F32, dt_out=dt_out, round_mode=0, tau=0.0, scale=1.0, zp=zp, numel=203),
                x=x, ref=ref, uniform=O.quantize(x, O.F32, dt_out, 1.0, zp))

    # ---------------- dequantize ----------------
    dparams = {O.UINT8: [(0.0078431377, 127), (0.5, 3)], O.UINT4: [(0.13333334, 7), (0.3, -2)], O.UINT2: [(0.6666667, 1), (1.25, 2)]}
    for dt_q in (O.UINT8, O.UINT4, O.UINT2):
        for dt_f in (O.F32, O.BF16):
            for op in (O.SET, O.ADD):
                for (scale, zp) in dparams[dt_q]:
                    for n in LENGTHS:
                        q = rng.integers(0, 256, O.packed_numel(n, dt_q)).astype(np.uint8)
                        prev = rng.uniform(-4, 4, n).astype(np.float32)
                        prev_in = prev if dt_f == O.F32 else O.f32_to_bf16(prev)
                        ref = R.dequantize(q, dt_q, dt_f, n, scale, zp, op, isa=isa, out=prev_in.copy())
                        uni = O.dequantize(q, dt_q, dt_f, n, scale, zp, op, form=O.FORM_UNIFORM, out=prev_in.copy())
                        name = f"d_{Q_NAMES[dt_q]}_{F32_NAMES[dt_f]}_{'add' if op else 'set'}_{len(manifest)}"
                        add("dequantize", name, dict(dt_in=dt_q, dt_out=dt_f, op=op, scale=scale, zp=zp, numel=n),
                            q=q, prev=prev_in, ref=ref, uniform=uni)
    q = np.arange(70, dtype=np.uint8)
    prev = np.full(70, 100.0, dtype=np.float32)
    ref = R.dequantize(q, O.UINT8, O.F32, 70, 0.5, 3, O.ADD, isa=isa, out=prev.copy())
    add("dequantize", "d_known_add", dict(dt_in=O.UINT8, dt_out=O.F32, op=O.ADD, scale=0.5, zp=3, numel=70),
        q=q, prev=prev, ref=ref, uniform=O.dequantize(q, O.UINT8, O.F32, 70, 0.5, 3, O.ADD, out=prev.copy()))

    # ---------------- fused quantize->dequantize (reference kernels.inl:30-52, command quant_dequant) ----------------
    for dt in (O.F32, O.BF16):
        for qd in (O.UINT8, O.UINT4, O.UINT2):
            for rm, tau in ((O.NEAREST, 0.0), (O.STOCHASTIC, 0.25)):
                for op in (O.SET, O.ADD):
                    for (scale, zp) in params[qd]:
                        for n in (0, 1, 5, 17, 65, 257, 1000):
                            x = rng.uniform(-1.2, 1.2, n).astype(np.float32)
                            if n >= 64:
                                ev = edge_values(scale)
                                pos = rng.choice(n, size=min(len(ev), n), replace=False)
                                x[pos] = ev[: len(pos)]
                            xin = x if dt == O.F32 else O.f32_to_bf16(x)
                            prev = rng.uniform(-4, 4, n).astype(np.float32)
                            prev_in = prev if dt == O.F32 else O.f32_to_bf16(prev)
                            ref = R.requantize(xin, dt, qd, scale, zp, rm, tau, op, isa=isa, out=prev_in.copy())
                            name = f"r_{F32_NAMES[dt]}_{Q_NAMES[qd]}_{'st' if rm else 'nr'}_{'add' if op else 'set'}_{len(manifest)}"
                            add("requantize", name, dict(dt=dt, quant_dtype=qd, round_mode=rm, tau=tau, op=op, scale=scale, zp=zp, numel=n),
                                x=xin, prev=prev_in, ref=ref)

    # ---------------- min/max ----------------
    for n in (1, 2, 63, 64, 65, 1000, 4099):
        x = rng.normal(size=n).astype(np.float32)
        lo, hi = R.minmax(x, O.F32, isa=isa)
        xb = O.f32_to_bf16(x)
        lob, hib = R.minmax(xb, O.BF16, isa=isa)
        add("minmax", f"mm_{n}", dict(numel=n, min_f32=lo, max_f32=hi, min_bf16=lob, max_bf16=hib), x=x, xb=xb)

    # one byte blob + an index (4000 tiny zip members would cost more than the data)
    index, chunks, off = {}, [], 0
    for k, v in arrays.items():
        b = np.ascontiguousarray(v).view(np.uint8).reshape(-1)
        index[k] = [str(v.dtype), int(v.size), off]
        chunks.append(b)
        off += b.size
    blob = np.concatenate(chunks) if chunks else np.zeros(0, np.uint8)
    np.savez_compressed(OUT / "ref_vectors.npz", blob=blob)
    (OUT / "ref_vectors.json").write_text(json.dumps({"cases": manifest, "index": index}, separators=(",", ":")))
    print(f"{len(manifest)} cases, {blob.size/1e6:.2f} MB raw ->", (OUT / 'ref_vectors.npz').stat().st_size / 1e6, "MB")


if __name__ == "__main__":
    main()
