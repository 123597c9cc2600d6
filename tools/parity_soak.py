#!/usr/bin/env python3
"""Long randomized parity run on the GPU: HIP kernels vs the oracle, bit for bit, for a wall-clock budget.

Every iteration draws a configuration (dtype pair, rounding mode, store op, ragged size up to 2 M elements, scale over 60
decades, zero point from the int64 range or a sane one, any fp32 bit pattern as data incl. NaN/inf/denormals, or ordinary
data) and checks quantize, dequantize, the fused quantize->dequantize, the min/max scan, dynamic (params + quantize, fused and unfused)
and dequantize_sum.  Prints one JSON line: iterations, elements compared, mismatches (must be 0).

  python tools/parity_soak.py --seconds 600 [--seed 1]        (tests/ holds the short, deterministic versions)
"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "pi-quant_amd"), str(ROOT / "tests")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle as O  # noqa: E402
import piquant  # noqa: E402
import piquant.torch as pt  # noqa: E402
from helpers import gpu_dequantize, gpu_quantize, same_floats  # noqa: E402
from test_gpu_parity import _fuzz_values, gpu_quantize_dynamic, gpu_requantize  # noqa: E402


DEGENERATE_SCALES = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, -0.05, -1.0, 1e-40, -1e-40, 3e38, -3e38, 1.1754944e-38, 3.4028235e38, 1e-45, 2.0 ** -126, 2.0 ** -127,
                              8.6e37], dtype=np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=600.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    ctx, plain = piquant.Context(), piquant.Context()
    plain.set_fusion(False)
    # the ordinary checks compare with the oracle's UNIFORM form; the reference layout (the plain calls' default since round 6) has a block of its own below
    ctx.set_reference_layout(False)
    plain.set_reference_layout(False)
    t_end = time.time() + args.seconds
    it = elems = bad = 0
    fail_dir = ROOT / "gpurun_out"

    def check(ok, kind, **what):
        """counts a failed comparison and leaves what is needed to replay it: one JSON line on stderr, the arrays in gpurun_out/soak_fail_<it>_<kind>.npz"""
        nonlocal bad
        if ok:
            return
        bad += 1
        arrays = {k: v for k, v in what.items() if isinstance(v, np.ndarray)}
        scalars = {k: (v if isinstance(v, (int, float, str, bool)) else repr(v)) for k, v in what.items() if not isinstance(v, np.ndarray)}
        try:
            g, w = np.ascontiguousarray(arrays["got"]).view(np.uint8).reshape(-1), np.ascontiguousarray(arrays["want"]).view(np.uint8).reshape(-1)
            diff = np.nonzero(g != w)[0] if g.shape == w.shape else np.array([-1])
            scalars["differing_bytes"] = int(diff.size)
            scalars["first_differing_bytes"] = [int(i) for i in diff[:8]]
        except Exception as exc:   # the report must not stop the run
            scalars["diff_error"] = repr(exc)
        print(json.dumps({"soak_mismatch": kind, "iteration": it, "seed": args.seed, **scalars}), file=sys.stderr, flush=True)
        if bad <= 20:
            fail_dir.mkdir(exist_ok=True)
            np.savez_compressed(fail_dir / f"soak_fail_{it}_{kind}.npz", **arrays)
    kinds = {"quantize": 0, "dequantize": 0, "requantize": 0, "dynamic": 0, "dequantize_sum": 0, "reduce_quantize": 0, "batch": 0}
    tq = {4: torch.quint8, 3: torch.quint4x2, 2: torch.quint2x4}
    while time.time() < t_end:
        it += 1
        n = int(rng.integers(1, 2_000_000)) if it % 4 == 0 else int(rng.integers(1, 50_000))
        wild = it % 2 == 0
        x = _fuzz_values(rng, n) if wild else rng.uniform(-3, 3, n).astype(np.float32)
        scale = float(np.float32(10.0 ** rng.uniform(-30, 30))) if wild else float(np.float32(rng.uniform(0.001, 0.1)))
        if wild and it % 16 == 2:   # any float is a legal scale for the reference (src/piquant.cpp:286-295): zeros, infinities, NaN, negative, denormal, denormal reciprocal
            scale = float(rng.choice(DEGENERATE_SCALES))
        zp = int(rng.integers(-2**63, 2**63 - 1)) if it % 5 == 0 else int(rng.integers(-300, 300))
        dt_f, dt_q = int(rng.integers(0, 2)), int(rng.integers(2, 5))
        xin = x if dt_f == 0 else O.f32_to_bf16(x)
        # every third ordinary tensor carries NaNs of both kinds and signs (1 % of its elements): parameters derived from such data are
        # still defined (NaNs are skipped), so the dynamic, reduce and batch paths see them too
        nanny = (not wild) and it % 3 == 1

        def plant(a):
            idx = rng.choice(a.size, max(1, a.size // 100), replace=False)
            if a.dtype == np.float32:
                a.view(np.uint32)[idx] = rng.choice(np.array([0x7fc00000, 0xffc00000, 0x7f800001, 0xff800123, 0x7fa00000], dtype=np.uint32), idx.size)
            else:   # bf16 bit patterns
                a.view(np.uint16)[idx] = rng.choice(np.array([0x7fc0, 0xffc0, 0x7f81, 0xff83, 0x7fa0], dtype=np.uint16), idx.size)

        if nanny and n > 1:
            plant(xin)
        rm = int(rng.integers(0, 2))
        tau = float(rng.uniform(0, 1)) if rm else 0.0
        ctx.set_stochastic_threshold(tau if rm else None)
        op = int(rng.integers(0, 2))

        # every third iteration on buffers that are only element-aligned (round 3: such calls run the vector kernels -- scalar head up to the next
        # cache line of the store stream, misaligned 16-byte loads, the realigned loads of a bf16 tensor that starts on an odd element)
        off_f = int(rng.integers(0, 40)) * (4 if dt_f == 0 else 2) if it % 3 == 0 else 0
        off_q = int(rng.integers(0, 200)) if it % 3 == 0 else 0
        got = gpu_quantize(ctx, xin, dt_f, dt_q, scale, zp, rm, offset_in=off_f, offset_out=off_q)
        want = O.quantize(xin, dt_f, dt_q, scale, zp, rm, tau)
        check(np.array_equal(got, want), "quantize", n=n, dt_f=dt_f, dt_q=dt_q, scale=scale, zp=zp, rm=rm, tau=tau, wild=wild, off_f=off_f, off_q=off_q, x=xin, got=got,
              want=want)
        kinds["quantize"] += 1

        q = rng.integers(0, 256, O.packed_numel(n, dt_q)).astype(np.uint8)
        prev = (_fuzz_values(rng, n) if wild else rng.uniform(-5, 5, n).astype(np.float32))
        prev = prev if dt_f == 0 else O.f32_to_bf16(prev)
        if nanny and n > 1:
            plant(prev)
        got = gpu_dequantize(ctx, q, dt_q, dt_f, n, scale, zp, op, prev=prev, offset_in=off_q, offset_out=off_f)
        want = O.dequantize(q, dt_q, dt_f, n, scale, zp, op, out=prev.copy())
        check(same_floats(got, want), "dequantize", n=n, dt_f=dt_f, dt_q=dt_q, scale=scale, zp=zp, op=op, wild=wild, off_f=off_f, off_q=off_q, q=q, prev=prev, got=got,
              want=want)
        kinds["dequantize"] += 1

        got = gpu_requantize(ctx, xin, dt_f, dt_q, scale, zp, rm, op, prev)
        want = O.requantize(xin, dt_f, dt_q, scale, zp, rm, tau, op, out=prev.copy())
        check(same_floats(got, want), "requantize", n=n, dt_f=dt_f, dt_q=dt_q, scale=scale, zp=zp, rm=rm, tau=tau, op=op, wild=wild, x=xin, prev=prev, got=got, want=want)
        kinds["requantize"] += 1
        elems += 3 * n

        if it % 3 == 2:
            # reference-layout mode: the scalar heads / tails of a T-thread reference context (src/piquant.cpp:145-157) take the reference's scalar formulas
            # at the reference's positions; the expected bytes are the oracle's REFERENCE form, the head placed by the output pointer's alignment
            threads = int(rng.choice([1, 1, 2, 3, 7, 64, 255, 255, 1000]))   # round 6: inside the vector launch, any size, any T
            ctx.set_reference_layout(True, threads=threads)
            try:
                off = int(rng.integers(0, 16)) if (dt_f, dt_q) == (0, 4) else 0
                nbytes = O.packed_numel(n, dt_q)
                wbuf = np.zeros(nbytes + 48, dtype=np.uint8)
                base = (-wbuf.ctypes.data) % 16
                want = O.quantize(xin, dt_f, dt_q, scale, zp, rm, tau, form=O.FORM_REFERENCE, threads=threads, out=wbuf[base + off: base + off + nbytes]).copy()
                got = gpu_quantize(ctx, xin, dt_f, dt_q, scale, zp, rm, offset_out=off)
                check(np.array_equal(got, want), "reference_layout_quantize", n=n, dt_f=dt_f, dt_q=dt_q, scale=scale, zp=zp, rm=rm, tau=tau, threads=threads, off=off, x=xin,
                      got=got, want=want)
                want = O.dequantize(q, dt_q, dt_f, n, scale, zp, op, form=O.FORM_REFERENCE, threads=threads, out=prev.copy())
                got = gpu_dequantize(ctx, q, dt_q, dt_f, n, scale, zp, op, prev=prev.copy())
                check(same_floats(got, want), "reference_layout_dequantize", n=n, dt_f=dt_f, dt_q=dt_q, scale=scale, zp=zp, op=op, threads=threads, q=q, prev=prev, got=got,
                      want=want)
            finally:
                ctx.set_reference_layout(False, threads=1)
            kinds["reference_layout"] = kinds.get("reference_layout", 0) + 2
            elems += 2 * n

        # min/max scan on whatever the data is (NaNs of both kinds are skipped, infinities count); numeric comparison: -0.0 == 0.0
        xd_buf = torch.empty(xin.nbytes + off_f, dtype=torch.uint8, device="cuda")
        xd = xd_buf[off_f:]
        xd.copy_(torch.from_numpy(xin.view(np.uint8)))
        keys_d = torch.empty(2, dtype=torch.int32, device="cuda")
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.set_blocking(False)
        ctx.minmax_keys_ptr(xd.data_ptr(), piquant.DataType(dt_f), n, keys_d.data_ptr(), True)
        k = keys_d.cpu().numpy()
        got_mm, want_mm = piquant.decode_minmax_keys(int(k[0]), int(k[1])), O.minmax(xin, dt_f)
        check(float(got_mm[0]) == float(want_mm[0]) and float(got_mm[1]) == float(want_mm[1]), "minmax", n=n, dt_f=dt_f, wild=wild, got_minmax=got_mm,
              want_minmax=want_mm, x=xin)
        kinds["minmax"] = kinds.get("minmax", 0) + 1
        elems += n

        if not wild:   # data-derived parameters need a positive finite scale
            want_p = O.compute_quant_params(xin, dt_f, dt_q)
            want = O.quantize(xin, dt_f, dt_q, want_p[0], want_p[1], rm, tau)
            for c in (ctx, plain):
                c.set_stochastic_threshold(tau if rm else None)
                got, got_p = gpu_quantize_dynamic(c, xin, dt_f, dt_q, rm)
                check(got_p == want_p and np.array_equal(got, want), "dynamic_fused" if c is ctx else "dynamic_two_launches", n=n, dt_f=dt_f, dt_q=dt_q, rm=rm, tau=tau,
                      got_params=got_p, want_params=want_p, x=xin, got=got, want=want)
            kinds["dynamic"] += 2
            elems += 2 * n
            K = int(rng.integers(1, 5))
            if K:
                xs = [rng.uniform(-2 - i, 2 + i, n).astype(np.float32) for i in range(K)]
                qs, recs = [], []
                want_acc = prev.copy()
                for xk in xs:
                    qq, rec = pt.quantize_dynamic(torch.from_numpy(xk).cuda(), dtype=tq[dt_q])
                    qs.append(pt.packed_bytes(qq))
                    recs.append(rec)
                    s_k, z_k = pt.params_to_host(rec)
                    want_acc = O.dequantize(qs[-1].cpu().numpy(), dt_q, dt_f, n, s_k, z_k, 1, out=want_acc)
                fdt = torch.float32 if dt_f == 0 else torch.bfloat16
                acc = torch.from_numpy(prev).cuda() if dt_f == 0 else torch.from_numpy(prev.view(np.int16)).cuda().view(torch.bfloat16)
                pt.dequantize_sum(qs, recs, dtype=fdt, reduce_op="add", out=acc, quant_dtype=tq[dt_q], shape=(n,))
                got_acc = acc.cpu().numpy() if dt_f == 0 else acc.view(torch.int16).cpu().numpy().view(np.uint16)
                check(same_floats(got_acc, want_acc), "dequantize_sum", n=n, dt_f=dt_f, dt_q=dt_q, K=K, prev=prev, got=got_acc, want=want_acc)
                kinds["dequantize_sum"] += 1
                elems += K * n
                # the same sum re-quantized in one call (parameters from the sum)
                acc2 = torch.from_numpy(prev).cuda() if dt_f == 0 else torch.from_numpy(prev.view(np.int16)).cuda().view(torch.bfloat16)
                ctx.set_stochastic_threshold(tau if rm else None)
                rq, rrec = pt.reduce_quantize_dynamic(acc2, qs, recs, dtype=tq[dt_q], round_mode="stochastic" if rm else "nearest", ctx=ctx)
                wp = O.compute_quant_params(want_acc, dt_f, dt_q)
                if wp[0] > 0 and np.isfinite(wp[0]):
                    got_rq, want_rq = pt.packed_bytes(rq).cpu().numpy(), O.quantize(want_acc, dt_f, dt_q, wp[0], wp[1], rm, tau)
                    check(pt.params_to_host(rrec) == wp and np.array_equal(got_rq, want_rq), "reduce_quantize", n=n, dt_f=dt_f, dt_q=dt_q, K=K, rm=rm, tau=tau,
                          got_params=pt.params_to_host(rrec), want_params=wp, acc=want_acc, got=got_rq, want=want_rq)
                    kinds["reduce_quantize"] += 1
                    elems += n
        if not wild and it % 7 == 0:   # several tensors per launch, then back
            sizes = [int(rng.integers(1, 400_000)) for _ in range(int(rng.integers(2, 20)))]
            parts = [rng.uniform(-1 - i, 1.5 + i, m).astype(np.float32) for i, m in enumerate(sizes)]
            pin = [a if dt_f == 0 else O.f32_to_bf16(a) for a in parts]
            fdt = torch.float32 if dt_f == 0 else torch.bfloat16
            pd = [torch.from_numpy(a).cuda() if dt_f == 0 else torch.from_numpy(a.view(np.int16)).cuda().view(torch.bfloat16) for a in pin]
            qs_b, recs_b = pt.quantize_dynamic_batch(pd, dtype=tq[dt_q], ctx=ctx)
            outs_b = pt.dequantize_dynamic_batch(qs_b, recs_b, dtype=fdt, ctx=ctx)
            for a, qq, rr, oo in zip(pin, qs_b, recs_b, outs_b):
                wp = O.compute_quant_params(a, dt_f, dt_q)
                wq = O.quantize(a, dt_f, dt_q, wp[0], wp[1])
                go = oo.cpu().numpy() if dt_f == 0 else oo.view(torch.int16).cpu().numpy().view(np.uint16)
                gq = pt.packed_bytes(qq).cpu().numpy()
                check(pt.params_to_host(rr) == wp and np.array_equal(gq, wq) and same_floats(go, O.dequantize(wq, dt_q, dt_f, a.size, wp[0], wp[1])), "batch",
                      n=int(a.size), sizes=sizes, dt_f=dt_f, dt_q=dt_q, got_params=pt.params_to_host(rr), want_params=wp, x=a, got=gq, want=wq, got_dequantized=go)
                elems += 2 * a.size
            kinds["batch"] += len(sizes)
    print(json.dumps({"seconds": args.seconds, "seed": args.seed, "iterations": it, "checks": kinds, "elements_compared": elems, "mismatching_checks": bad,
                      "device": torch.cuda.get_device_name(0)}))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
