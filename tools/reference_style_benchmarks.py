#!/usr/bin/env python3
"""The reference's two Python benchmark scripts, re-stated for ROCm device tensors.

  python/benchmark/benchmark.py      torch.quantize_per_tensor vs piquant.torch.quantize, NUMEL = 1e6, 1000 runs, total
                                     wall seconds per quantized dtype (the bars of python/quant_benchmark.png)
  python/benchmark/throughput_avg.py GiB/s of bf16 <-> quint4x2 / quint2x4 over a large tensor, 10 iterations

Same measurement conventions (wall clock around the Python-level call, result allocation included; here with a device
synchronisation inside the timed region so the GPU work is counted).  torch.quantize_per_tensor runs on the same GPU for
quint8 (PyTorch-ROCm implements it); for the packed dtypes PyTorch has no device kernel, so the torch column is measured
on the host CPU like the reference does.  One JSON document is printed; --plot FILE.png also draws the reference's grouped bar
chart (python/benchmark/benchmark.py:60-72: seconds per NUM_RUNS runs, torch vs piquant per quantized dtype) when matplotlib
can be imported.  Usage: python tools/reference_style_benchmarks.py [--gib 8] [--runs 1000] [--plot profiles/rNN_quant_benchmark.png] > profiles/rNN_reference_style.json
"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "pi-quant_amd"))

import torch  # noqa: E402

import piquant  # noqa: E402

NUM_RUNS = 1000
NUMEL = 1_000_000


def torch_vs_piquant():
    rows = []
    for qdt in (torch.quint8, torch.quint4x2, torch.quint2x4):
        x = torch.rand(NUMEL, dtype=torch.float32, device="cuda")
        scale, zp = piquant.torch.compute_quant_params(x, dtype=qdt)
        on_gpu = qdt == torch.quint8
        xt = x if on_gpu else x.cpu()
        torch.quantize_per_tensor(xt, scale=scale, zero_point=zp, dtype=qdt)
        piquant.torch.quantize(x, scale=scale, zero_point=zp, dtype=qdt)
        torch.cuda.synchronize()

        def thousand(f):
            t0 = time.perf_counter()
            for _ in range(NUM_RUNS):
                f()
            torch.cuda.synchronize()
            return time.perf_counter() - t0

        # the reference script times ONE pass of 1000 runs each; at 5 us a call a pass is 5 ms and the order of the two decides the answer, so five
        # alternating passes are timed here, the best of each reported, and all of them kept
        passes = [(thousand(lambda: torch.quantize_per_tensor(xt, scale=scale, zero_point=zp, dtype=qdt)),
                   thousand(lambda: piquant.torch.quantize(x, scale=scale, zero_point=zp, dtype=qdt))) for _ in range(5 if on_gpu else 1)]
        t_torch, t_pi = min(p[0] for p in passes), min(p[1] for p in passes)
        # same check as the reference script: dequantized results agree within 1e-1
        dq_t = torch.quantize_per_tensor(xt, scale=scale, zero_point=zp, dtype=qdt).dequantize().cpu()
        dq_p = piquant.torch.dequantize(piquant.torch.quantize(x, scale=scale, zero_point=zp, dtype=qdt), scale=scale, zero_point=zp,
                                        dtype=torch.float32).cpu()
        rows.append({"dtype": str(qdt).replace("torch.", ""), f"torch_s_per_{NUM_RUNS}": round(t_torch, 6), f"piquant_s_per_{NUM_RUNS}": round(t_pi, 6),
                     "passes_torch_piquant": [[round(a, 6), round(b, 6)] for a, b in passes],
                     "torch_device": "cuda" if on_gpu else "cpu (no device kernel in PyTorch for this dtype)",
                     # the reference script PRINTS the elements that differ (it does not assert): exact ties round half-to-even in torch and
                     # half-away-from-zero here (as in the reference), one quantization step apart -- a third of the range for quint2x4
                     "results_allclose_1e-1": bool(torch.allclose(dq_t, dq_p, atol=1e-1)),
                     "elements_beyond_1e-1": int((~torch.isclose(dq_t, dq_p, atol=1e-1)).sum())})
    return rows


def readme_headline():
    """The chart of the reference's README (README.md:72-97, media/bench*.png): total seconds for 1000 runs of fp32 -> quint8 at
    numel 27 264 000, pi-quant vs torch.quantize_per_tensor -- here both on the same MI355X."""
    n = 27_264_000
    x = torch.rand(n, dtype=torch.float32, device="cuda") * 2 - 1
    scale, zp = piquant.torch.compute_quant_params(x, dtype=torch.quint8)
    rows = {}
    for name, fn in (("torch.quantize_per_tensor (device)", lambda: torch.quantize_per_tensor(x, scale=scale, zero_point=zp, dtype=torch.quint8)),
                     ("piquant.torch.quantize (device)", lambda: piquant.torch.quantize(x, scale=scale, zero_point=zp, dtype=torch.quint8))):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(NUM_RUNS):
            fn()
        torch.cuda.synchronize()
        rows[name] = round(time.perf_counter() - t0, 5)
    a = torch.quantize_per_tensor(x, scale=scale, zero_point=zp, dtype=torch.quint8).int_repr()
    b = piquant.torch.packed_bytes(piquant.torch.quantize(x, scale=scale, zero_point=zp, dtype=torch.quint8))
    rows["elements_differing_from_torch"] = int((a.view(-1) != b).sum())   # torch rounds x/scale half-to-even, pi-quant x*(1/scale) half-away
    rows["reference_published_s_per_1000_runs"] = {"EPYC 9654 (AVX-512F)": 1.7, "EPYC 7742 (AVX2)": 2.8, "Apple M3 Pro (NEON)": 1.4,
                                              "torch builtin on EPYC 9654": 11.0}
    return rows


def throughput(total_gib: float):
    rows = []
    for dq_type, q_type in ((torch.bfloat16, torch.quint4x2), (torch.bfloat16, torch.quint2x4), (torch.float32, torch.quint8)):
        bpe = torch.tensor([], dtype=dq_type).element_size()
        n = int(total_gib * (1 << 30)) // bpe
        x = torch.rand(n, dtype=torch.float32, device="cuda").to(dq_type) if dq_type != torch.float32 else torch.rand(n, device="cuda")
        scale, zp = piquant.torch.compute_quant_params(x, dtype=q_type)
        q = piquant.torch.quantize(x, scale=scale, zero_point=zp, dtype=q_type)
        piquant.torch.dequantize(q, scale=scale, zero_point=zp, dtype=dq_type)
        torch.cuda.synchronize()
        qs, dqs = [], []
        for _ in range(10):
            t0 = time.perf_counter()
            q = piquant.torch.quantize(x, scale=scale, zero_point=zp, dtype=q_type)
            torch.cuda.synchronize()
            qs.append(total_gib / (time.perf_counter() - t0))
            t0 = time.perf_counter()
            y = piquant.torch.dequantize(q, scale=scale, zero_point=zp, dtype=dq_type)
            torch.cuda.synchronize()
            dqs.append(total_gib / (time.perf_counter() - t0))      # GiB of float data produced (the reference divides the packed size)
            del y
        rows.append({"pair": f"{str(dq_type).replace('torch.', '')} <-> {str(q_type).replace('torch.', '')}", "numel": n,
                     "quantize_GiB/s_of_float_input": round(sum(qs) / len(qs), 1), "dequantize_GiB/s_of_float_output": round(sum(dqs) / len(dqs), 1)})
        del x, q
    return rows


def plot(rows, path):
    """The reference's chart (python/benchmark/benchmark.py:60-72): one pair of bars per quantized dtype.  Returns False without
    matplotlib."""
    try:
        import matplotlib

        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except ImportError:
        return False
    labels = [r["dtype"] for r in rows]
    pos = list(range(len(labels)))
    width = 0.35
    fig, ax = plt.subplots(figsize=(7, 4))
    ax.bar([p_ - width / 2 for p_ in pos], [r[f"torch_s_per_{NUM_RUNS}"] for r in rows], width, label="torch.quantize_per_tensor")
    ax.bar([p_ + width / 2 for p_ in pos], [r[f"piquant_s_per_{NUM_RUNS}"] for r in rows], width, label="piquant.torch.quantize (MI355X)")
    ax.set_ylabel(f"seconds per {NUM_RUNS} runs")
    ax.set_xlabel("Quantized dtype")
    ax.set_title(f"Quantization benchmark (numel={NUMEL}, {NUM_RUNS} runs)")
    ax.set_xticks(pos)
    ax.set_xticklabels(labels)
    ax.set_yscale("log")
    ax.legend()
    fig.tight_layout()
    fig.savefig(path, dpi=120)
    return True


def main():
    global NUM_RUNS
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=8.0, help="size of the float tensor for the throughput part (reference: 32)")
    ap.add_argument("--runs", type=int, default=NUM_RUNS, help="runs per timing (reference: 1000)")
    ap.add_argument("--plot", default=None, help="write the reference's bar chart to this PNG")
    args = ap.parse_args()
    NUM_RUNS = args.runs
    bars = torch_vs_piquant()
    out = {"device": torch.cuda.get_device_name(0), "runs": NUM_RUNS,
           f"README headline (numel=27264000, seconds per {NUM_RUNS} runs)": readme_headline(),
           f"benchmark_py (NUMEL=1e6, {NUM_RUNS} runs)": bars,
           f"throughput_avg_py ({args.gib} GiB float tensor, 10 iterations, allocation + sync inside the timed call)": throughput(args.gib)}
    if args.plot:
        out["plot"] = args.plot if plot(bars, args.plot) else "matplotlib is not importable: no plot"
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
