#!/usr/bin/env python3
"""Prints the rows of DESIGN.md section 4's fixed-cost table from profiles/r02_fixed_cost_fit.json (and, with --write, replaces them in DESIGN.md)."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
f = json.loads((ROOT / "profiles" / "r02_fixed_cost_fit.json").read_text())["kernels"]


def row(k):
    v = f[k]
    return v["t0_us"], v["BW_GB/s"], v["frac_at_27264000_if_t0_were_zero"], v["frac_at_27264000_measured"], v["numel_for_70_percent"]


def bw(x):
    x = int(round(x))
    return f"{x // 1000} {x % 1000:03d}"


def m(n):
    return f"{n / 1e6:.1f} M"


t = []
a = row("quantize_f32_u8_nearest")
t.append(f"| quantize fp32→uint8 nearest | {a[0]:.2f} | {bw(a[1])} | {a[2] * 100:.0f} % | {a[3] * 100:.0f} % | {m(a[4])} elements |")
a = row("quantize_f32_u8_stochastic")
t.append(f"| quantize fp32→uint8 stochastic | {a[0]:.2f} | {bw(a[1])} | {a[2] * 100:.0f} % | {a[3] * 100:.0f} % | {m(a[4])} |")
a = row("quantize_bf16_u4_nearest")
t.append(f"| quantize bf16→uint4 | {a[0]:.2f} | {bw(a[1])} | {a[2] * 100:.0f} % | {a[3] * 100:.0f} % | **{m(a[4])}** |")
a = row("dequantize_u4_bf16_set")
t.append(f"| dequantize uint4→bf16 SET (non-temporal stores) | {a[0]:.2f} | {bw(a[1])} | {a[2] * 100:.0f} % | {a[3] * 100:.0f} % | {m(a[4])} |")
a, b = row("dequantize_u8_f32_set"), row("dequantize_u8_f32_add")
t.append(f"| dequantize uint8→fp32 SET / ADD | {a[0]:.2f} / {b[0]:.2f} | {bw(a[1])} / {bw(b[1])} | {a[2] * 100:.0f} % / {b[2] * 100:.0f} % | {a[3] * 100:.0f} % / {b[3] * 100:.0f} % | "
         f"{m(a[4])} / {m(b[4])} |")
a = row("minmax_f32")
t.append(f"| min/max fp32 | {a[0]:.2f} | {bw(a[1])} | {a[2] * 100:.0f} % | {a[3] * 100:.0f} % | {m(a[4])} |")
a = row("quantize_dynamic_f32_u8_fused")
t.append(f"| params + quantize in one launch | {a[0]:.2f} | {bw(a[1])} | {a[2] * 100:.0f} % | {a[3] * 100:.0f} % | (beyond what stays on chip) |")
print("\n".join(t))
if "--write" in sys.argv:
    p = ROOT / "DESIGN.md"
    s = p.read_text()
    i = s.index("| quantize fp32→uint8 nearest | ")
    j = s.index("\n", s.index("| params + quantize in one launch | ", i))
    p.write_text(s[:i] + "\n".join(t) + s[j:])
