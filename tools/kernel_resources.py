#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel in a gfx950 code object (llvm-readelf --notes output on stdin or a file).

  hipcc --offload-arch=gfx950 ... --cuda-device-only -c pi-quant_amd/csrc/kernels.hip -o /tmp/k_dev.o
  clang-offload-bundler --unbundle --type=o --input=/tmp/k_dev.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/k.co
  llvm-readelf --notes /tmp/k.co | python tools/kernel_resources.py [substring]
"""
import re
import sys

txt = sys.stdin.read()
flt = sys.argv[1] if len(sys.argv) > 1 else ""
rows = []
for e in re.split(r"\n\s+- \.agpr_count", txt)[1:]:
    g = lambda k, d=0: int(m.group(1)) if (m := re.search(rf"\.{k}:\s+(\d+)", e)) else d
    name = re.search(r"\.name:\s+(\S+)", e).group(1)
    rows.append((name, g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"), g("vgpr_spill_count"), g("sgpr_spill_count")))
sel = [r for r in rows if flt in r[0]]
print(f"{len(rows)} kernels, {len(sel)} selected; max vgpr {max(r[1] for r in sel)}, max sgpr {max(r[2] for r in sel)}, "
      f"with scratch {sum(1 for r in sel if r[3] > 0)}, with vgpr spills {sum(1 for r in sel if r[5] > 0)}")
for r in sorted(sel, key=lambda r: -r[1])[:12]:
    print(f"vgpr {r[1]:4d} sgpr {r[2]:4d} scratch {r[3]:5d} lds {r[4]:7d} vspill {r[5]:3d} sspill {r[6]:3d}  {r[0][:140]}")
