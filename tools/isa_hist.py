#!/usr/bin/env python3
"""Instruction histogram per kernel from hipcc -save-temps output (the gfx950 .s file)."""
import collections
import re
import sys

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "step_kernel"
lines = open(path).read().split("\n")
i = 0
while i < len(lines):
    m = re.match(r"^(_Z\S+):\s*; @", lines[i])
    if not m:
        i += 1
        continue
    name = m.group(1)
    j = i + 1
    while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
        j += 1
    if flt in name:
        ins = [l.strip().split()[0] for l in lines[i + 1:j] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        c = collections.Counter(ins)
        f64 = sum(v for k, v in c.items() if "f64" in k)
        imul = sum(v for k, v in c.items() if re.search(r"mul_(lo|hi)_u32|mad_u64_u32", k))
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        trans = sum(v for k, v in c.items() if re.search(r"v_(rcp|rsq|sqrt|div_scale|div_fmas|div_fixup|ldexp|frexp|trig)", k))
        short = re.sub(r"^_ZN3mxv12_GLOBAL__N_1\d+", "", name)[:40]
        print(f"{short:42s} total={len(ins)} valu={valu} f64={f64} intmul={imul} div/rcp-family={trans} "
              f"cbranch={sum(v for k, v in c.items() if k.startswith('s_cbranch'))} "
              f"gload={sum(v for k, v in c.items() if k.startswith('global_load'))} "
              f"gstore={sum(v for k, v in c.items() if k.startswith('global_store'))}")
        if len(sys.argv) > 3:
            print("    ", c.most_common(int(sys.argv[3])))
    i = j
