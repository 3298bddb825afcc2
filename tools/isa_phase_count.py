#!/usr/bin/env python
"""Instruction mix of a kernel between s_barrier's, from hipcc -save-temps assembly (the VALU count per MFMA is the
number that matters for fp32: tools/pmc_gemm.sh shows matrix and vector ALU time add up).
usage: isa_phase_count.py <file.s> <mangled-name-substring>"""
import collections
import re
import sys

a = open(sys.argv[1]).read()
m = re.search(r"^(\S*%s\S*):[^\n]*\n" % re.escape(sys.argv[2]), a, re.M)
i = m.end()
j = a.index("s_endpgm", i)
seg, cur = [], collections.Counter()
for l in a[i:j].split("\n"):
    l = l.strip()
    if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"):
        continue
    op = l.split()[0]
    if op == "s_barrier":
        seg.append(cur)
        cur = collections.Counter()
        continue
    key = ("mfma" if op.startswith("v_mfma") else "acc_mov" if op.startswith("v_accvgpr") else
           "valu_pk" if op.startswith("v_pk") else "exp" if op.startswith("v_exp") else
           "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else
           "vmem" if op.startswith(("global_", "flat_", "buffer_")) else "other")
    cur[key] += 1
seg.append(cur)
print(m.group(1)[:100])
for k, sg in enumerate(seg):
    print(f"  segment {k}: " + ", ".join(f"{n} {c}" for n, c in sorted(sg.items())))
