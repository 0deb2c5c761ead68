#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in a hipcc -S listing: tools/loop_mix.py file.s <kernel-substr> [min].
Prints blocks (label .. next label) with their instruction counts by class, to find where non-MFMA issue slots go."""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split("\n")
sub = sys.argv[2]
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and sub in l)
end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or lines[i].startswith(".Lfunc_end"))
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_pk_"): return "vpk"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_read") : return "ldsr"
    if op.startswith("ds_"): return "ldsw"
    if op.startswith("global_load") or op.startswith("buffer_load"): return "gld"
    if op.startswith("global_store") or op.startswith("buffer_store"): return "gst"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "br"
    if op.startswith("s_"): return "salu"
    if op.startswith("scratch"): return "scr"
    return "other"
blocks, cur, name = [], Counter(), "entry"
for l in lines[start + 1:end]:
    t = l.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m:
        blocks.append((name, cur)); cur, name = Counter(), m.group(1); continue
    if not t or t.startswith(";") or t.startswith("."): continue
    op = t.split()[0]
    cur[cls(op)] += 1
    if cls(op) == "br": cur["->" + t.split()[-1]] += 0
blocks.append((name, cur))
tot = Counter()
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 30
for n, c in blocks:
    k = sum(v for kk, v in c.items() if not kk.startswith("->"))
    if k >= mn:
        tg = [kk[2:] for kk in c if kk.startswith("->")]
        print("%-10s n=%4d " % (n, k) + " ".join("%s=%d" % (kk, v) for kk, v in sorted(c.items()) if not kk.startswith("->")) + ("  -> " + ",".join(tg) if tg else ""))
    tot.update({kk: v for kk, v in c.items() if not kk.startswith("->")})
print("TOTAL", dict(tot))
