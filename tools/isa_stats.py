#!/usr/bin/env python3
"""Per-kernel instruction mix of a hipcc -S listing (tools for kernel tuning; not shipped path)."""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w*kernel\w*:\s", l)]
for i, name in starts:
    j = i
    while j < len(lines) and ".end_amdhsa_kernel" not in lines[j]:
        j += 1
    body = "\n".join(lines[i:j])
    c = lambda pat: len(re.findall(pat, body))
    m = re.search(r"(\d+)(\w+kernel)", name)
    print(name[:110])
    print("   mfma", c(r"\bv_mfma"), "| ds_read b32", c(r"ds_read_b32"), "2b32", c(r"ds_read2_b32"), "2st64", c(r"ds_read2st64_b32"),
          "b64", c(r"ds_read_b64"), "b128", c(r"ds_read_b128"), "| ds_write", c(r"ds_write"), "| gload", c(r"global_load"),
          "gstore", c(r"global_store"), "| fma", c(r"\bv_fma_f32|\bv_fmac_f32"), "pk", c(r"\bv_pk_"), "| waitcnt", c(r"s_waitcnt"),
          "barrier", c(r"s_barrier"), "scratch", c(r"scratch_"), "| lines", j - i)
