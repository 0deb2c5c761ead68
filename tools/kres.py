#!/usr/bin/env python3
"""Register / LDS / spill summary per kernel of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage, gfx950).
usage: python tools/kres.py video-stylization-with-nca_amd/csrc/nca_cond_bwd_fm.hip [name-filter] [-- extra hipcc flags]"""
import re
import subprocess
import sys

args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--")
    args, extra = args[:i], args[i + 1:]
src = args[0]
flt = args[1] if len(args) > 1 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0].replace("void ", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z /\[\]]+): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
    if "error" in line:
        print(line)
for k, v in rows.items():
    if flt in k:
        print(f"{k:90s} vgpr {v.get('VGPRs', -1):4d} agpr {v.get('AGPRs', -1):4d} scratch {v.get('ScratchSize [bytes/lane]', -1):5d} "
              f"sspill {v.get('SGPRs Spill', -1):4d} vspill {v.get('VGPRs Spill', -1):4d} occ {v.get('Occupancy [waves/SIMD]', -1)} lds {v.get('LDS Size [bytes/block]', -1)}")
