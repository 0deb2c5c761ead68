#!/bin/bash
# kernel resource usage of one source file, one line per kernel:  tools/kres.sh csrc/nca_cond_bwd.hip [name filter]
cd "$(dirname "$0")/../video-stylization-with-nca_amd"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/kres.o 2>&1 | python3 -c '
import re, sys, subprocess
cur = {}
rows = []
for l in sys.stdin:
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    for k in ("VGPRs", "AGPRs", "SGPRs", "ScratchSize \[bytes/lane\]", "Occupancy \[waves/SIMD\]", "SGPRs Spill", "VGPRs Spill", "LDS Size \[bytes/block\]"):
        m = re.search(r"remark:\s+" + k + r": (\d+)", l)
        if m and cur: cur[k.split(" [")[0].replace("\\","")] = m.group(1)
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for r in rows:
    n = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    n = n.replace("(anonymous namespace)::", "")
    if flt in n:
        print("%-110s v%s a%s s%s scr%s occ%s spillS%s spillV%s" % (n[:110], r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs"), r.get("ScratchSize"), r.get("Occupancy"), r.get("SGPRs Spill"), r.get("VGPRs Spill")))
' "$2"
