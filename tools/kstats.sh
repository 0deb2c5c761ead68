#!/bin/bash
# rocprofv3 kernel stats of selected bench_paths legs.  usage: bash tools/kstats.sh <tag> "<legs>" [rows]   -> gpurun_out/<tag>_kernel_stats.txt
set -o pipefail
tag=$1; legs=$2; rows=${3:-14}
out=gpurun_out; mkdir -p $out; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/${tag}_prof -o stats --output-format csv -- python tools/bench_paths.py $legs > $out/${tag}_prof.log 2>&1 || { tail -20 $out/${tag}_prof.log; exit 1; }
python - "$out/${tag}_prof" "$out/${tag}_kernel_stats.txt" "$rows" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
with open(sys.argv[2], "w") as o:
    for r in rows[:int(sys.argv[3])]:
        line = "%-120s calls %6s avg_ns %12s pct %6s" % (r["Name"][:120], r["Calls"], r["AverageNs"], r["Percentage"])
        print(line); o.write(line + "\n")
PY
rm -rf $out/${tag}_prof
