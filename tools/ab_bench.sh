#!/bin/bash
# A/B on the GPU box: bench.py (no CPU baseline) under rocprof kernel stats for the current build; prints cadence + kernel avg
set -o pipefail
export TMPDIR=/tmp
tag=${1:-ab}
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -5 gpurun_out/${tag}_bench.err; exit 1; }
python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_bench.json"))
print("${tag}: value %.3f G/s  ms_per_step %.3f  launch_ms %.4f  frac %.3f" % (d["value"] / 1e9, d["ms_per_step"], d["roofline"]["launch_ms"], d["roofline"]["frac"]))
PY
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_prof -o stats --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_prof.log 2>&1 || { tail -5 gpurun_out/${tag}_prof.log; exit 1; }
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/${tag}_prof/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:2]:
    print("   %-70s calls %5s avg_ns %10s" % (r["Name"][:70], r["Calls"], r["AverageNs"]))
PY
