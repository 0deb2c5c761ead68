#!/bin/bash
# One GPU-box pass: parity suite, bench line, rocprof kernel stats, FETCH/WRITE PMC passes (separate runs).
# usage (from the repo root, through gpurun):  bash tools/gpu_round.sh <tag>     -> gpurun_out/<tag>_*
set -o pipefail
tag=${1:-r1}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/${tag}_pytest.log 2>&1 || { tail -30 $out/${tag}_pytest.log; exit 1; }
tail -2 $out/${tag}_pytest.log
timeout -k 10 400 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err || { tail -20 $out/${tag}_bench.err; exit 1; }
cat $out/${tag}_bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/${tag}_prof -o stats --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/${tag}_prof.log 2>&1 || { tail -20 $out/${tag}_prof.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $out/${tag}_pmc_f -o f --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/${tag}_pmcf.log 2>&1 || { tail -20 $out/${tag}_pmcf.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $out/${tag}_pmc_w -o w --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/${tag}_pmcw.log 2>&1 || { tail -20 $out/${tag}_pmcw.log; exit 1; }
mkdir -p $out/${tag}_pmc && cp -r $out/${tag}_pmc_f $out/${tag}_pmc_w $out/${tag}_pmc/
python tools/collect_traffic.py $out/${tag}_pmc $out/${tag}_traffic.json > /dev/null
python - <<PY
import csv, glob
f = glob.glob("$out/${tag}_prof/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
with open("$out/${tag}_kernel_stats.txt", "w") as o:
    for r in rows[:14]:
        line = "%-120s calls %6s avg_ns %12s pct %6s" % (r["Name"][:120], r["Calls"], r["AverageNs"], r["Percentage"])
        print(line); o.write(line + "\n")
PY
# MFMA-pipe occupancy (PMC pass of its own: counters only)
MC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
timeout -k 10 300 rocprofv3 --pmc $MC -d $out/${tag}_pmc_mfma_fwd -o m --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/${tag}_pmcm.log 2>&1 || { tail -20 $out/${tag}_pmcm.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc $MC -d $out/${tag}_pmc_mfma_bwd -o m --output-format csv -- python tools/bench_train.py > $out/${tag}_pmcm2.log 2>&1 || { tail -20 $out/${tag}_pmcm2.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc $MC -d $out/${tag}_pmc_mfma_dy -o m --output-format csv -- python tools/bench_dynca_train.py > $out/${tag}_pmcm3.log 2>&1 || { tail -20 $out/${tag}_pmcm3.log; exit 1; }
python tools/pmc_mfma.py $out/${tag}_pmc_mfma_fwd $out/${tag}_pmc_mfma_bwd $out/${tag}_pmc_mfma_dy > $out/${tag}_pmc_mfma.txt
# training-shaped pass (forward with history + backward), bf16 / bf16x3 forward, backward phase accounting
timeout -k 10 200 python tools/bench_train.py > $out/${tag}_train.json 2> $out/${tag}_train.err || { tail -20 $out/${tag}_train.err; exit 1; }
timeout -k 10 200 python tools/bench_train.py cfg3 >> $out/${tag}_train.json 2>> $out/${tag}_train.err || { tail -20 $out/${tag}_train.err; exit 1; }
timeout -k 10 200 python tools/bench_bf16.py > $out/${tag}_bf16.json 2> $out/${tag}_bf16.err || { tail -20 $out/${tag}_bf16.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/${tag}_trainprof -o stats --output-format csv -- python tools/bench_train.py > $out/${tag}_trainprof.log 2>&1 || { tail -20 $out/${tag}_trainprof.log; exit 1; }
python - <<PY
import csv, glob
f = glob.glob("$out/${tag}_trainprof/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
with open("$out/${tag}_train_kernel_stats.txt", "w") as o:
    for r in rows[:10]:
        line = "%-120s calls %6s avg_ns %12s pct %6s" % (r["Name"][:120], r["Calls"], r["AverageNs"], r["Percentage"])
        print(line); o.write(line + "\n")
PY
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/${tag}_dyprof -o stats --output-format csv -- python tools/bench_dynca_train.py > $out/${tag}_dyprof.log 2>&1 || { tail -20 $out/${tag}_dyprof.log; exit 1; }
python - <<PY
import csv, glob
f = glob.glob("$out/${tag}_dyprof/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
with open("$out/${tag}_dynca_train_kernel_stats.txt", "w") as o:
    for r in rows[:14]:
        line = "%-120s calls %6s avg_ns %12s pct %6s" % (r["Name"][:120], r["Calls"], r["AverageNs"], r["Percentage"])
        print(line); o.write(line + "\n")
PY
if [ -f video-stylization-with-nca_amd/libncahip_stamps.so ]; then
  timeout -k 10 200 python tools/stamp_bwd.py > $out/${tag}_bwd_phases.txt 2>&1 || { tail -20 $out/${tag}_bwd_phases.txt; exit 1; }
  cat $out/${tag}_bwd_phases.txt
fi
timeout -k 10 200 python tools/bench_dynca_train.py >> $out/${tag}_train.json 2>> $out/${tag}_train.err || { tail -20 $out/${tag}_train.err; exit 1; }
cat $out/${tag}_train.json $out/${tag}_bf16.json
