#!/bin/bash
# One GPU-box pass: parity suite, bench line, rocprof kernel stats, FETCH/WRITE PMC passes (separate runs), path timings.
# usage (from the repo root, through gpurun):  bash tools/gpu_round.sh <tag> [notest]   -> gpurun_out/<tag>_*
set -o pipefail
tag=${1:-r2}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
stats() {  # <prof dir> <out txt> <rows>
python - "$1" "$2" "$3" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
with open(sys.argv[2], "w") as o:
    for r in rows[:int(sys.argv[3])]:
        line = "%-120s calls %6s avg_ns %12s pct %6s" % (r["Name"][:120], r["Calls"], r["AverageNs"], r["Percentage"])
        print(line); o.write(line + "\n")
PY
}
if [ "$2" != "notest" ]; then
  timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $out/${tag}_pytest.log 2>&1 || { tail -30 $out/${tag}_pytest.log; exit 1; }
  tail -2 $out/${tag}_pytest.log
fi
timeout -k 10 400 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err || { tail -20 $out/${tag}_bench.err; exit 1; }
cat $out/${tag}_bench.json
P="python bench.py --no-cpu-baseline --no-extras"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/${tag}_prof -o stats --output-format csv -- $P --steps 20 --warmup 3 > $out/${tag}_prof.log 2>&1 || { tail -20 $out/${tag}_prof.log; exit 1; }
stats $out/${tag}_prof $out/${tag}_kernel_stats.txt 8
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $out/${tag}_pmc/f -o f --output-format csv -- $P --steps 1 --warmup 0 > $out/${tag}_pmcf.log 2>&1 || { tail -20 $out/${tag}_pmcf.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $out/${tag}_pmc/w -o w --output-format csv -- $P --steps 1 --warmup 0 > $out/${tag}_pmcw.log 2>&1 || { tail -20 $out/${tag}_pmcw.log; exit 1; }
python tools/collect_traffic.py $out/${tag}_pmc $out/${tag}_traffic.json > /dev/null
# MFMA-pipe occupancy (PMC pass of its own: counters only)
MC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
timeout -k 10 300 rocprofv3 --pmc $MC -d $out/${tag}_pmc_mfma_fwd -o m --output-format csv -- $P --steps 1 --warmup 0 > $out/${tag}_pmcm.log 2>&1 || { tail -20 $out/${tag}_pmcm.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc $MC -d $out/${tag}_pmc_mfma_bwd -o m --output-format csv -- python tools/bench_paths.py cond_train > $out/${tag}_pmcm2.log 2>&1 || { tail -20 $out/${tag}_pmcm2.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc $MC -d $out/${tag}_pmc_mfma_dy -o m --output-format csv -- python tools/bench_paths.py dynca_train > $out/${tag}_pmcm3.log 2>&1 || { tail -20 $out/${tag}_pmcm3.log; exit 1; }
python tools/pmc_mfma.py $out/${tag}_pmc_mfma_fwd $out/${tag}_pmc_mfma_bwd $out/${tag}_pmc_mfma_dy > $out/${tag}_pmc_mfma.txt
# every other path: median-of-10 timings, then kernel stats of the two training-shaped passes
timeout -k 10 500 python tools/bench_paths.py > $out/${tag}_paths.jsonl 2> $out/${tag}_paths.err || { tail -20 $out/${tag}_paths.err; exit 1; }
cat $out/${tag}_paths.jsonl
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/${tag}_trainprof -o stats --output-format csv -- python tools/bench_paths.py cond_train > $out/${tag}_trainprof.log 2>&1 || { tail -20 $out/${tag}_trainprof.log; exit 1; }
stats $out/${tag}_trainprof $out/${tag}_train_kernel_stats.txt 10
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/${tag}_c20prof -o stats --output-format csv -- python tools/bench_paths.py cond_c20 > $out/${tag}_c20prof.log 2>&1 || { tail -20 $out/${tag}_c20prof.log; exit 1; }
stats $out/${tag}_c20prof $out/${tag}_c20_kernel_stats.txt 10
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/${tag}_dyprof -o stats --output-format csv -- python tools/bench_paths.py dynca_train > $out/${tag}_dyprof.log 2>&1 || { tail -20 $out/${tag}_dyprof.log; exit 1; }
stats $out/${tag}_dyprof $out/${tag}_dynca_train_kernel_stats.txt 14
# HBM traffic of the backward kernels (separate FETCH / WRITE passes per leg) -> <tag>_bwd_hbm.json / <tag>_dynca_bwd_hbm.json
for leg in cond_train dynca_train; do
  name=bwd_hbm; [ $leg = dynca_train ] && name=dynca_bwd_hbm
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $out/${tag}_pmc_${leg}/f -o f --output-format csv -- python tools/bench_paths.py $leg > $out/${tag}_pmc_${leg}_f.log 2>&1 || { tail -20 $out/${tag}_pmc_${leg}_f.log; exit 1; }
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $out/${tag}_pmc_${leg}/w -o w --output-format csv -- python tools/bench_paths.py $leg > $out/${tag}_pmc_${leg}_w.log 2>&1 || { tail -20 $out/${tag}_pmc_${leg}_w.log; exit 1; }
  python tools/collect_traffic.py $out/${tag}_pmc_${leg} $out/${tag}_${name}.json > /dev/null
done
if [ -f video-stylization-with-nca_amd/libncahip_stamps.so ]; then   # diagnostic build (make stamps): cycles per phase of backward kernel A
  timeout -k 10 200 python tools/stamp_bwd.py > $out/${tag}_bwd_phases.txt 2>&1 || { tail -20 $out/${tag}_bwd_phases.txt; exit 1; }
  timeout -k 10 200 python tools/stamp_bwd.py bf16 > $out/${tag}_bwd_phases_bf16.txt 2>&1 || { tail -20 $out/${tag}_bwd_phases_bf16.txt; exit 1; }
  cat $out/${tag}_bwd_phases_bf16.txt
fi
