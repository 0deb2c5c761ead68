#!/bin/bash
# quick GPU pass for one change: selected tests + selected bench_paths legs.  usage: bash tools/gpu_quick.sh <tag> "<pytest -k expr>" "<bench_paths names>"
set -o pipefail
tag=${1:-q}; kexpr=${2:-}; legs=${3:-}
out=gpurun_out; mkdir -p $out
if [ -n "$kexpr" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$kexpr" > $out/${tag}_pytest.log 2>&1 || { tail -40 $out/${tag}_pytest.log; exit 1; }
  tail -3 $out/${tag}_pytest.log
fi
if [ -n "$legs" ]; then
  timeout -k 10 500 python tools/bench_paths.py $legs > $out/${tag}_paths.jsonl 2> $out/${tag}_paths.err || { tail -20 $out/${tag}_paths.err; exit 1; }
  cat $out/${tag}_paths.jsonl
fi
