#!/bin/bash
# A/B of two builds of the library in ONE gpurun call: alternating runs of the headline bench (no extras).
# usage: bash tools/ab_bench.sh <other .so> [rounds]
other=$1; n=${2:-3}
cur=video-stylization-with-nca_amd/libncahip.so
for i in $(seq $n); do
  for lib in $cur $other; do
    NCAHIP_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-extras --train-iters 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['value']/1e9,4), round(d['roofline']['launch_ms']*1e3,2), round(d['roofline_stencil']['frac'],3))"
  done
done
