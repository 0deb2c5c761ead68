#!/bin/bash
# robustness probe for the tests whose bounds depend on near-zero ReLU gates: different host thread counts (the oracle's summation
# order) and fuzz seeds.  usage (GPU box): bash tools/flake_probe.sh
set -o pipefail
for th in 1 4 16; do
  echo "== OMP_NUM_THREADS=$th: cfg3 crops + C=20 module test"
  OMP_NUM_THREADS=$th timeout -k 10 500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_modules.py -q -m gpu -k "cfg3_forward_backward or default_arguments_c20 or g11" 2>&1 | tail -2
done
for seed in 1 31337 90210 4242; do
  echo "== NCAHIP_FUZZ_SEED=$seed (48 cases)"
  NCAHIP_FUZZ_SEED=$seed NCAHIP_FUZZ_CASES=48 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "shape_fuzz" 2>&1 | tail -2
done
for seed in 7 2024; do
  echo "== NCAHIP_FUZZ_SEED=$seed: the two forms of backward kernel A + bf16 tests"
  NCAHIP_FUZZ_SEED=$seed NCAHIP_FUZZ_CASES=24 timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py -q -m gpu 2>&1 | tail -2
done
