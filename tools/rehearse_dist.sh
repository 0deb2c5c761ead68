#!/bin/bash
# Rehearsal of bench.py's N > 1 protocol on a ONE-GPU box: N ranks share cuda:0, gloo instead of RCCL (the driver's real run is one rank
# per GPU over RCCL).  Checks the rendezvous, the barriers, the max over ranks, rank 0's single JSON line and the trainers' all-reduce.
# usage: bash tools/rehearse_dist.sh [N=2]
set -o pipefail
n=${1:-2}
out=gpurun_out; mkdir -p $out
NCAHIP_BENCH_SHARE_GPU=1 NCAHIP_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n \
  --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $n --steps 3 --warmup 1 --train-iters 1 > $out/rehearse_dist.json 2> $out/rehearse_dist.err \
  || { tail -30 $out/rehearse_dist.err; exit 1; }
python - <<'PY'
import json
lines = [l for l in open("gpurun_out/rehearse_dist.json") if l.startswith("{")]
assert len(lines) == 1, f"expected ONE JSON line, got {len(lines)}"
d = json.loads(lines[0])
print({k: d[k] for k in ("value", "n_gpus", "ms_per_step", "scaling")}, d["config"]["parallelism"])
t = d.get("train") or {}
print({k: round(v["ms_per_iteration"], 1) for k, v in t.items() if isinstance(v, dict) and "ms_per_iteration" in v}, "allreduce_floats", t.get("allreduce_floats"))
PY
