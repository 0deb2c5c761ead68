"""A/B: ConditionedNCA forward-with-history + backward with and without the alive-mask machinery (alive_ch = -1 skips the
pending-mask resolution, both max-poolings and the life gate): an upper bound on what a stored life-mask history could save."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import tools.bench_paths as bp
from tools.bench_paths import ops, timed, cond_case

for dtype in (torch.float32, torch.bfloat16):
    for ach in (3, -1):
        x, goal, cot, w = cond_case(8, dtype=dtype)
        T, box = 16, {}
        def fwd():
            box["h"] = ops.cond_grow(x, T, goal, None, w, ach, seed=1, step0=0, keep_history=True)
        def bwd():
            _, states, pre = box.pop("h")
            ops.cond_grow_backward(states, pre, goal, None, w, cot, T, ach, seed=1, step0=0)
        (tf, tb), _ = timed([fwd, bwd], iters=10)
        print(dtype, "alive_ch", ach, "fwd us/step %.1f  bwd us/step %.1f" % (tf / T * 1e3, tb / T * 1e3), flush=True)
