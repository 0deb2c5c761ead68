#!/usr/bin/env python3
"""Timing of the training-shaped pass (BASELINE configs[2] shape family): grow with history + backward."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-stylization-with-nca_amd")]
import torch
import bench
from ncahip import ops

def run(B, T, H=256, W=256, C=16, iters=3, warm=2):
    dev = "cuda"
    gen = torch.Generator().manual_seed(0)
    prm = bench.make_weights(gen)
    x = torch.rand(B, C, H, W, generator=gen).to(dev)
    goal = (torch.randn(B, 12, H, W, generator=gen) * 0.5).to(dev)
    cot = torch.randn(B, C, H, W, generator=gen).to(dev)
    w = ops.CondWeights(prm["perception_net.weight"], prm["update_net.out.0.weight"], prm["update_net.out.0.bias"],
                        prm["update_net.out.2.weight"], prm["update_net.out.2.bias"], prm["update_net.out.4.weight"], x)
    def fwd():
        return ops.cond_grow(x, T, goal, None, w, 3, seed=1, step0=0, keep_history=True)
    for _ in range(warm):   # the history ring is GBs: let the caching allocator settle before timing
        out, states, pre = fwd()
        g = ops.cond_grow_backward(states, pre, goal, None, w, cot, T, 3, seed=1, step0=0)
        del out, states, pre, g
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(iters):
        e[0].record(); out, states, pre = fwd(); e[1].record()
        g = ops.cond_grow_backward(states, pre, goal, None, w, cot, T, 3, seed=1, step0=0); e[2].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
        del out, states, pre, g     # as a trainer does after optimizer.step(): the next forward reuses the cached ring
    tf /= iters; tb /= iters
    cells = B * H * W * T
    print(json.dumps({"B": B, "T": T, "fwd_ms": tf, "bwd_ms": tb, "fwd_us_per_step": tf / T * 1e3, "bwd_us_per_step": tb / T * 1e3,
                      "fwd_Gcells_s": cells / tf / 1e6, "fwd_bwd_Gcells_s": cells / (tf + tb) / 1e6,
                      "bwd_over_fwd": tb / tf}))

import sys
if len(sys.argv) > 1 and sys.argv[1] == 'cfg3':
    run(32, 96, iters=2, warm=1)      # BASELINE configs[2] shape: B=32, 96 steps, forward with history + backward
    sys.exit(0)
run(8, 16)


def run_dynca(B=8, C=16, fc=128, cc=3, H=256, W=256, T=32, pad="circular"):
    dev = "cuda"
    gen = torch.Generator().manual_seed(0)
    k1 = 4 * C + cc
    w = ops.DyncaWeights(torch.randn(fc, k1, generator=gen) * (0.5 / k1 ** 0.5), torch.randn(fc, generator=gen) * 0.1,
                         torch.randn(C, fc, generator=gen) * (0.02 / fc ** 0.5), torch.zeros(C), torch.zeros(1, device=dev))
    x = (torch.rand(B, C, H, W, generator=gen) - 0.5).to(dev)
    cond = (torch.rand(B, cc, H, W, generator=gen) * 2 - 1).to(dev) if cc else None
    ops.dynca_nsteps(x, T, cond, None, w, pad, 0.5, seed=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        ops.dynca_nsteps(x, T, cond, None, w, pad, 0.5, seed=1)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    flops = 2 * (27 * C + fc * (5 * C + cc))
    print(json.dumps({"dynca": True, "B": B, "C": C, "fc": fc, "us_per_step": ms / T * 1e3, "Gcells_s": B * H * W * T / ms / 1e6,
                      "TFLOPs": B * H * W * T * flops / ms / 1e9, "frac_f32_mfma": B * H * W * T * flops / ms / 1e9 / 157.3}))

run_dynca()
run_dynca(C=12, fc=96)
run_dynca(B=2, C=32, fc=256, H=512, W=512, T=8)   # SURVEY 8d's cfg5 shape: two launches per step (fc slices of 128)
run_dynca(B=2, C=32, fc=128, H=512, W=512, T=8)
