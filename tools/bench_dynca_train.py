#!/usr/bin/env python3
"""Timing of the DyNCA training-shaped pass: forward_nsteps with history + backward (ops level), B=8 C=16 fc=128 256^2."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-stylization-with-nca_amd")]
import torch
from ncahip import ops

def run(B=8, C=16, fc=128, cc=3, H=256, W=256, T=16, pad="circular", iters=3):
    dev = "cuda"
    gen = torch.Generator().manual_seed(0)
    k1 = 4 * C + cc
    w = ops.DyncaWeights(torch.randn(fc, k1, generator=gen) * (0.5 / k1 ** 0.5), torch.randn(fc, generator=gen) * 0.1,
                         torch.randn(C, fc, generator=gen) * (0.02 / fc ** 0.5), torch.zeros(C), torch.zeros(1, device=dev))
    x = (torch.rand(B, C, H, W, generator=gen) - 0.5).to(dev)
    cond = (torch.rand(B, cc, H, W, generator=gen) * 2 - 1).to(dev) if cc else None
    cot = torch.randn(B, C, H, W, generator=gen).to(dev)
    def fwd():
        return ops.dynca_nsteps(x, T, cond, None, w, pad, 0.5, seed=1, keep_history=True)
    for _ in range(2):
        out, states = fwd()
        g = ops.dynca_nsteps_backward(states, cond, None, w, cot, None, T, pad, 0.5, seed=1)
        del out, states, g
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(iters):
        e[0].record(); out, states = fwd(); e[1].record()
        g = ops.dynca_nsteps_backward(states, cond, None, w, cot, None, T, pad, 0.5, seed=1); e[2].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
        del out, states, g
    tf /= iters; tb /= iters
    cells = B * H * W * T
    print(json.dumps({"dynca_train": True, "B": B, "C": C, "fc": fc, "T": T, "fwd_us_per_step": tf / T * 1e3,
                      "bwd_us_per_step": tb / T * 1e3, "fwd_bwd_Gcells_s": cells / (tf + tb) / 1e6, "bwd_over_fwd": tb / tf}))

run()
run(C=12, fc=96)
