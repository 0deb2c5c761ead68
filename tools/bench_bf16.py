#!/usr/bin/env python3
"""Timing of the bf16-storage grow loop on the bench configuration (B=8, C=16, 256^2, T=64) next to the fp32 one."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-stylization-with-nca_amd")]
import bench
from ncahip import ops
B, C, H, W, T = bench.B, bench.C, bench.H, bench.W, 64
dev = "cuda"
gen = torch.Generator().manual_seed(0)
prm = bench.make_weights(gen)
x = torch.rand(B, C, H, W, generator=gen).to(dev)
goal = (torch.randn(B, 12, H, W, generator=gen) * 0.5).to(dev)
w = ops.CondWeights(prm["perception_net.weight"], prm["update_net.out.0.weight"], prm["update_net.out.0.bias"],
                    prm["update_net.out.2.weight"], prm["update_net.out.2.bias"], prm["update_net.out.4.weight"], x)
res = {}
for name, xx, gg in (("f32", x, goal), ("bf16", x.bfloat16(), goal.bfloat16())):
    ops.cond_grow(xx, T, gg, None, w, 3, seed=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 5
    for i in range(n):
        out, _, _ = ops.cond_grow(xx, T, gg, None, w, 3, seed=1, step0=64 * i)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    alive = float(ops.cond_alive(out.float(), 3).float().mean())
    res[name] = {"ms_per_grow": ms, "us_per_step": ms * 1e3 / T, "Gcells_s": B * H * W * T / ms / 1e6, "alive": alive}
print(json.dumps(res))
ops.set_cond_precision("bf16x3")
ops.cond_grow(x, T, goal, None, w, 3, seed=1); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(5):
    outs, _, _ = ops.cond_grow(x, T, goal, None, w, 3, seed=1, step0=64 * i)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
xs, _ = ops.cond_step(x, None, goal, None, w, 3, seed=5)
ops.set_cond_precision("exact")
xe, _ = ops.cond_step(x, None, goal, None, w, 3, seed=5)
err = float(((xs - xe).abs() / xe.abs().clamp_min(1.0)).max())
print(json.dumps({"f32_bf16x3": {"us_per_step": ms * 1e3 / T, "Gcells_s": B * H * W * T / ms / 1e6, "max_rel_err_vs_exact_one_step": err}}))
