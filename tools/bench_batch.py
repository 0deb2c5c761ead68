#!/usr/bin/env python3
"""fp32 grow-loop throughput vs batch size (fixed launch costs amortise over more tiles per workgroup)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-stylization-with-nca_amd")]
import bench
from ncahip import ops
C, H, W, T = bench.C, bench.H, bench.W, 32
dev = "cuda"
gen = torch.Generator().manual_seed(0)
prm = bench.make_weights(gen)
for B in (4, 8, 16, 32):
    x = torch.rand(B, C, H, W, generator=gen).to(dev)
    goal = (torch.randn(B, 12, H, W, generator=gen) * 0.5).to(dev)
    w = ops.CondWeights(prm["perception_net.weight"], prm["update_net.out.0.weight"], prm["update_net.out.0.bias"],
                        prm["update_net.out.2.weight"], prm["update_net.out.2.bias"], prm["update_net.out.4.weight"], x)
    row = {"B": B}
    for name, xx, gg in (("f32", x, goal), ("bf16", x.bfloat16(), goal.bfloat16())):
        ops.cond_grow(xx, T, gg, None, w, 3, seed=1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(3):
            ops.cond_grow(xx, T, gg, None, w, 3, seed=1, step0=T * i)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        row[name] = {"us_per_step": round(ms * 1e3 / T, 1), "Gcells_s": round(B * H * W * T / ms / 1e6, 2),
                     "frac_f32_mfma": round(B * H * W * T * bench.FLOPS_PER_CELL / ms / 1e9 / 157.3, 3)}
    print(json.dumps(row))
