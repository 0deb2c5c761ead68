#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the wave-private ConditionedNCA step kernel from in-kernel
s_memtime stamps (libncahip_stamps.so, `make -C video-stylization-with-nca_amd stamps`).
Read SHARES, not absolute run time (the stamped build is fenced)."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "video-stylization-with-nca_amd")
os.environ["NCAHIP_LIB"] = os.path.join(PKG, "libncahip_stamps.so")
sys.path[:0] = [ROOT, PKG]
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ncahip import ops  # noqa: E402

B, C, H, W = bench.B, bench.C, bench.H, bench.W
dev = "cuda"
gen = torch.Generator().manual_seed(0)
prm = bench.make_weights(gen)
x = torch.rand(B, C, H, W, generator=gen).to(dev)
goal = (torch.randn(B, 12, H, W, generator=gen) * 0.5).to(dev)
w = ops.CondWeights(prm["perception_net.weight"], prm["update_net.out.0.weight"], prm["update_net.out.0.bias"],
                    prm["update_net.out.2.weight"], prm["update_net.out.2.bias"], prm["update_net.out.4.weight"], x)
L = ops.lib()
NW, NT = 256 * 8, 8
buf = torch.zeros(NW * NT * 16, dtype=torch.int64, device=dev)
xp, pre = ops.cond_step(x, None, goal, None, w, 3)          # warm, unstamped (dbg null)
for _ in range(5):
    xp2, pre2 = ops.cond_step(xp, pre, goal, None, w, 3, step=1)
torch.cuda.synchronize()
L.nca_debug_set_stamp_buffer_c.argtypes = [ctypes.c_void_p]
L.nca_debug_set_stamp_buffer_c(buf.data_ptr())
xp2, pre2 = ops.cond_step(xp, pre, goal, None, w, 3, step=2)
torch.cuda.synchronize()
L.nca_debug_set_stamp_buffer_c(None)
s = buf.cpu().numpy().reshape(NW, NT, 16).astype(np.float64)
names = ["goal loads + stage S1-S4", "perception pass 0", "issue next-tile loads (+philox)", "MLP pass 0 + perception/MLP pass 1", "store"]
LAST = len(names)
valid = s[:, :, LAST] > 0
d = np.diff(s[:, :, :LAST + 1], axis=2)
print(f"tiles stamped: {int(valid.sum())}; per-phase cycles (median / mean) over all stamped wave tiles")
tot = 0
for i, n in enumerate(names):
    v = d[:, :, i][valid]
    print(f"  {n:28s} {np.median(v):9.0f} {v.mean():9.0f}")
    tot += v.mean()
print(f"  {'tile total':28s} {'':9s} {tot:9.0f}")
gap = (s[:, 1:, 0] - s[:, :-1, LAST])[valid[:, 1:] & valid[:, :-1]]
print(f"  between tiles (loop overhead)  {np.median(gap):9.0f} {gap.mean():9.0f}")
per_wave = (s[:, :, LAST].max(axis=1) - np.where(valid, s[:, :, 0], np.inf).min(axis=1))[valid.any(axis=1)]
print(f"  wave lifetime in tiles: median {np.median(per_wave):.0f} cycles; tiles per wave {valid.sum(axis=1).mean():.2f}")

print("inside staging (cycles, median / mean):")
sub = ["goal-load issue (0->8)", "S1 write alpha (8->9)", "S2 life (9->10)", "S3 pn+mask (10->11)", "S4a state (11->12)", "S4b goal+z (12->13)", "S4 halo cols (13->1)"]
pts = [0, 8, 9, 10, 11, 12, 13, 1]
for i, n in enumerate(sub):
    v = (s[:, :, pts[i + 1]] - s[:, :, pts[i]])[valid & (s[:, :, 12] > 0)]
    print(f"  {n:28s} {np.median(v):9.0f} {v.mean():9.0f}")
