#!/usr/bin/env python3
"""Diagnostic: cycles per phase of backward kernel A (stamps build, `make stamps`), summed over each wave's tiles."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "video-stylization-with-nca_amd")
os.environ["NCAHIP_LIB"] = os.path.join(PKG, "libncahip_stamps.so")
sys.path[:0] = [ROOT, PKG]
import bench
from ncahip import ops
B, C, H, W, T = 8, 16, 256, 256, 2
if os.environ.get("NCAHIP_STAMP_SHAPE"):     # e.g. "8,20,64,64": the reference's default training shape (front + matrix kernels)
    B, C, H, W = [int(v) for v in os.environ["NCAHIP_STAMP_SHAPE"].split(",")]
    bench.C = C
dev = "cuda"
gen = torch.Generator().manual_seed(0)
prm = bench.make_weights(gen)
DT = torch.bfloat16 if "bf16" in sys.argv[1:] else torch.float32      # bf16: the bf16-history / bf16-MFMA backward
x = torch.rand(B, C, H, W, generator=gen).to(dev, DT)
goal = (torch.randn(B, C - 4, H, W, generator=gen) * 0.5).to(dev, DT)
cot = torch.randn(B, C, H, W, generator=gen).to(dev)
w = ops.CondWeights(prm["perception_net.weight"], prm["update_net.out.0.weight"], prm["update_net.out.0.bias"],
                    prm["update_net.out.2.weight"], prm["update_net.out.2.bias"], prm["update_net.out.4.weight"], x)
L = ops.lib()
L.nca_debug_set_stamp_buffer_pc.argtypes = [ctypes.c_void_p]
out, states, pre = ops.cond_grow(x, T, goal, None, w, 3, seed=int(os.environ.get("NCAHIP_STAMP_SEED", "1")), keep_history=True)
for _ in range(2):
    ops.cond_grow_backward(states, pre, goal, None, w, cot, T, 3, seed=int(os.environ.get("NCAHIP_STAMP_SEED", "1")))
torch.cuda.synchronize()
NW = 256 * 4
buf = torch.zeros(NW * 16 + 4096 * 64, dtype=torch.int64, device=dev)
L.nca_debug_set_stamp_buffer_pc(buf.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.cond_grow_backward(states, pre, goal, None, w, cot, 1, 3, seed=int(os.environ.get("NCAHIP_STAMP_SEED", "1"))) if False else ops.cond_grow_backward(states, pre, goal, None, w, cot, T, 3, seed=int(os.environ.get("NCAHIP_STAMP_SEED", "1")))
e1.record(); torch.cuda.synchronize()
L.nca_debug_set_stamp_buffer_pc(None)
k = buf[:NW * 16].cpu().numpy().reshape(NW, 16).astype(np.float64)   # last launch (t = 0) wins
# phases 1-3 belong to the one-launch kernel (the front + matrix form has no staging in its matrix kernel: they read 0 there)
names = ["loop/tail", "fwd staging", "x'/g loads, z out", "gate", "perception", "fwd recompute", "layer 3", "layer 2",
         "layer 1", "dP out", "start-up", "slab flush"]
tot = k[:, :12].sum(1)
print("backward of %d steps: %.1f us/step (event, stamps build)" % (T, e0.elapsed_time(e1) * 1e3 / T))
print("per wave: total %.0f cycles (median), min %.0f max %.0f  [100 MHz s_memtime ticks x clock ratio apply]" % (np.median(tot), tot.min(), tot.max()))
print("kernel span (entry -> before the slab flush): %.1f us (s_memrealtime), %.0f shader ticks -> %.2f GHz" % (np.median(k[:, 12]) / 100.0, np.median(k[:, 13]), np.median(k[:, 13]) / np.median(k[:, 12]) / 10.0))
for i, n in enumerate(names):
    print("  %-20s %9.0f  %5.1f %%" % (n, np.median(k[:, i]), 100 * np.median(k[:, i]) / np.median(tot)))
