#!/usr/bin/env python3
"""Diagnostic: producer / consumer / barrier-wait cycles per tile of the wave-specialised kernel (stamps build)."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "video-stylization-with-nca_amd")
os.environ["NCAHIP_LIB"] = os.path.join(PKG, "libncahip_stamps.so")
sys.path[:0] = [ROOT, PKG]
import bench
from ncahip import ops
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
import sys as _s
def run(tag, u=None, seed=0):
    buf.zero_()
    xp, pre = ops.cond_step(x, None, goal, None, w, 3)
    for _ in range(5):
        xp2, pre2 = ops.cond_step(xp, pre, goal, None, w, 3, step=1)
    torch.cuda.synchronize()
    L.nca_debug_set_stamp_buffer_pc(buf.data_ptr())
    xp2, pre2 = ops.cond_step(xp, pre, goal, u, w, 3, step=2, seed=seed)
    torch.cuda.synchronize()
    L.nca_debug_set_stamp_buffer_pc(None)
    s = buf.cpu().numpy().reshape(256, 8, NT, 16).astype(np.float64)
    print("====", tag)
    for name, sl in (("consumer waves 0-3", slice(0, 4)), ("producer waves 4-7", slice(4, 8))):
        v = s[:, sl]
        ok = v[..., 2] > 0
        work = (v[..., 1] - v[..., 0])[ok]; wait = (v[..., 2] - v[..., 1])[ok]
        print(f"{name}: tiles {int(ok.sum())}  work median {np.median(work):.0f} mean {work.mean():.0f}   barrier wait median {np.median(wait):.0f} mean {wait.mean():.0f}")
    v = s[:, 0]
    span = (v[:, :, 2].max(axis=1) - np.where(v[:, :, 0] > 0, v[:, :, 0], np.inf).min(axis=1))
    print("loop span per WG (cycles): median", np.median(span))
    c = s[:, 0:4]; ok = c[..., 8] > 0
    if ok.any():
        for nm, a0, a1 in (("consumer: perception pass 0", 4, 5), ("consumer: MLP pass 0 (256 MFMA)", 5, 6), ("consumer: perception+MLP pass 1", 6, 7), ("consumer: store", 7, 8)):
            v = (c[..., a1] - c[..., a0])[ok]; print(f"  {nm:36s} median {np.median(v):8.0f} mean {v.mean():8.0f}")
    pw = s[:, 4:8]; ok = pw[..., 6] > 0
    if ok.any():
        for nm, a0, a1 in (("producer: issue loads", 4, 5), ("producer: stage (incl. load wait)", 5, 6)):
            v = (pw[..., a1] - pw[..., a0])[ok]; print(f"  {nm:36s} median {np.median(v):8.0f} mean {v.mean():8.0f}")

L.nca_debug_set_stamp_buffer_pc.argtypes = [ctypes.c_void_p]
run("normal (in-kernel philox)")
run("explicit uniforms (no philox)", u=torch.rand(B, 1, H, W, device=dev))
run("idle producers", seed=0xD1A6)
run("idle consumers", seed=0xD1A7)
