#!/usr/bin/env python3
"""Diagnostic: per-kernel phase stamps of the producer/consumer kernel (light stamps build, `make stamps`):
entry -> first barrier (weight image / first tile) -> loop -> end, per workgroup, against the event-timed launch."""
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
x0, goal0 = x, goal
L = ops.lib()
L.nca_debug_set_stamp_buffer_pc.argtypes = [ctypes.c_void_p]
NWG = 256
buf = torch.zeros(NWG * 8 * 8 * 16 + NWG * 8 * 8, dtype=torch.int64, device=dev)
def run(tag, seed=0, bf16=False):
    global x, goal
    x, goal = (x0.bfloat16(), goal0.bfloat16()) if bf16 else (x0, goal0)
    buf.zero_()
    xp, pre = ops.cond_step(x, None, goal, None, w, 3)
    for _ in range(5):
        xp2, pre2 = ops.cond_step(xp, pre, goal, None, w, 3, step=1)
    torch.cuda.synchronize()
    L.nca_debug_set_stamp_buffer_pc(buf.data_ptr())
    xp2, pre2 = ops.cond_step(xp, pre, goal, None, w, 3, step=2, seed=seed)
    torch.cuda.synchronize()
    L.nca_debug_set_stamp_buffer_pc(None)
    k = buf[NWG * 8 * 8 * 16:].cpu().numpy().reshape(NWG, 8, 8).astype(np.float64)
    v = k[:, 0:4]
    if seed == 0:
        print("   entry -> first barrier arrival: consumers %.0f   producers %.0f" % (
            np.median(k[:, 0:4, 1] - k[:, 0:4, 0]), np.median(k[:, 4:8, 1] - k[:, 4:8, 0])))
        c = k[:, 0:4]
        print("   consumer, tile 2: consume %.0f   barrier wait %.0f   next tile's top after %.0f cycles" % (
            np.median(c[..., 5] - c[..., 4]), np.median(c[..., 6] - c[..., 5]), np.median(c[..., 7] - c[..., 6])))
    print(f"{tag:34s} startup {np.median(v[..., 2] - v[..., 0]):7.0f}   loop {np.median(v[..., 3] - v[..., 2]):8.0f} = {np.median(v[..., 3] - v[..., 2]) / 8:7.0f} per tile   whole {np.median(v[..., 3] - v[..., 0]):8.0f} cycles")

run("normal")
run("idle producers", 0xD1A6)
run("idle consumers", 0xD1A7)
run("consumer: no perception", 0xD1A8)
run("consumer: no MLP", 0xD1A9)
print("---- bf16 storage")
run("bf16 normal", 0, True)
run("bf16 idle producers", 0xD1A6, True)
run("bf16 idle consumers", 0xD1A7, True)
run("bf16 consumer: no perception", 0xD1A8, True)
run("bf16 consumer: no MLP", 0xD1A9, True)
run("bf16 consumer alone, no perception", 0xD1AA, True)
run("bf16 consumer alone, no MLP", 0xD1AB, True)
run("bf16 consumer alone, no store", 0xD1AC, True)
run("bf16 consumer alone, nothing", 0xD1AD, True)
run("f32 consumer alone, nothing", 0xD1AD, False)
print("---- per-workgroup spread of (entry -> loop end), fp32 normal run")
x, goal = x0, goal0
buf.zero_()
xp, pre = ops.cond_step(x, None, goal, None, w, 3)
for _ in range(3):
    xp2, pre2 = ops.cond_step(xp, pre, goal, None, w, 3, step=1)
torch.cuda.synchronize()
L.nca_debug_set_stamp_buffer_pc(buf.data_ptr())
xp2, pre2 = ops.cond_step(xp, pre, goal, None, w, 3, step=2)
torch.cuda.synchronize()
L.nca_debug_set_stamp_buffer_pc(None)
k = buf[NWG * 8 * 8 * 16:].cpu().numpy().reshape(NWG, 8, 8).astype(np.float64)
wl = (k[:, 0, 3] - k[:, 0, 0])
print("   whole: min %.0f  p10 %.0f  median %.0f  p90 %.0f  max %.0f   (max/median %.3f)" % (wl.min(), np.percentile(wl, 10), np.median(wl), np.percentile(wl, 90), wl.max(), wl.max() / np.median(wl)))
print("---- store on the critical path?")
run("f32 normal", 0, False)
run("f32 normal, no store", 0xD1AE, False)
print("---- fp32 consumer alone")
run("f32 consumer alone", 0xD1A6, False)
run("f32 consumer alone, no perception", 0xD1AA, False)
run("f32 consumer alone, no MLP", 0xD1AB, False)
run("f32 consumer alone, no store", 0xD1AC, False)
run("f32 consumer alone, nothing", 0xD1AD, False)
