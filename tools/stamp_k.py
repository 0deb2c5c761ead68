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
L = ops.lib()
L.nca_debug_set_stamp_buffer_pc.argtypes = [ctypes.c_void_p]
NWG = 256
buf = torch.zeros(NWG * 8 * 8 * 16 + NWG * 8 * 8, dtype=torch.int64, device=dev)
def run(tag, seed=0):
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
        pw = k[:, 4:8]
        print("   producer tile 0: entry->loads issued %.0f   wp fill %.0f   stage %.0f   perception+P write %.0f   (arrive at barrier %.0f)" % (
            np.median(pw[..., 4] - pw[..., 0]), np.median(pw[..., 5] - pw[..., 4]), np.median(pw[..., 6] - pw[..., 5]),
            np.median(pw[..., 1] - pw[..., 6]), np.median(pw[..., 1] - pw[..., 0])))
        print("   consumer: entry->weights issued + biases %.0f   weights landed %.0f   (arrive at barrier %.0f)" % (
            np.median(v[..., 4] - v[..., 0]), np.median(v[..., 5] - v[..., 4]), np.median(v[..., 1] - v[..., 0])))
    print(f"{tag:34s} startup {np.median(v[..., 2] - v[..., 0]):7.0f}   loop {np.median(v[..., 3] - v[..., 2]):8.0f} = {np.median(v[..., 3] - v[..., 2]) / 8:7.0f} per tile   whole {np.median(v[..., 3] - v[..., 0]):8.0f} cycles")

run("normal")
run("idle producers", 0xD1A6)
run("idle consumers", 0xD1A7)
run("consumer: no perception", 0xD1A8)
run("consumer: no MLP", 0xD1A9)

# fused launch: stamps of step 1 (warm) inside an 8-step persistent launch
states = torch.empty(2, B, C, H, W, device=dev); states[0].copy_(x)
prebuf = torch.empty(2, B, H, W, device=dev, dtype=torch.uint8)
outb = torch.empty_like(x)
def grow(T):
    ops.check(L.ncahip_cond_grow_fwd_f32(states.data_ptr(), prebuf.data_ptr(), 2, T, outb.data_ptr(), goal.data_ptr(), 12, None,
                                         w.wp.data_ptr(), w.w1.data_ptr(), w.b1.data_ptr(), w.w2.data_ptr(), w.b2.data_ptr(), w.w3.data_ptr(),
                                         B, C, H, W, 64, 3, 0.1, 0.5, -10.0, 10.0, 42, 0, torch.cuda.current_stream().cuda_stream), "grow")
grow(8); torch.cuda.synchronize()
buf.zero_()
L.nca_debug_set_stamp_buffer_pc(buf.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); grow(8); e1.record(); torch.cuda.synchronize()
L.nca_debug_set_stamp_buffer_pc(None)
print("fused 8-step grow: %.1f us per step (events, incl. finalize)" % (e0.elapsed_time(e1) * 1e3 / 8))
k = buf[NWG * 8 * 8 * 16:].cpu().numpy().reshape(NWG, 8, 8).astype(np.float64)
for nm, sl in (("consumers", slice(0, 4)), ("producers", slice(4, 8))):
    v = k[:, sl]
    print(f"  {nm}: step start -> first-tile barrier arrive {np.median(v[..., 1] - v[..., 0]):.0f}  wait {np.median(v[..., 2] - v[..., 1]):.0f}   "
          f"tile loop {np.median(v[..., 3] - v[..., 2]):.0f} = {np.median(v[..., 3] - v[..., 2]) / 8:.0f}/tile   grid barrier {np.median(v[..., 7] - v[..., 3]):.0f}   step total {np.median(v[..., 7] - v[..., 0]):.0f}")
v = k[:, 0]
loop = v[:, 3] - v[:, 0]
gb = v[:, 7] - v[:, 3]
print("  per-WG (wave 0): step work (start->loop end) min %.0f  p10 %.0f  median %.0f  p90 %.0f  max %.0f" % (loop.min(), np.percentile(loop, 10), np.median(loop), np.percentile(loop, 90), loop.max()))
print("  per-WG grid barrier wait: min %.0f  p10 %.0f  median %.0f  p90 %.0f  max %.0f" % (gb.min(), np.percentile(gb, 10), np.median(gb), np.percentile(gb, 90), gb.max()))
idx = np.argsort(loop)[-8:]
print("  slowest WGs:", idx.tolist(), " their work:", loop[idx].astype(int).tolist())
print("  work by XCD (wg % 8): ", [int(np.median(loop[x::8])) for x in range(8)])
