#!/usr/bin/env python3
"""Probe: does replaying the 64-launch grow loop as a HIP graph shorten the launch cadence?  (bench.py's headline loop, B=8 C=16 256^2.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-stylization-with-nca_amd")]
import torch
import bench
from ncahip import ops
dev = torch.device("cuda", 0)
B, C, H, W, T = bench.B, bench.C, bench.H, bench.W, bench.T
gen = torch.Generator().manual_seed(0)
prm = bench.make_weights(gen)
x0 = torch.rand(B, C, H, W, generator=gen).to(dev)
goal = (torch.randn(B, 12, H, W, generator=gen) * 0.5).to(dev)
w = ops.CondWeights(prm["perception_net.weight"], prm["update_net.out.0.weight"], prm["update_net.out.0.bias"],
                    prm["update_net.out.2.weight"], prm["update_net.out.2.bias"], prm["update_net.out.4.weight"], x0)
states = torch.empty(2, B, C, H, W, device=dev); pre = torch.empty(2, B, H, W, device=dev, dtype=torch.uint8); out = torch.empty_like(x0)
L = ops.lib()
def grow(stream):
    ops.check(L.ncahip_cond_grow_fwd_f32(states.data_ptr(), pre.data_ptr(), 2, T, out.data_ptr(), goal.data_ptr(), 12, None, w.wp.data_ptr(),
                                         w.w1.data_ptr(), w.b1.data_ptr(), w.w2.data_ptr(), w.b2.data_ptr(), w.w3.data_ptr(), B, C, H, W, 64, 3, 0.1, 0.5,
                                         -10.0, 10.0, 42, 0, stream), "grow")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def plain():
    states[0].copy_(x0); grow(torch.cuda.current_stream().cuda_stream)
print("plain launches : %.3f ms per grow (%.2f us per step)" % (timeit(plain), timeit(plain) / T * 1e3))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    plain()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    states[0].copy_(x0); grow(torch.cuda.current_stream().cuda_stream)
ref = out.clone()
def replay(): g.replay()
t = timeit(replay)
print("graph replay   : %.3f ms per grow (%.2f us per step)" % (t, t / T * 1e3))
plain(); torch.cuda.synchronize()
print("same result as plain launches:", bool(torch.equal(out, ref)))
