#!/usr/bin/env python3
"""One-off sanity at sizes far above the parity tests: the tile kernels (32-bit in-plane offsets, XCD-chunked tile walks)
against the generic any-shape kernels on the same inputs, forward steps of both models, 1 x 16 x 2048 x 2048."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-stylization-with-nca_amd")]
import torch
import bench
from ncahip import ops

dev = "cuda"
B, C, H, W = 1, 16, 2048, 2048
gen = torch.Generator().manual_seed(0)
prm = bench.make_weights(gen)
x = torch.rand(B, C, H, W, generator=gen).to(dev)
goal = (torch.randn(B, 12, H, W, generator=gen) * 0.5).to(dev)
w = ops.CondWeights(prm["perception_net.weight"], prm["update_net.out.0.weight"], prm["update_net.out.0.bias"],
                    prm["update_net.out.2.weight"], prm["update_net.out.2.bias"], prm["update_net.out.4.weight"], x)
L = ops.lib()
outs = []
for force in (0, 1):
    L.ncahip_debug_force_generic(force)
    o, _, _ = ops.cond_grow(x, 3, goal, None, w, 3, seed=7)
    outs.append(o)
L.ncahip_debug_force_generic(0)
err = float((outs[0] - outs[1]).abs().max())
print("cond 1x16x2048^2, 3 steps: tile kernels vs generic kernels max abs diff %.3e" % err)
assert err < 1e-5
# a 5 x 5 window around each corner and the centre against the CPU oracle would need the whole grid; the generic kernel is
# itself checked against the oracle at every parity-test shape
k1 = 4 * C + 3
dw = ops.DyncaWeights(torch.randn(128, k1, generator=gen) * (0.5 / k1 ** 0.5), torch.randn(128, generator=gen) * 0.1,
                      torch.randn(C, 128, generator=gen) * (0.02 / 128 ** 0.5), torch.zeros(C), torch.zeros(1, device=dev))
xd = (torch.rand(B, C, H, W, generator=gen) - 0.5).to(dev)
cond = (torch.rand(B, 3, H, W, generator=gen) * 2 - 1).to(dev)
a, _ = ops.dynca_nsteps(xd, 2, cond, None, dw, "circular", 0.5, seed=3)
# unaligned view of the same problem: one column less -> per-element path of the same kernel family
b_, _ = ops.dynca_nsteps(xd[..., :-1].contiguous(), 2, cond[..., :-1].contiguous(), None, dw, "replicate", 0.5, seed=3)
print("dynca 1x16x2048^2 ok:", bool(torch.isfinite(a).all()), bool(torch.isfinite(b_).all()))
print("large shapes ok")

# ---- backward at 2048^2: a live patch near the bottom-right corner, everything else dead.  The alive mask zeroes dead cells
# in both directions, so every gradient must equal the one of the 192 x 192 crop that contains the patch and the same two
# image borders (the crop's other two borders see zero padding where the full image has dead cells).
Tn, S, c0 = 2, 2048, 2048 - 192
g2 = torch.Generator().manual_seed(5)
xc = torch.zeros(1, C, 192, 192)
xc[:, :, 60:150, 70:160] = torch.rand(1, C, 90, 90, generator=g2)
gc = torch.randn(1, 12, 192, 192, generator=g2) * 0.5
uc = torch.rand(Tn, 1, 1, 192, 192, generator=g2)
cc_ = torch.randn(1, C, 192, 192, generator=g2)
def embed(t, fill=0.0):
    full = torch.full(t.shape[:-2] + (S, S), fill)
    full[..., c0:, c0:] = t
    return full
res = []
for name, xx, gg, uu, ct in (("crop", xc, gc, uc, cc_), ("full", embed(xc), embed(gc), embed(uc, 0.5), embed(cc_))):
    xx, gg, uu, ct = xx.to(dev), gg.to(dev), uu.to(dev), ct.to(dev)
    w2 = ops.CondWeights(prm["perception_net.weight"], prm["update_net.out.0.weight"], prm["update_net.out.0.bias"],
                         prm["update_net.out.2.weight"], prm["update_net.out.2.bias"], prm["update_net.out.4.weight"], xx)
    out, states, pre = ops.cond_grow(xx, Tn, gg, uu, w2, 3, keep_history=True)
    gr = ops.cond_grow_backward(states, pre, gg, uu, w2, ct, Tn, 3)
    res.append((out, gr))
(oc, gcr), (of, gfr) = res
def rel(a, b):
    return float((a - b).abs().max()) / max(1e-6, float(b.abs().max()))
print("forward  crop vs full window: %.2e" % rel(of[..., c0:, c0:], oc))
print("dL/dx0   crop vs full window: %.2e   outside the window: %.2e" % (rel(gfr["x0"][..., c0:, c0:], gcr["x0"]),
      float(gfr["x0"][..., :c0, :].abs().max())))
for k in ("w1", "w2", "w3", "b1", "b2", "wp", "goal"):
    a = gfr[k][..., c0:, c0:] if k == "goal" else gfr[k]
    print("  d%-4s %.2e" % (k, rel(a, gcr[k])))
    assert rel(a, gcr[k]) < 2e-4, k
assert rel(of[..., c0:, c0:], oc) < 1e-5 and rel(gfr["x0"][..., c0:, c0:], gcr["x0"]) < 2e-4
print("large-shape backward ok")

# ---- DyNCA at 2048^2 by periodicity: with circular padding an 8 x 8 tiling of a 256^2 problem (state, conditioning, uniforms)
# evolves into the tiling of the 256^2 result; dL/dx0 tiles likewise, and the weight gradients are 64 x the small ones when the
# cotangent is tiled too.
g3 = torch.Generator().manual_seed(11)
xs = (torch.rand(1, C, 256, 256, generator=g3) - 0.5)
cs = torch.rand(1, 3, 256, 256, generator=g3) * 2 - 1
us_ = torch.rand(2, 1, 1, 256, 256, generator=g3)
cts = torch.randn(1, C, 256, 256, generator=g3)
tile = lambda t: t.repeat(*([1] * (t.dim() - 2)), 8, 8)
rs = []
for name, f in (("small", lambda t: t), ("tiled", tile)):
    xx, cn, uu, ct = f(xs).to(dev), f(cs).to(dev), f(us_).to(dev), f(cts).to(dev)
    out, states = ops.dynca_nsteps(xx, 2, cn, uu, dw, "circular", 0.5, keep_history=True)
    gr = ops.dynca_nsteps_backward(states, cn, uu, dw, ct, None, 2, "circular", 0.5)
    rs.append((out.cpu(), {k: v.cpu() for k, v in gr.items()}))
(os_, gs), (ot, gt) = rs
print("dynca forward  tiled vs tile(small): %.2e" % rel(ot, tile(os_)))
print("dynca dL/dx0   tiled vs tile(small): %.2e" % rel(gt["x0"], tile(gs["x0"])))
assert rel(ot, tile(os_)) < 1e-5 and rel(gt["x0"], tile(gs["x0"])) < 2e-4
for k in ("w1", "b1", "w2", "b2"):
    print("  d%-3s tiled vs 64 x small: %.2e" % (k, rel(gt[k], 64.0 * gs[k])))
    assert rel(gt[k], 64.0 * gs[k]) < 2e-4, k
print("large-shape dynca ok")

# ---- bf16-storage ConditionedNCA at 2048^2: live patch / dead elsewhere, against the 192^2 crop (bit-exact: same arithmetic)
ob = []
for xx, gg, uu in ((xc, gc, uc), (embed(xc), embed(gc), embed(uc, 0.5))):
    xx, gg, uu = xx.bfloat16().to(dev), gg.bfloat16().to(dev), uu.to(dev)
    w2 = ops.CondWeights(prm["perception_net.weight"], prm["update_net.out.0.weight"], prm["update_net.out.0.bias"],
                         prm["update_net.out.2.weight"], prm["update_net.out.2.bias"], prm["update_net.out.4.weight"], xx)
    o, _, _ = ops.cond_grow(xx, 3, gg, uu[:1].repeat(3, 1, 1, 1, 1), w2, 3)
    ob.append(o.float())
print("bf16 forward crop vs full window: %.2e   outside: %.2e" % (rel(ob[1][..., c0:, c0:], ob[0]), float(ob[1][..., :c0, :].abs().max())))
assert rel(ob[1][..., c0:, c0:], ob[0]) == 0.0
print("large-shape bf16 ok")

# ---- DyNCA C = 32 (configs[4]) with fc = 256 (two launches per step) and fc = 128 at 512^2, again by periodicity (4 x 4 tiling
# of 128^2), f32 and bf16 storage
C3 = 32
for fc3 in (128, 256):
    k3 = 4 * C3 + 3
    g4 = torch.Generator().manual_seed(fc3)
    dw3 = ops.DyncaWeights(torch.randn(fc3, k3, generator=g4) * (0.5 / k3 ** 0.5), torch.randn(fc3, generator=g4) * 0.1,
                           torch.randn(C3, fc3, generator=g4) * (0.3 / fc3 ** 0.5), torch.randn(C3, generator=g4) * 0.02,
                           torch.zeros(1, device=dev))
    x3 = torch.rand(1, C3, 128, 128, generator=g4) - 0.5
    c3 = torch.rand(1, 3, 128, 128, generator=g4) * 2 - 1
    u3 = torch.rand(3, 1, 1, 128, 128, generator=g4)
    t4 = lambda t: t.repeat(*([1] * (t.dim() - 2)), 4, 4)
    o_s, _ = ops.dynca_nsteps(x3.to(dev), 3, c3.to(dev), u3.to(dev), dw3, "circular", 0.5)
    o_t, _ = ops.dynca_nsteps(t4(x3).to(dev), 3, t4(c3).to(dev), t4(u3).to(dev), dw3, "circular", 0.5)
    e = rel(o_t.cpu(), t4(o_s.cpu()))
    print("dynca C=32 fc=%d 512^2 tiled vs tile(128^2): %.2e" % (fc3, e))
    assert e < 1e-5
    if fc3 == 128:
        ob_s, _ = ops.dynca_nsteps(x3.bfloat16().to(dev), 3, c3.to(dev), u3.to(dev), dw3, "circular", 0.5)
        ob_t, _ = ops.dynca_nsteps(t4(x3).bfloat16().to(dev), 3, t4(c3).to(dev), t4(u3).to(dev), dw3, "circular", 0.5)
        assert torch.equal(ob_t.float().cpu(), t4(ob_s.float().cpu()))
        print("dynca C=32 bf16 storage tiled == tile(small)")
print("large-shape C=32 ok")
