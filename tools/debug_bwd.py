#!/usr/bin/env python3
"""Debug aid: one backward step of the ConditionedNCA path, every intermediate vs oracle autograd."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-stylization-with-nca_amd"), os.path.join(ROOT, "tests")]
import torch
from oracle import nca_oracle as O
from ncahip import ops
from ncahip.ops import _p, _stream, lib, check

def err(name, got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    e = (got - ref).abs()
    print(f"  {name:10s} max|ref| {float(ref.abs().max()):9.3e}  max err {float(e.max()):9.3e}  frac>1e-4*scale {float((e > 1e-4 * max(float(ref.abs().max()),1e-9)).float().mean()):.4f}  nan {int(torch.isnan(got).sum())}")

def run(C, B, H, W, gch, alive, seed=0):
    print(f"=== C={C} B={B} {H}x{W} goal_ch={gch} alive={alive}")
    gen = torch.Generator().manual_seed(seed)
    from test_gpu_parity import rand_cond_prm
    prm = rand_cond_prm(C, seed=C + 1, out_scale=2.0)
    x0 = torch.rand(B, C, H, W, generator=gen)
    goal = torch.randn(B, gch, H, W, generator=gen)
    u = torch.rand(1, B, 1, H, W, generator=gen)
    cot = torch.randn(B, C, H, W, generator=gen)
    # oracle with intermediates
    xr = x0.clone().requires_grad_(True)
    g = O.cond_pad_goal(goal, C).clone().requires_grad_(True)
    p = {k: v.clone().requires_grad_(True) for k, v in prm.items()}
    ua = alive >= 0
    pre = O.cond_alive(xr, max(alive, 0), 0.1, ua)
    z = xr + g * pre; z.retain_grad()
    pp = O.cond_perceive(z, p["perception_net.weight"]); pp.retain_grad()
    out = O.cond_update_net(pp, p)
    x1 = xr + O.cond_fire_mask(u[0], 0.5) * out; x1.retain_grad()
    post = O.cond_alive(x1, max(alive, 0), 0.1, ua)
    x2 = torch.clamp(x1 * (pre & post).float(), -10, 10)
    (x2 * cot).sum().backward()
    dev = "cuda"
    from test_gpu_parity import cond_w
    w = cond_w(ops, prm, x0.to(dev))
    xT, states, prem = ops.cond_grow(x0.to(dev), 1, goal.to(dev), u.to(dev), w, alive, keep_history=True)
    err("x_final", xT, x2)
    hid = 64
    nbytes = lib().ncahip_cond_grow_bwd_workspace(B, C, H, W, hid)
    ws = torch.zeros(nbytes, device=dev, dtype=torch.uint8)
    gr = {k: torch.zeros(s, device=dev) for k, s in dict(x0=(B, C, H, W), goal=(B, gch, H, W), wp=(3 * C, 9), w1=(hid, 3 * C), b1=(hid,), w2=(hid, hid), b2=(hid,), w3=(C, hid)).items()}
    ud, gd, cd = u.to(dev), goal.to(dev), cot.to(dev)
    check(lib().ncahip_cond_grow_bwd_f32(_p(states), _p(prem), 1, _p(gd), gch, _p(ud), _p(w.wp), _p(w.w1), _p(w.b1), _p(w.w2), _p(w.b2), _p(w.w3),
                                         B, C, H, W, hid, alive, 0.1, 0.5, -10.0, 10.0, 0, 0, _p(cd), _p(gr["x0"]), _p(gr["goal"]), _p(gr["wp"]), _p(gr["w1"]),
                                         _p(gr["b1"]), _p(gr["w2"]), _p(gr["b2"]), _p(gr["w3"]), _p(ws), nbytes, _stream()), "bwd")
    torch.cuda.synchronize()
    n = B * C * H * W
    al = lambda v: (v + 255) & ~255
    f = ws.view(torch.float32) if nbytes % 4 == 0 else None
    o = 2 * al(n * 4) // 4
    gx = f[o:o + n].view(B, C, H, W); o += al(n * 4) // 4
    zb = f[o:o + n].view(B, C, H, W); o += al(n * 4) // 4
    dP = f[o:o + 3 * n].view(B, 3 * C, H, W)
    err("z", zb, z)
    err("gx(dx1)", gx, x1.grad)
    err("dP", dP, pp.grad)
    err("dx0", gr["x0"], xr.grad)
    err("dgoal", gr["goal"], g.grad[:, C - gch:])
    err("dwp", gr["wp"].view(3 * C, 1, 3, 3), p["perception_net.weight"].grad)
    err("dw1", gr["w1"], p["update_net.out.0.weight"].grad[:, :, 0, 0])
    err("db1", gr["b1"], p["update_net.out.0.bias"].grad)
    err("dw2", gr["w2"], p["update_net.out.2.weight"].grad[:, :, 0, 0])
    err("db2", gr["b2"], p["update_net.out.2.bias"].grad)
    err("dw3", gr["w3"], p["update_net.out.4.weight"].grad[:, :, 0, 0])

run(16, 1, 16, 16, 16, -1)
run(16, 2, 32, 48, 12, 3)
run(12, 1, 20, 36, 8, 3)
