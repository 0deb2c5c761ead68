"""bf16 state storage: ncahip_cond_{step_fwd,finalize,grow_fwd}_bf16 against the oracle's cond_step_bf16 (the rounding
points of include/ncahip.h restated on the CPU) and against the fp32 oracle.  The reference is fp32 only, so there is no
golden fixture for this path ("parity unpinned" by reference data); it is pinned to the fp32 step through the error
bound below.  Tolerances: bf16 has 8 significand bits; GPU and oracle differ only in f32 accumulation order, which can
flip the final rounding of an element by one bf16 ulp (2^-8 relative)."""
import pytest
import torch

from oracle import nca_oracle as O
from util import load, T

pytestmark = pytest.mark.gpu
DEV = "cuda"
ULP = 2.0 ** -8


def bfr(t):
    return t.to(torch.bfloat16).to(torch.float32)


def make_case(C, B, H, W, gch, seed=0, hidden=64):
    g = torch.Generator().manual_seed(seed)
    prm = {"perception_net.weight": (torch.rand(3 * C, 1, 3, 3, generator=g) - 0.5) * 0.6,
           "update_net.out.0.weight": torch.randn(hidden, 3 * C, 1, 1, generator=g) * 0.15,
           "update_net.out.0.bias": torch.randn(hidden, generator=g) * 0.05,
           "update_net.out.2.weight": torch.randn(hidden, hidden, 1, 1, generator=g) * 0.12,
           "update_net.out.2.bias": torch.randn(hidden, generator=g) * 0.05,
           "update_net.out.4.weight": torch.randn(C, hidden, 1, 1, generator=g) * 0.1}
    x = bfr(torch.rand(B, C, H, W, generator=g) * 1.2 - 0.2)
    x[:, 3] = bfr(torch.rand(B, H, W, generator=g) * 0.5)          # alpha straddles the 0.1 threshold
    goal = bfr(torch.randn(B, gch, H, W, generator=g) * 0.5) if gch else None
    u = torch.rand(B, 1, H, W, generator=g)
    return prm, x, goal, u


def weights(ops, prm, like):
    return ops.CondWeights(prm["perception_net.weight"], prm["update_net.out.0.weight"], prm["update_net.out.0.bias"],
                           prm["update_net.out.2.weight"], prm["update_net.out.2.bias"], prm["update_net.out.4.weight"], like)


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    from ncahip import ops as _ops
    _ops.selftest()
    return _ops


@pytest.mark.parametrize("C,shape,gch", [(16, (2, 32, 48), 12), (12, (2, 24, 32), 8), (16, (1, 8, 16), 12), (16, (1, 5, 20), 0),
                                         (10, (2, 16, 16), 6),
                                         # the reference's default C = 20 and a width below the padded 20 (wide LDS carve, forward only)
                                         (20, (2, 32, 48), 16), (20, (1, 40, 64), 16), (18, (1, 20, 24), 14)])
def test_single_step_vs_bf16_oracle(ops, C, shape, gch):
    B, H, W = shape
    prm, x, goal, u = make_case(C, B, H, W, gch)
    gpad = O.cond_pad_goal(goal, C) if goal is not None else torch.zeros_like(x)
    ref_next, ref_pre, ref_pend = O.cond_step_bf16(x, gpad, u, prm)
    xd = x.to(DEV).bfloat16()
    w = weights(ops, prm, xd)
    xp, pre = ops.cond_step(xd, None, None if goal is None else goal.to(DEV).bfloat16(), u.to(DEV), w, 3)
    assert xp.dtype == torch.bfloat16
    assert torch.equal(pre.cpu().bool().reshape(B, 1, H, W), ref_pre)            # computed from the input state: exact
    got = xp.float().cpu()
    err = (got - ref_pend).abs() / ref_pend.abs().clamp_min(1.0)
    assert float(err.max()) <= 2 * ULP, float(err.max())                        # at most a rounding flip
    assert float((got != ref_pend).float().mean()) < 0.03                       # and only on a few elements
    # resolved state: compare where the pending alpha is not within one ulp of the threshold
    res = ops.cond_finalize(xp, pre, 3).float().cpu()
    pooled = torch.nn.functional.max_pool2d(ref_pend[:, 3:4], 3, 1, 1)
    safe = ((pooled - 0.1).abs() > 2 * ULP).expand_as(res)
    e2 = ((res - ref_next).abs() / ref_next.abs().clamp_min(1.0))[safe]
    assert float(e2.max()) <= 2 * ULP


def test_close_to_fp32_step(ops):
    """One bf16 step stays within bf16 resolution of the exact fp32 step (nca.py:181-195) on the same inputs."""
    C, B, H, W, gch = 16, 2, 32, 32, 12
    prm, x, goal, u = make_case(C, B, H, W, gch, seed=3)
    gpad = O.cond_pad_goal(goal, C)
    d = O.cond_step(x, gpad, u, prm, 3, return_all=True)
    xd = x.to(DEV).bfloat16()
    xp, pre = ops.cond_step(xd, None, goal.to(DEV).bfloat16(), u.to(DEV), weights(ops, prm, xd), 3)
    err = (xp.float().cpu() - d["x1"]).abs() / d["x1"].abs().clamp_min(1.0)
    assert float(err.max()) < 3e-2 and float(err.mean()) < 3e-3, (float(err.max()), float(err.mean()))


def test_grow_free_running_vs_oracle(ops):
    C, B, H, W, gch, T_ = 16, 2, 32, 32, 12, 8
    prm, x, goal, _ = make_case(C, B, H, W, gch, seed=5)
    g = torch.Generator().manual_seed(11)
    us = torch.rand(T_, B, 1, H, W, generator=g)
    gpad = O.cond_pad_goal(goal, C)
    ref = x
    for t in range(T_):
        ref, _, _ = O.cond_step_bf16(ref, gpad, us[t], prm)
    xd = x.to(DEV).bfloat16()
    out, _, _ = ops.cond_grow(xd, T_, goal.to(DEV).bfloat16(), us.to(DEV), weights(ops, prm, xd), 3)
    got = out.float().cpu()
    err = (got - ref).abs() / ref.abs().clamp_min(1.0)
    # rounding flips feed back through 8 steps and through the life threshold: bounded drift, few outliers
    assert float((err > 8 * ULP).float().mean()) < 0.01, float((err > 8 * ULP).float().mean())
    assert float(err.mean()) < 2e-3


def test_properties(ops):
    C, B, H, W, gch = 16, 2, 16, 32, 12
    prm, x, goal, u = make_case(C, B, H, W, gch, seed=7)
    xd, gd = x.to(DEV).bfloat16(), goal.to(DEV).bfloat16()
    w = weights(ops, prm, xd)
    # fire rate 0: the state only passes the life mask / clamp
    xp, pre = ops.cond_step(xd, None, gd, u.to(DEV), w, 3, fire_rate=0.0)
    assert torch.equal(xp, xd)
    # in-kernel Philox == explicit uniforms drawn by the library's generator, bit for bit
    uu = ops.philox_uniform(B, H, W, seed=9, step=4, device=DEV)
    a1, p1 = ops.cond_step(xd, None, gd, uu, w, 3)
    a2, p2 = ops.cond_step(xd, None, gd, None, w, 3, seed=9, step=4)
    assert torch.equal(a1, a2) and torch.equal(p1, p2)
    # a dead grid stays dead
    dead = torch.zeros_like(xd)
    out, _, _ = ops.cond_grow(dead, 3, gd, None, w, 3, seed=1)
    assert float(out.float().abs().max()) == 0.0
    # batch independence
    b0, _ = ops.cond_step(xd[:1].contiguous(), None, gd[:1].contiguous(), u[:1].to(DEV), w, 3)
    full, _ = ops.cond_step(xd, None, gd, u.to(DEV), w, 3)
    assert torch.equal(b0, full[:1])


def test_refuses_unaligned_width(ops):
    from ncahip._capi import NcaHipError
    prm, x, goal, u = make_case(16, 1, 6, 10, 12)
    xd = x.to(DEV).bfloat16()
    with pytest.raises(NcaHipError):
        ops.cond_step(xd, None, goal.to(DEV).bfloat16(), u.to(DEV), weights(ops, prm, xd), 3)


@pytest.mark.parametrize("hidden_ch", [12, 16])
def test_module_grow_bf16(ops, hidden_ch, monkeypatch):
    """ConditionedNCA.grow on a bf16 state (no_grad): same class surface, bf16 in / bf16 out, close to the fp32 grow.
    hidden_ch = 16 is the reference's default model (C = 20): its no_grad grow runs the bf16-storage kernels too, with
    gradients enabled it steps in fp32 and returns the pool's dtype."""
    from ncahip.nca import ConditionedNCA
    torch.manual_seed(0)
    m = ConditionedNCA(target_shape=(3, 32, 32), num_hidden_channels=hidden_ch, living_channel_dim=3).to(DEV)
    seen = []
    real = ops.cond_grow
    monkeypatch.setattr(ops, "cond_grow", lambda x, *a, **k: (seen.append(x.dtype), real(x, *a, **k))[1])
    with torch.no_grad():
        for p_ in m.update_net.parameters():
            p_.add_(torch.randn_like(p_) * 0.05)
    m.mask_rng = "philox"
    x = (torch.rand(2, m.num_channels, 32, 32, device=DEV) * 0.8).bfloat16()
    goal = torch.rand(2, 3, 32, 32, device=DEV)
    with torch.no_grad():
        m._mask_step = 0
        yb = m.grow(x, 4, goal)
        m._mask_step = 0
        yf = m.grow(x.float(), 4, goal)
    assert yb.dtype == torch.bfloat16 and yb.shape == yf.shape
    assert seen == [torch.bfloat16, torch.float32], seen          # the bf16 grow really ran on the bf16-storage kernels
    hist = []
    real_bwd = ops.cond_grow_backward
    monkeypatch.setattr(ops, "cond_grow_backward", lambda st, *a, **k: (hist.append(st.dtype), real_bwd(st, *a, **k))[1])
    d = (yb.float() - yf).abs()
    assert float(d.mean()) < 2e-2, float(d.mean())
    # autograd through the bf16 path: bf16 history ring, fp32 gradients for every parameter (encoder included)
    m._mask_step = 0
    out = m.grow(x.clone().requires_grad_(True), 3, goal)
    assert out.dtype == torch.bfloat16
    out.float().square().mean().backward()
    assert hist == [torch.bfloat16], hist                          # bf16 history ring in the backward, default model included
    for n_, p_ in m.named_parameters():
        if p_.requires_grad:
            assert p_.grad is not None and p_.grad.dtype == torch.float32 and bool(torch.isfinite(p_.grad).all()), n_


@pytest.mark.parametrize("C,shape,gch,Tn", [(16, (2, 32, 48), 12, 4), (12, (2, 24, 32), 8, 3), (16, (1, 16, 16), 16, 2),
                                            (20, (2, 32, 48), 16, 3), (18, (1, 16, 32), 14, 2)])
def test_cond_grow_backward_bf16_history(ops, C, shape, gch, Tn):
    """ncahip_cond_grow_bwd_bf16 (bf16 history, matrix products on bf16 MFMA, everything else fp32; C = 20 / 18: the reference's
    default model on the front + matrix kernels at CP = 20).  Checked three ways:
    (i) the storage plumbing: with the exact-f32 product hook the bf16-history kernel equals the fp32 kernel fed the widened
    history bit for bit; (ii) the bf16-MFMA products against those exact-f32 products of the SAME history (relative L2);
    (iii) against oracle autograd through the fp32 steps started from the same bf16-representable inputs (different
    trajectory: bf16 storage + bf16 matrix operands in the forward), within the bf16 budget.  The alpha channel is held
    fixed (zero output row) so no life mask sits near its threshold and the comparisons are smooth."""
    B, H, W = shape
    gen = torch.Generator().manual_seed(C + Tn)
    from test_gpu_parity import rand_cond_prm
    prm = rand_cond_prm(C, seed=C + 1, out_scale=1.0)
    prm["update_net.out.4.weight"][3] = 0.0                         # alpha never changes
    x0 = bfr(torch.rand(B, C, H, W, generator=gen))
    x0[:, 3] = bfr(0.5 + 0.5 * torch.rand(B, H, W, generator=gen))
    x0[0, :, : H // 3] = 0.0                                        # a dead band with a live frontier
    goal = bfr(torch.randn(B, gch, H, W, generator=gen) * 0.5)
    us = torch.rand(Tn, B, 1, H, W, generator=gen)
    cot = torch.randn(B, C, H, W, generator=gen)
    xd, gd, ud, cd = x0.to(DEV).bfloat16(), goal.to(DEV).bfloat16(), us.to(DEV), cot.to(DEV)
    w = weights(ops, prm, xd)
    out, states, pre = ops.cond_grow(xd, Tn, gd, ud, w, 3, keep_history=True)
    assert states.dtype == torch.bfloat16
    g16 = ops.cond_grow_backward(states, pre, gd, ud, w, cd, Tn, 3)
    g32 = ops.cond_grow_backward(states.float(), pre, gd.float(), ud, w, cd, Tn, 3)
    ops.force_generic(4)
    try:
        g16x = ops.cond_grow_backward(states, pre, gd, ud, w, cd, Tn, 3)
    finally:
        ops.force_generic(0)
    rel2 = lambda a, b: float((a.cpu().double().reshape(-1) - b.cpu().double().reshape(-1)).norm() / b.double().norm().clamp_min(1e-12))
    names = {"wp": "perception_net.weight", "w1": "update_net.out.0.weight", "b1": "update_net.out.0.bias",
             "w2": "update_net.out.2.weight", "b2": "update_net.out.2.bias", "w3": "update_net.out.4.weight"}
    for k in g16:
        assert g16x[k].dtype == torch.float32 and torch.equal(g16x[k], g32[k]), k          # (i) storage plumbing, bit for bit
        # (ii) bf16-MFMA products vs exact-f32 products of the same history.  The two evaluate ReLU' on different
        # pre-activations (bf16-operand vs exact recomputation): a fraction p of hidden units near zero flips and moves the
        # relative L2 distance by ~sqrt(p) -- a few per cent, the price of gates that are consistent with the bf16 forward
        assert g16[k].dtype == torch.float32 and rel2(g16[k], g32[k]) < 8e-2, (k, rel2(g16[k], g32[k]))
    # (iii) the function the kernel differentiates: oracle autograd through the bf16-faithful steps (straight-through
    # roundings).  Same gates, same operands; the kernel additionally rounds the gradient operands of its products to bf16.
    _, gx, gg, gw = O.cond_grow_bf16_loss_grads(x0, O.cond_pad_goal(goal, C), list(us), prm, 3, 0.1, 0.5, cot)
    tol = 2e-2 if Tn <= 2 else 4e-2          # longer roll-outs: the two trajectories drift apart by final-rounding flips
    assert rel2(g16["x0"], gx) < tol and rel2(g16["goal"], gg[:, C - gch:]) < tol, (rel2(g16["x0"], gx), rel2(g16["goal"], gg[:, C - gch:]))
    for k, n in names.items():
        assert rel2(g16[k], gw[n]) < tol, (k, rel2(g16[k], gw[n]))
    # (iv) and the fp32 oracle (exact step function, fp32 trajectory from the same inputs): the bf16 budget end to end
    _, gx, gg, gw = O.cond_grow_loss_grads(x0, O.cond_pad_goal(goal, C), list(us), prm, 3, 0.1, 0.5, cot)
    assert rel2(g16["x0"], gx) < 8e-2 and rel2(g16["goal"], gg[:, C - gch:]) < 8e-2
    for k, n in names.items():
        assert rel2(g16[k], gw[n]) < 8e-2, (k, rel2(g16[k], gw[n]))


@pytest.mark.parametrize("C,fc,cc,pad,shape", [(12, 96, 3, "circular", (2, 24, 32)), (16, 128, 3, "replicate", (1, 16, 48)),
                                               (12, 96, 0, "reflect", (2, 9, 11)), (16, 128, 2, "constant", (1, 8, 36))])
def test_dynca_bf16_storage(ops, C, fc, cc, pad, shape):
    """ncahip_dynca_nsteps_fwd_bf16: exact fp32 step on the widened state, rounded to bf16 on store.  Per step the GPU and
    the oracle (dynca.py:117-138 restated, then .bfloat16()) can differ by one final-rounding flip."""
    B, H, W = shape
    g = torch.Generator().manual_seed(C + fc + cc)
    k1 = 4 * C + cc
    prm = {"w1.weight": torch.randn(fc, k1, 1, 1, generator=g) * (0.5 / k1 ** 0.5), "w1.bias": torch.randn(fc, generator=g) * 0.1,
           "w2.weight": torch.randn(C, fc, 1, 1, generator=g) * (0.3 / fc ** 0.5), "w2.bias": torch.randn(C, generator=g) * 0.02}
    x = bfr(torch.rand(B, C, H, W, generator=g) - 0.5)
    cond = (torch.rand(B, cc, H, W, generator=g) * 2 - 1) if cc else None
    T_ = 3
    us = torch.rand(T_, B, 1, H, W, generator=g)
    w = ops.DyncaWeights(prm["w1.weight"], prm["w1.bias"], prm["w2.weight"], prm["w2.bias"], x.to(DEV))
    # teacher-forced single steps
    cur = x
    for t in range(T_):
        ref = bfr(O.dynca_step(cur, cond, us[t], prm, pad, 0.5))
        got, _ = ops.dynca_nsteps(cur.to(DEV).bfloat16(), 1, None if cond is None else cond.to(DEV), us[t:t + 1].to(DEV), w, pad, 0.5)
        assert got.dtype == torch.bfloat16
        gf = got.float().cpu()
        err = (gf - ref).abs() / ref.abs().clamp_min(1.0)
        assert float(err.max()) <= 2 * ULP and float((gf != ref).float().mean()) < 0.01
        cur = ref
    # free-running T steps in one call, in-kernel Philox == explicit uniforms
    uu = torch.stack([ops.philox_uniform(B, H, W, seed=3, step=t, device=DEV) for t in range(T_)])
    a1, _ = ops.dynca_nsteps(x.to(DEV).bfloat16(), T_, None if cond is None else cond.to(DEV), uu, w, pad, 0.5)
    a2, _ = ops.dynca_nsteps(x.to(DEV).bfloat16(), T_, None if cond is None else cond.to(DEV), None, w, pad, 0.5, seed=3, step0=0)
    assert torch.equal(a1, a2)


def test_dynca_module_bf16(ops):
    from ncahip.models.dynca import DyNCA
    torch.manual_seed(0)
    m = DyNCA(12, 3, fc_dim=96, padding_mode="circular", conditioning="edges", device=torch.device(DEV))
    m.mask_rng = "philox"
    x = (torch.rand(2, 12, 32, 32, device=DEV) - 0.5).bfloat16()
    img = torch.rand(2, 1, 32, 32, device=DEV) * 2 - 1
    with torch.no_grad():
        m._mask_step = 0
        yb, rgb = m.forward_nsteps(x, 4, cond_img=img)
        m._mask_step = 0
        yf, _ = m.forward_nsteps(x.float(), 4, cond_img=img)
    assert yb.dtype == torch.bfloat16 and rgb.shape == (2, 3, 32, 32)
    assert float((yb.float() - yf).abs().mean()) < 1e-2
    # autograd over the bf16 history (ncahip_dynca_nsteps_bwd_bf16): storage format only, so the gradients equal the fp32
    # backward fed the widened history bit for bit, and match oracle autograd through straight-through bf16 stores
    from oracle import nca_oracle as O2
    prm = {k: v.detach().cpu().clone() for k, v in m.state_dict().items() if k.startswith(("w1", "w2"))}
    us = torch.rand(3, 2, 1, 32, 32, device=DEV)
    it = iter(us)
    m._draw = lambda x_, steps, rate=None: torch.stack([next(it) for _ in range(steps)])
    xg = x.clone().requires_grad_(True)
    cot = torch.randn(2, 12, 32, 32, device=DEV)
    out, _ = m.forward_nsteps(xg, 3, cond_img=img)
    assert out.dtype == torch.bfloat16
    (out.float() * cot).sum().backward()
    assert xg.grad.dtype == torch.bfloat16
    w = ops.DyncaWeights(m.w1.weight, m.w1.bias, m.w2.weight, m.w2.bias, x)
    cond = m._cond(x, img)
    _, st16 = ops.dynca_nsteps(x, 3, cond, us, w, "circular", 0.5, keep_history=True)
    g16 = ops.dynca_nsteps_backward(st16, cond, us, w, cot, None, 3, "circular", 0.5)
    g32 = ops.dynca_nsteps_backward(st16.float(), cond, us, w, cot, None, 3, "circular", 0.5)
    for k in g16:
        assert torch.equal(g16[k], g32[k]), k
    gmod = ops.dynca_nsteps_backward(st16, cond, us, w, cot.bfloat16().float(), None, 3, "circular", 0.5)   # autograd hands the bf16 output a bf16 cotangent
    assert torch.equal(m.w1.weight.grad[:, :, 0, 0], gmod["w1"]) and torch.equal(m.w2.bias.grad, gmod["b2"])
    # oracle: x <- bf16(x + dx * m) with a straight-through store
    xr = x.float().cpu().clone().requires_grad_(True)
    p = {k: v.clone().requires_grad_(True) for k, v in prm.items()}
    cur, cnd = xr, cond.cpu()
    for t in range(3):
        nxt = O2.dynca_step(cur, cnd, us[t].cpu(), p, "circular", 0.5)
        cur = nxt + (nxt.to(torch.bfloat16).float() - nxt).detach()
    (cur * cot.cpu()).sum().backward()
    rel2 = lambda a, b: float((a.double().cpu().reshape(-1) - b.double().reshape(-1)).norm() / b.double().norm().clamp_min(1e-12))
    assert rel2(g16["x0"], xr.grad) < 1e-2 and rel2(g16["w1"], p["w1.weight"].grad[:, :, 0, 0]) < 1e-2
    assert rel2(g16["w2"], p["w2.weight"].grad[:, :, 0, 0]) < 1e-2


# ------------------------------------------------------------------------------------------------ opt-in bf16x3 products
@pytest.mark.parametrize("C,shape,gch", [(16, (2, 32, 48), 12), (12, (2, 24, 32), 8)])
def test_bf16x3_precision_mode(ops, C, shape, gch):
    """ncahip_cond_precision(1): fp32 storage, every UpdateNet product from three bf16 MFMAs.  Must stay inside the 1e-4
    parity bar against the fp32 oracle (nca.py:181-195) on a teacher-forced step, and well above the exact mode's error."""
    from util import REL_TOL
    B, H, W = shape
    g = torch.Generator().manual_seed(1)
    prm, x, goal, u = make_case(C, B, H, W, gch, seed=2)
    x = torch.rand(B, C, H, W, generator=g) * 1.2 - 0.2          # full fp32 values
    x[:, 3] = torch.rand(B, H, W, generator=g) * 0.5
    goal = torch.randn(B, gch, H, W, generator=g) * 0.5
    gpad = O.cond_pad_goal(goal, C)
    d = O.cond_step(x, gpad, u, prm, 3, return_all=True)
    xd = x.to(DEV)
    w = weights(ops, prm, xd)
    try:
        ops.set_cond_precision("bf16x3")
        xs, pre_s = ops.cond_step(xd, None, goal.to(DEV), u.to(DEV), w, 3)
    finally:
        ops.set_cond_precision("exact")
    xe, pre_e = ops.cond_step(xd, None, goal.to(DEV), u.to(DEV), w, 3)
    assert torch.equal(pre_s, pre_e)
    ref = d["x1"]
    es = float(((xs.cpu() - ref).abs() / ref.abs().clamp_min(1.0)).max())
    ee = float(((xe.cpu() - ref).abs() / ref.abs().clamp_min(1.0)).max())
    assert es < REL_TOL, es                      # the parity bar
    assert ee < 5e-6 and es > ee                 # and it is NOT the exact path
    with pytest.raises(Exception):
        ops.set_cond_precision(7)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,shape,gch,Tn", [(16, (2, 80, 112), 12, 3), (12, (1, 36, 52), 8, 2)])
def test_cond_backward_kernel_forms_agree(ops, dtype, C, shape, gch, Tn):
    """Backward kernel A exists in two forms (csrc/nca_cond_bwd.hip: ONE launch, everything for a tile in one wave;
    csrc/nca_cond_bwd_fm.hip: a front kernel -- staging, gate, perception into an operand-order scratch -- and a matrix
    kernel).  With fp32 products both issue the same products in the same per-wave order, so every gradient must agree BIT FOR
    BIT -- on evolving life masks, image borders (ragged sizes) included; the bf16-MFMA matrix kernel sums in another fixed
    order (two waves per SIMD), so there the bound is 2e-5 of the largest entry.  Each form is checked against the oracle through the
    kernel-family fixture of test_gpu_parity.py (the default form per mode and, in the third family, the other one); this
    pins them to each other, for the fp32 products and for the bf16-MFMA products."""
    B, H, W = shape
    gen = torch.Generator().manual_seed(7 * C + Tn)
    from test_gpu_parity import rand_cond_prm
    prm = rand_cond_prm(C, seed=C + 5, out_scale=1.0)
    x0 = bfr(torch.rand(B, C, H, W, generator=gen))
    x0[0, :, : H // 3] = 0.0
    goal = bfr(torch.randn(B, gch, H, W, generator=gen) * 0.5)
    us = torch.rand(Tn, B, 1, H, W, generator=gen)
    cot = torch.randn(B, C, H, W, generator=gen)
    xd, gd, ud, cd = x0.to(DEV).to(dtype), goal.to(DEV).to(dtype), us.to(DEV), cot.to(DEV)
    w = weights(ops, prm, xd)
    _, states, pre = ops.cond_grow(xd, Tn, gd, ud, w, 3, keep_history=True)
    res = []
    try:
        # bit 3 (8) = the form that is not the default, bit 4 (16) = the matrix kernel walks whole super-tiles.  On these small grids the
        # default is front + matrix with every super-tile split over two workgroups, so: 16 = front + matrix unsplit, 8 | 16 = one launch,
        # 0 = front + matrix split
        for form in (8 | 16, 16, 0):
            ops.force_generic(form)
            res.append(ops.cond_grow_backward(states, pre, gd, ud, w, cd, Tn, 3))
            ops.check_errors()
    finally:
        ops.force_generic(0)
    one, two, split = res
    for k in one:
        # the split: same products, the partial sums of a super-tile meet in another order (two slabs)
        ds = float((split[k] - two[k]).abs().max()) / max(1e-12, float(two[k].abs().max()))
        assert ds <= 2e-6, (k, ds)
        if dtype == torch.float32:
            assert torch.equal(one[k], two[k]), k
        else:
            # the bf16-MFMA matrix kernel runs two waves per SIMD and merges the pair's partial accumulators at the flush (and takes
            # its ReLU gates from the stored bf16 activations): same products, another fixed summation order
            d = float((one[k] - two[k]).abs().max()) / max(1e-12, float(one[k].abs().max()))
            assert d <= 2e-5, (k, d)


def test_cond_backward_forms_fuzz(ops):
    """Seeded shape / option fuzz of the two forms of backward kernel A against each other (fp32 products: bit for bit; bf16
    MFMA: 2e-5 of the largest entry): ragged heights, widths that are multiples of 4 only, C in 9..16 (padded channel counts),
    goal widths, alive channel on / off, explicit uniforms or in-kernel Philox, 1..4 steps.  NCAHIP_FUZZ_SEED / _CASES widen it."""
    import os
    import numpy as np
    from test_gpu_parity import rand_cond_prm
    rng = np.random.RandomState(int(os.environ.get("NCAHIP_FUZZ_SEED", "4242")))
    for case in range(int(os.environ.get("NCAHIP_FUZZ_CASES", "10"))):
        C = int(rng.randint(9, 17))
        gch = int(rng.randint(0, C - 3))
        B, H, W, Tn = int(rng.randint(1, 4)), int(rng.randint(5, 50)), 4 * int(rng.randint(2, 14)), int(rng.randint(1, 5))
        ach = 3 if rng.rand() < 0.8 else -1
        dtype = torch.bfloat16 if rng.rand() < 0.6 else torch.float32
        philox = rng.rand() < 0.5
        gen = torch.Generator().manual_seed(1000 + case)
        prm = rand_cond_prm(C, seed=50 + case, out_scale=1.0)
        x0 = bfr(torch.rand(B, C, H, W, generator=gen))
        x0[0, :, : H // 3] = 0.0
        goal = bfr(torch.randn(B, gch, H, W, generator=gen) * 0.5) if gch else None
        us = None if philox else torch.rand(Tn, B, 1, H, W, generator=gen).to(DEV)
        cot = torch.randn(B, C, H, W, generator=gen).to(DEV)
        xd = x0.to(DEV).to(dtype)
        gd = goal.to(DEV).to(dtype) if gch else None
        w = weights(ops, prm, xd)
        _, states, pre = ops.cond_grow(xd, Tn, gd, us, w, ach, seed=case, keep_history=True)
        res = []
        try:
            for form in (16, 8 | 16):     # the two forms with whole super-tiles per workgroup (the split of small grids: test_cond_backward_kernel_forms_agree)
                ops.force_generic(form)
                res.append(ops.cond_grow_backward(states, pre, gd, us, w, cot, Tn, ach, seed=case))
                ops.check_errors()
        finally:
            ops.force_generic(0)
        one, two = res
        tag = (case, C, gch, B, H, W, Tn, ach, str(dtype), philox)
        for k in one:
            if one[k] is None:
                assert two[k] is None
                continue
            if dtype == torch.float32:
                assert torch.equal(one[k], two[k]), (k, tag)
            else:
                d = float((one[k] - two[k]).abs().max()) / max(1e-12, float(one[k].abs().max()))
                assert d <= 2e-5, (k, d, tag)
