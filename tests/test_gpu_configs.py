"""Every BASELINE.json config through the HIP path (the C ABI), at the config's own size.

  configs[0]  B=4  C=12 128^2  T=32          the committed G2L fixture (reference-generated) replayed on the GPU
  configs[1]  B=8  C=16 256^2  T=64  fp32    64 teacher-forced steps along the oracle trajectory + free-running bound
  configs[2]  B=32 C=16 256^2  fwd + bwd     fp32 and bf16 history: live patches against oracle autograd on the crops,
                                             batch independence, T=96 smoke of the whole trainer-shaped call
  configs[4]  DyNCA C=32 fc=256 512^2        circular tiling of a 64^2 problem against the oracle on the tile, fwd + bwd
  (configs[3] is configs[2]'s shape sharded over 8 GPUs: the per-rank work is exactly the configs[2] case; the exchange is
   covered by tests/test_dist_gloo.py)

Beyond the oracle's reach (2048^2 planes): size-independent properties, promoted from tools/large_shape_check.py.
"""
import numpy as np
import pytest
import torch

from oracle import nca_oracle as O
from util import GATE_K, REL_TOL, T, grads_match_outside, load, rel_err, sd

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    from ncahip import ops as _ops
    _ops.selftest()
    _ops.force_generic(False)
    return _ops


def cond_w(ops, prm, like):
    return ops.CondWeights(prm["perception_net.weight"], prm["update_net.out.0.weight"], prm["update_net.out.0.bias"],
                           prm["update_net.out.2.weight"], prm["update_net.out.2.bias"],
                           prm["update_net.out.4.weight"], like)


def rand_cond_prm(C, seed, hidden=64, out_scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return {"perception_net.weight": torch.randn(3 * C, 1, 3, 3, generator=g) * 0.3,
            "update_net.out.0.weight": torch.randn(hidden, 3 * C, 1, 1, generator=g) * (1.0 / (3 * C) ** 0.5),
            "update_net.out.0.bias": torch.randn(hidden, generator=g) * 0.1,
            "update_net.out.2.weight": torch.randn(hidden, hidden, 1, 1, generator=g) * (1.0 / hidden ** 0.5),
            "update_net.out.2.bias": torch.randn(hidden, generator=g) * 0.1,
            "update_net.out.4.weight": torch.randn(C, hidden, 1, 1, generator=g) * (out_scale * 0.3 / hidden ** 0.5)}


def rand_dynca_prm(C, fc, c_cond, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    k1 = 4 * C + c_cond
    return {"w1.weight": torch.randn(fc, k1, 1, 1, generator=g) * (0.5 / k1 ** 0.5),
            "w1.bias": torch.randn(fc, generator=g) * 0.1,
            "w2.weight": torch.randn(C, fc, 1, 1, generator=g) * (scale * 0.3 / fc ** 0.5),
            "w2.bias": torch.randn(C, generator=g) * 0.02}


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max()) / max(1e-6, float(b.abs().max()))


def _rel2(a, b):
    """relative L2 error (the bf16 bound: one ReLU flipping between two trajectories moves a single cell by O(1))"""
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


def _near(x1, alive, thr=0.1, eps=2e-6):
    """cells whose pooled NEW alpha lies within eps of the threshold may legitimately resolve either way"""
    return ((torch.nn.functional.max_pool2d(x1[:, alive:alive + 1], 3, 1, 1) - thr).abs() < eps)


# ---------------------------------------------------------------------------------------------- configs[0]
@pytest.mark.parametrize("variant", [0, 1, 2])
def test_cfg1_g2l_fixture_through_hip(ops, variant):
    """BASELINE configs[0] exactly (B=4 C=12 128^2 T=32): the reference-generated fixture (state_dict, per-step sums and
    alive counts, final 16x16 crop) replayed through the HIP grow loop, every kernel family."""
    g = load("g2l_cfg1")
    prm = sd(g)
    torch.manual_seed(int(g["data_seed"]))
    x0 = torch.rand(4, 12, 128, 128)
    goal = torch.rand(4, 3, 128, 128)
    genc = O.image_encoder(goal, prm)                     # [4, 8, 128, 128], unpadded as the C ABI takes it
    Tn = int(g["T"])
    torch.manual_seed(int(g["rng_seed"]))
    us = torch.stack([torch.rand_like(x0[:, 0:1]) for _ in range(Tn)])   # the reference's own draws (CPU stream)
    ops.force_generic(variant)
    try:
        w = cond_w(ops, prm, x0.to(DEV))
        xT, states, pre = ops.cond_grow(x0.to(DEV), Tn, genc.to(DEV), us.to(DEV), w, 3, keep_history=True)
        sums, nal = [], []
        for t in range(1, Tn + 1):
            st = ops.cond_finalize(states[t], pre[t], 3)
            sums.append(float(st.double().sum()))
            nal.append(int(ops.cond_alive(st, 3).sum()))
    finally:
        ops.force_generic(False)
    ref_s, ref_n = np.asarray(g["sums"], dtype=np.float64), np.asarray(g["nalive"])
    ncell = 4 * 128 * 128
    # free-running: a near-threshold cell may flip (it then carries |x| <= 10 per channel into the sum)
    assert np.abs(np.asarray(nal) - ref_n).max() <= max(2, ncell // 20000), (nal, list(ref_n))
    assert np.abs(np.asarray(sums) - ref_s).max() <= 1e-4 * np.abs(ref_s).max() + 120.0
    crop = xT[:, :, 56:72, 56:72].cpu()
    bad = ((crop - T(g["crop"])).abs() > REL_TOL * max(1.0, float(np.abs(g["crop"]).max()))).float().mean()
    assert float(bad) < 0.01


# ---------------------------------------------------------------------------------------------- configs[1]
@pytest.fixture(scope="module")
def cfg2_case():
    """B=8 C=16 256^2 T=64 on the CPU oracle, once per module (about 20-40 s of host time)."""
    B, C, H, W, Tn = 8, 16, 256, 256, 64
    prm = rand_cond_prm(C, seed=0, out_scale=0.5)
    gen = torch.Generator().manual_seed(1234)
    x0 = torch.rand(B, C, H, W, generator=gen)
    goal = torch.randn(B, 12, H, W, generator=gen) * 0.5
    us = torch.rand(Tn, B, 1, H, W, generator=gen)
    gpad = O.cond_pad_goal(goal, C)
    refs, x1s, x = [], [], x0
    for t in range(Tn):
        d = O.cond_step(x, gpad, us[t], prm, 3, return_all=True)
        refs.append(d["x2"])
        x1s.append(_near(d["x1"], 3))
        x = d["x2"]
    return dict(prm=prm, x0=x0, goal=goal, us=us, refs=refs, near=x1s)


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_cfg2_64_teacher_forced_steps(ops, cfg2_case, variant):
    """BASELINE configs[1]: every one of the 64 steps, started from the oracle's state, within 1e-4 relative."""
    c = cfg2_case
    ops.force_generic(variant)
    try:
        x0d, goald = c["x0"].to(DEV), c["goal"].to(DEV)
        w = cond_w(ops, c["prm"], x0d)
        prev, worst, excl = x0d, 0.0, 0
        for t in range(64):
            xp, pre = ops.cond_step(prev, None, goald, c["us"][t].to(DEV), w, 3)
            x2 = ops.cond_finalize(xp, pre, 3).cpu()
            near = c["near"][t].expand_as(x2)
            excl += int(near.sum())
            worst = max(worst, rel_err(x2[~near], c["refs"][t][~near]))
            prev = c["refs"][t].to(DEV)
        assert worst < REL_TOL, worst
        assert excl < 64 * 16 * 50, excl          # the exclusion is a handful of cells per step, not a loophole
    finally:
        ops.force_generic(False)


def test_cfg2_free_running_64_steps(ops, cfg2_case):
    """The bench's own call (ncahip_cond_grow_fwd_f32, T=64) against the oracle's free-running end state."""
    c = cfg2_case
    x0d = c["x0"].to(DEV)
    w = cond_w(ops, c["prm"], x0d)
    got, _, _ = ops.cond_grow(x0d, 64, c["goal"].to(DEV), c["us"].to(DEV), w, 3)
    ref = c["refs"][-1]
    bad = ((got.cpu() - ref).abs() > REL_TOL * max(1.0, float(ref.abs().max()))).any(dim=1)
    assert float(bad.float().mean()) < 5e-3, float(bad.float().mean())


# ---------------------------------------------------------------------------------------------- configs[2]
def _patch_case(B, C, S, P, crop, seed):
    """Zero grids with one live P x P patch each (random position, away from the borders by more than the crop margin).
    Returns full tensors and the per-item crop windows."""
    gen = torch.Generator().manual_seed(seed)
    x = torch.zeros(B, C, S, S)
    goal = torch.randn(B, C - 4, S, S, generator=gen) * 0.5
    cot = torch.randn(B, C, S, S, generator=gen)
    m = (crop - P) // 2
    oy = torch.randint(m, S - P - m, (B,), generator=gen)
    ox = torch.randint(m, S - P - m, (B,), generator=gen)
    for b in range(B):
        x[b, :, oy[b]:oy[b] + P, ox[b]:ox[b] + P] = torch.rand(C, P, P, generator=gen)
    wins = [(int(oy[b]) - m, int(ox[b]) - m) for b in range(B)]
    return x, goal, cot, wins


@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_cfg3_forward_backward_vs_oracle_on_crops(ops, storage):
    """BASELINE configs[2]'s shape (B=32 C=16 256^2) through grow + its backward.  Each item holds one live 40x40 patch in a
    dead image; dead cells stay dead and carry no gradient, so item b must equal the oracle's autograd on the 64x64 crop
    around its patch (zero padding == dead cells), and the weight gradients the sum over the 32 crops."""
    B, C, S, P, CR, Tn = 32, 16, 256, 40, 64, 6
    prm = rand_cond_prm(C, seed=2, out_scale=1.0)
    x, goal, cot, wins = _patch_case(B, C, S, P, CR, seed=9)
    us = torch.rand(Tn, B, 1, S, S, generator=torch.Generator().manual_seed(10))
    bf = storage == "bf16"
    # Gradients through 6 FREE-RUNNING steps are compared, so no life mask may sit within rounding distance of its threshold
    # (a mask that resolves differently on the two sides is a legitimate O(1) difference: with evolving alpha this test flipped
    # with the host's thread count, i.e. with the ORACLE's summation order).  Hold the alpha channel fixed: zero output row,
    # alpha in {0} u [0.5, 1].  The live region is then each patch plus its one-cell frontier, for every step; evolving masks are
    # covered strictly by the teacher-forced tests (cfg2 above, G1/G2/G8 goldens) and by the fuzz tests at small shapes.
    prm["update_net.out.4.weight"][3] = 0.0
    live = (x[:, 3:4] > 0).float()
    x[:, 3:4] = live * (0.5 + 0.5 * x[:, 3:4])
    if bf:
        x, goal = x.bfloat16().float(), goal.bfloat16().float()
    # oracle on the crops
    regions, n_amb = [], []
    gx_ref = torch.zeros(B, C, CR, CR)
    gg_ref = torch.zeros(B, C - 4, CR, CR)
    out_ref = torch.zeros(B, C, CR, CR)
    wsum = None
    for b in range(B):
        y0, x0_ = wins[b]
        sl = (slice(b, b + 1), slice(None), slice(y0, y0 + CR), slice(x0_, x0_ + CR))
        xT, gx0, ggoal, gw = O.cond_grow_loss_grads(x[sl], O.cond_pad_goal(goal[sl], C), [u[sl[0], :, sl[2], sl[3]] for u in us],
                                                    prm, 3, 0.1, 0.5, cot[sl])
        out_ref[b], gx_ref[b], gg_ref[b] = xT[0], gx0[0], ggoal[0, 4:]
        wsum = gw if wsum is None else {k: wsum[k] + v for k, v in gw.items()}
        if not bf:   # proof hook: where the oracle's own near-zero ReLU gates can move this item's gradients (util.GATE_K)
            reg, cnt = O.cond_gate_influence(x[sl], O.cond_pad_goal(goal[sl], C), [u[sl[0], :, sl[2], sl[3]] for u in us], prm, 3, GATE_K)
            regions.append(reg[0])
            n_amb.append(int(cnt[0]))
    dt = torch.bfloat16 if bf else torch.float32
    xd, gd = x.to(DEV, dt), goal.to(DEV, dt)
    w = cond_w(ops, prm, xd)
    out, states, pre = ops.cond_grow(xd, Tn, gd, us.to(DEV), w, 3, keep_history=True)
    gr = ops.cond_grow_backward(states, pre, gd, us.to(DEV), w, cot.to(DEV), Tn, 3)
    # fp32: max-norm, the north_star's bar.  bf16 (storage rounding of T states, bf16 matrix operands, ReLU gates taken from the
    # bf16 recomputation): the trajectory and a few per cent of the gates differ from the fp32 oracle's, so forward within 3e-2
    # max-norm and gradients within 8 % relative L2 (tests/test_gpu_bf16.py bounds the same kernel at 2-4 % against the oracle
    # that shares its rounding points)
    # fp32 gradients, per item: the north_star-level max-norm bound 2e-4.  With ~1e8 hidden pre-activations in this test a few
    # lie within fp32 rounding of zero; such a gate resolves differently under the MFMA's and the CPU convolution's summation
    # orders and moves ONE cell's gradient by ~1e-3 of the maximum (seen: 1e-3 / 3e-3 at single cells, a different item for a
    # different host thread count).  That excuse is PROVEN per item, not assumed: an item that misses 2e-4 must (i) miss it only
    # inside the influence region of gates the ORACLE itself reports within GATE_K of zero (cond_gate_influence: Chebyshev
    # distance t + 1 of the cell), (ii) still meet relative L2 1e-3 with the largest deviation below 1e-2, and (iii) such items
    # are at most 4 of the 32.  The weight gradients (sums over every cell of every item) get the L2 bound only when some item
    # needed the excuse, the max-norm bound otherwise.
    ftol = REL_TOL if not bf else 3e-2
    excused = []
    for b in range(B):
        y0, x0_ = wins[b]
        win = (b, slice(None), slice(y0, y0 + CR), slice(x0_, x0_ + CR))
        assert rel_err(out[win].float(), out_ref[b]) < ftol, b
        if bf:
            assert _rel2(gr["x0"][win], gx_ref[b]) < 8e-2 and _rel2(gr["goal"][win], gg_ref[b]) < 8e-2, b
        else:
            strict = _rel(gr["x0"][win], gx_ref[b]) < 2e-4 and _rel(gr["goal"][win], gg_ref[b]) < 2e-4
            if not strict:
                ok1, out1, in1 = grads_match_outside(gr["x0"][win], gx_ref[b], regions[b])
                ok2, out2, in2 = grads_match_outside(gr["goal"][win], gg_ref[b], regions[b])
                assert n_amb[b] > 0 and ok1 and ok2, (b, n_amb[b], out1, in1, out2, in2)      # (i)
                assert _rel2(gr["x0"][win], gx_ref[b]) < 1e-3 and _rel(gr["x0"][win], gx_ref[b]) < 1e-2, b   # (ii)
                assert _rel2(gr["goal"][win], gg_ref[b]) < 1e-3 and _rel(gr["goal"][win], gg_ref[b]) < 1e-2, b
                excused.append(b)
        dead = torch.ones(S, S, dtype=torch.bool)
        dead[y0:y0 + CR, x0_:x0_ + CR] = False
        assert float(gr["x0"][b][:, dead.to(DEV)].abs().max()) == 0.0 and float(out[b][:, dead.to(DEV)].float().abs().max()) == 0.0
    assert len(excused) <= 4, excused                                                        # (iii)
    names = {"wp": "perception_net.weight", "w1": "update_net.out.0.weight", "b1": "update_net.out.0.bias",
             "w2": "update_net.out.2.weight", "b2": "update_net.out.2.bias", "w3": "update_net.out.4.weight"}
    for k, n in names.items():
        if bf:
            assert _rel2(gr[k].reshape(-1), wsum[n].reshape(-1)) < 8e-2, k
        elif excused:
            assert _rel2(gr[k].reshape(-1), wsum[n].reshape(-1)) < 1e-3 and _rel(gr[k].reshape(-1), wsum[n].reshape(-1)) < 1e-2, k
        else:
            assert _rel(gr[k].reshape(-1), wsum[n].reshape(-1)) < 2e-4, k


def test_cfg3_batch_independence_and_determinism(ops):
    """B=32 C=16 256^2 fully alive: items 5..8 of the 32-item call equal a 4-item call bit for bit (forward and dL/dx0),
    and two identical backward calls agree bitwise (no float atomics)."""
    B, C, S, Tn = 32, 16, 256, 3
    prm = rand_cond_prm(C, seed=4, out_scale=0.5)
    gen = torch.Generator().manual_seed(21)
    x = torch.rand(B, C, S, S, generator=gen).to(DEV)
    goal = (torch.randn(B, 12, S, S, generator=gen) * 0.5).to(DEV)
    us = torch.rand(Tn, B, 1, S, S, generator=gen).to(DEV)
    cot = torch.randn(B, C, S, S, generator=gen).to(DEV)
    w = cond_w(ops, prm, x)
    out, states, pre = ops.cond_grow(x, Tn, goal, us, w, 3, keep_history=True)
    g1 = ops.cond_grow_backward(states, pre, goal, us, w, cot, Tn, 3)
    g2 = ops.cond_grow_backward(states, pre, goal, us, w, cot, Tn, 3)
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k
    sl = slice(5, 9)
    o4, s4, p4 = ops.cond_grow(x[sl].contiguous(), Tn, goal[sl].contiguous(), us[:, sl].contiguous(), w, 3, keep_history=True)
    g4 = ops.cond_grow_backward(s4, p4, goal[sl].contiguous(), us[:, sl].contiguous(), w, cot[sl].contiguous(), Tn, 3)
    assert torch.equal(out[sl], o4) and torch.equal(g1["x0"][sl], g4["x0"]) and torch.equal(g1["goal"][sl], g4["goal"])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_default_model_c20_full_size_properties(ops, dtype):
    """The reference's default channel count (C = 20, nca.py:62-94) at the bench grid, 8 x 20 x 256^2, 3 steps, fp32 and bf16 pool
    (size-independent properties; values are pinned at small sizes by G11 and the oracle tests): the backward is LINEAR in the
    cotangent (every gradient of a*c1 + b*c2 equals a*g(c1) + b*g(c2) to rounding), two identical calls agree bit for bit
    (deterministic slabs, no float atomics), items 2..3 of the batch equal a 2-item call (forward and dL/dx0 / dL/dgoal bit for bit)."""
    B, C, S, Tn = 8, 20, 256, 3
    prm = rand_cond_prm(C, seed=6, out_scale=0.5)
    gen = torch.Generator().manual_seed(22)
    x = torch.rand(B, C, S, S, generator=gen).to(DEV).to(dtype)
    goal = (torch.randn(B, 16, S, S, generator=gen) * 0.5).to(DEV).to(dtype)
    us = torch.rand(Tn, B, 1, S, S, generator=gen).to(DEV)
    c1 = torch.randn(B, C, S, S, generator=gen).to(DEV)
    c2 = torch.randn(B, C, S, S, generator=gen).to(DEV)
    w = cond_w(ops, prm, x)
    out, states, pre = ops.cond_grow(x, Tn, goal, us, w, 3, keep_history=True)
    assert states.dtype == dtype and bool(torch.isfinite(out.float()).all())
    g1 = ops.cond_grow_backward(states, pre, goal, us, w, c1, Tn, 3)
    g1b = ops.cond_grow_backward(states, pre, goal, us, w, c1, Tn, 3)
    g2 = ops.cond_grow_backward(states, pre, goal, us, w, c2, Tn, 3)
    g12 = ops.cond_grow_backward(states, pre, goal, us, w, 0.75 * c1 - 1.5 * c2, Tn, 3)
    tol = 2e-5 if dtype == torch.float32 else 2e-2      # bf16 MFMA rounds the gradient operands of its products: linear up to bf16 rounding
    for k in g1:
        assert torch.equal(g1[k], g1b[k]), k
        lin = 0.75 * g1[k] - 1.5 * g2[k]
        assert _rel(g12[k].reshape(-1), lin.reshape(-1)) < tol, (k, _rel(g12[k].reshape(-1), lin.reshape(-1)))
    sl = slice(2, 4)
    o2, s2, p2 = ops.cond_grow(x[sl].contiguous(), Tn, goal[sl].contiguous(), us[:, sl].contiguous(), w, 3, keep_history=True)
    h2 = ops.cond_grow_backward(s2, p2, goal[sl].contiguous(), us[:, sl].contiguous(), w, c1[sl].contiguous(), Tn, 3)
    assert torch.equal(out[sl], o2) and torch.equal(g1["x0"][sl], h2["x0"]) and torch.equal(g1["goal"][sl], h2["goal"])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_cfg3_full_length_module_training_step(ops, dtype):
    """configs[2] as the trainer issues it: ConditionedNCA.grow(B=32, 96 steps) + a loss + backward + finite gradients for
    every parameter (encoder included), fp32 and bf16 pool.  Values are pinned by the crop test above; this is the
    full-length, full-size call (history ring of 97 slots) going through once."""
    from ncahip.nca import ConditionedNCA
    torch.manual_seed(0)
    m = ConditionedNCA(target_shape=(3, 256, 256), num_hidden_channels=12, living_channel_dim=3).to(DEV)
    m.mask_rng = "philox"
    x = m.generate_seed(32).to(DEV, dtype)
    x[:, :, 96:160, 96:160] = torch.rand(32, 16, 64, 64, device=DEV).to(dtype)
    target = torch.rand(32, 3, 256, 256, device=DEV)
    out = m.grow(x, 96, target)
    assert out.dtype == dtype and out.shape == x.shape
    loss = (out[:, :3].float() - target).square().mean() + out.float().abs().mean() * 0.01
    loss.backward()
    n = 0
    for name, p in m.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()), name
            n += 1
    assert n == 9 and float(m.update_net.out[4].weight.grad.abs().max()) > 0.0


# ---------------------------------------------------------------------------------------------- configs[4]
@pytest.mark.parametrize("pad", ["circular"])
def test_cfg5_dynca_c32_fc256_512_by_periodicity(ops, pad):
    """BASELINE configs[4]'s shape (DyNCA C=32, fc=256, 3 conditioning channels, 2 x 512^2): with circular padding an 8x8
    tiling of a 64^2 problem evolves into the tiling of the 64^2 result, which the oracle computes directly; dL/dx0 tiles
    likewise and the weight gradients are 64 x the tile's (oracle autograd on the tile)."""
    C, fc, cc, S, Tn, R = 32, 256, 3, 64, 3, 8
    prm = rand_dynca_prm(C, fc, cc, seed=7)
    gen = torch.Generator().manual_seed(13)
    xs = torch.rand(2, C, S, S, generator=gen) - 0.5
    cs = torch.rand(2, cc, S, S, generator=gen) * 2 - 1
    us = torch.rand(Tn, 2, 1, S, S, generator=gen)
    cot = torch.randn(2, C, S, S, generator=gen)
    ref, gx_ref, gw_ref = O.dynca_nsteps_loss_grads(xs, cs, list(us), prm, pad, 0.5, cot)
    tile = lambda t: t.repeat(*([1] * (t.dim() - 2)), R, R)
    xd, cd, ud, ctd = tile(xs).to(DEV), tile(cs).to(DEV), tile(us).to(DEV), tile(cot).to(DEV)
    w = ops.DyncaWeights(prm["w1.weight"], prm["w1.bias"], prm["w2.weight"], prm["w2.bias"], xd)
    out, states = ops.dynca_nsteps(xd, Tn, cd, ud, w, pad, 0.5, keep_history=True)
    assert rel_err(out.cpu(), tile(ref)) < REL_TOL
    gr = ops.dynca_nsteps_backward(states, cd, ud, w, ctd, None, Tn, pad, 0.5)
    assert _rel(gr["x0"], tile(gx_ref)) < 2e-4
    for k, n in (("w1", "w1.weight"), ("b1", "w1.bias"), ("w2", "w2.weight"), ("b2", "w2.bias")):
        assert _rel(gr[k].reshape(-1), (R * R) * gw_ref[n].reshape(-1)) < 2e-4, k


def test_cfg5_dynca_module_trains_at_c32(ops):
    """The drop-in DyNCA(c_in=32, fc_dim=256, 'edges') at 1 x 512^2: forward_nsteps + backward on the native kernels
    (no composed / library-GEMM path), gradients finite and non-zero."""
    from ncahip.models.dynca import DyNCA
    torch.manual_seed(0)
    d = DyNCA(32, 3, fc_dim=256, padding_mode="replicate", conditioning="edges", edge_transform="tanh", device=torch.device(DEV))
    d.mask_rng = "philox"
    x = d.seed(1, size=512) + (torch.rand(1, 32, 512, 512, device=DEV) - 0.5)
    cimg = torch.rand(1, 1, 512, 512, device=DEV) * 2 - 1
    assert not d._composed(x)
    xT, rgb = d.forward_nsteps(x, 4, cond_img=cimg)
    (rgb.square().mean() + xT.abs().mean()).backward()
    for n, p in d.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()) and float(p.grad.abs().max()) > 0.0, n


# ---------------------------------------------------------------------------------------------- beyond the oracle: 2048^2
def test_large_plane_2048_cond_tile_vs_generic_and_crop(ops):
    """1 x 16 x 2048^2: (i) tile kernels (32-bit in-plane offsets, XCD-chunked walks) == generic kernels; (ii) forward and
    backward of a live patch in a dead image == the 192^2 crop that contains it (fp32, and bf16 storage bit-exact)."""
    C, S = 16, 2048
    prm = rand_cond_prm(C, seed=0, out_scale=0.5)
    gen = torch.Generator().manual_seed(0)
    x = torch.rand(1, C, S, S, generator=gen).to(DEV)
    goal = (torch.randn(1, 12, S, S, generator=gen) * 0.5).to(DEV)
    w = cond_w(ops, prm, x)
    outs = []
    for force in (0, 1):
        ops.force_generic(force)
        o, _, _ = ops.cond_grow(x, 3, goal, None, w, 3, seed=7)
        outs.append(o)
    ops.force_generic(False)
    assert float((outs[0] - outs[1]).abs().max()) < 1e-5
    del outs, x, goal
    Tn, c0 = 2, S - 192
    g2 = torch.Generator().manual_seed(5)
    xc = torch.zeros(1, C, 192, 192)
    xc[:, :, 60:150, 70:160] = torch.rand(1, C, 90, 90, generator=g2)
    gc = torch.randn(1, 12, 192, 192, generator=g2) * 0.5
    uc = torch.rand(Tn, 1, 1, 192, 192, generator=g2)
    ct = torch.randn(1, C, 192, 192, generator=g2)

    def embed(t, fill=0.0):
        full = torch.full(t.shape[:-2] + (S, S), fill)
        full[..., c0:, c0:] = t
        return full
    res = []
    for xx, gg, uu, cc in ((xc, gc, uc, ct), (embed(xc), embed(gc), embed(uc, 0.5), embed(ct))):
        xx, gg, uu, cc = xx.to(DEV), gg.to(DEV), uu.to(DEV), cc.to(DEV)
        out, states, pre = ops.cond_grow(xx, Tn, gg, uu, w, 3, keep_history=True)
        res.append((out, ops.cond_grow_backward(states, pre, gg, uu, w, cc, Tn, 3)))
        ob, _, _ = ops.cond_grow(xx.bfloat16(), Tn, gg.bfloat16(), uu, w, 3)
        res[-1] += (ob.float(),)
    (oc, gcr, obc), (of, gfr, obf) = res
    assert _rel(of[..., c0:, c0:], oc) < 1e-5 and _rel(gfr["x0"][..., c0:, c0:], gcr["x0"]) < 2e-4
    assert float(gfr["x0"][..., :c0, :].abs().max()) == 0.0
    for k in ("w1", "w2", "w3", "b1", "b2", "wp", "goal"):
        a = gfr[k][..., c0:, c0:] if k == "goal" else gfr[k]
        assert _rel(a, gcr[k]) < 2e-4, k
    assert torch.equal(obf[..., c0:, c0:], obc)


def test_large_plane_2048_dynca_by_periodicity(ops):
    """1 x 16 x 2048^2 DyNCA: an 8x8 circular tiling of a 256^2 problem, forward + backward (weight gradients = 64 x)."""
    C = 16
    prm = rand_dynca_prm(C, 128, 3, seed=1)
    g3 = torch.Generator().manual_seed(11)
    xs = torch.rand(1, C, 256, 256, generator=g3) - 0.5
    cs = torch.rand(1, 3, 256, 256, generator=g3) * 2 - 1
    us = torch.rand(2, 1, 1, 256, 256, generator=g3)
    cts = torch.randn(1, C, 256, 256, generator=g3)
    tile = lambda t: t.repeat(*([1] * (t.dim() - 2)), 8, 8)
    rs = []
    for f in (lambda t: t, tile):
        xx, cn, uu, ct = f(xs).to(DEV), f(cs).to(DEV), f(us).to(DEV), f(cts).to(DEV)
        w = ops.DyncaWeights(prm["w1.weight"], prm["w1.bias"], prm["w2.weight"], prm["w2.bias"], xx)
        out, states = ops.dynca_nsteps(xx, 2, cn, uu, w, "circular", 0.5, keep_history=True)
        gr = ops.dynca_nsteps_backward(states, cn, uu, w, ct, None, 2, "circular", 0.5)
        rs.append((out.cpu(), {k: v.cpu() for k, v in gr.items()}))
    (os_, gs), (ot, gt) = rs
    assert _rel(ot, tile(os_)) < 1e-5 and _rel(gt["x0"], tile(gs["x0"])) < 2e-4
    for k in ("w1", "b1", "w2", "b2"):
        assert _rel(gt[k], 64.0 * gs[k]) < 2e-4, k
