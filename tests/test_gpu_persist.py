"""GPU tests of the one-launch persistent DyNCA kernel (ncahip_dynca_nsteps_fwd_persist_f32, csrc/nca_dynca_persist.hip):
B = 1 video inference (ConditioneDyNCA/utils/misc/video_utils.py:50-82).  Its contract is "the same numbers as the per-step
kernels, bit for bit", so every case is checked with torch.equal against the per-step path (which the golden fixtures G3 / G5 /
G6 and the oracle pin), plus the oracle directly at 1e-4."""
import numpy as np
import pytest
import torch

from oracle import nca_oracle as O
from util import REL_TOL, T, load, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from ncahip import ops as _ops
    _ops.selftest()
    _ops.force_generic(0)
    yield _ops
    _ops.persistent_steps = True


def _prm(C, fc, cc, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    k1 = 4 * C + cc
    return {"w1.weight": torch.randn(fc, k1, 1, 1, generator=g) * (0.5 / k1 ** 0.5), "w1.bias": torch.randn(fc, generator=g) * 0.1,
            "w2.weight": torch.randn(C, fc, 1, 1, generator=g) * (scale * 0.3 / fc ** 0.5), "w2.bias": torch.randn(C, generator=g) * 0.02}


def _both(ops, x, Tn, cond, us, w, pad, rate, **kw):
    ops.persistent_steps = False
    ref, _ = ops.dynca_nsteps(x, Tn, cond, us, w, pad, rate, **kw)
    ref = ref.clone()
    ops.persistent_steps = True
    got, _ = ops.dynca_nsteps(x, Tn, cond, us, w, pad, rate, **kw)
    got = got.clone()
    ops.check_errors()
    return ref, got


@pytest.mark.parametrize("pad", ["replicate", "circular", "reflect", "constant"])
@pytest.mark.parametrize("C,fc,cc,shape", [(12, 96, 3, (1, 64, 64)), (16, 128, 2, (2, 32, 96)), (8, 64, 0, (1, 16, 16)), (12, 96, 3, (1, 256, 256))])
def test_persistent_equals_per_step_bit_for_bit(ops, pad, C, fc, cc, shape):
    B, H, W = shape
    gen = torch.Generator().manual_seed(C + H + len(pad))
    prm = _prm(C, fc, cc, seed=C + cc, scale=2.0)
    x = (torch.rand(B, C, H, W, generator=gen) - 0.5).to(DEV)
    cond = (torch.rand(B, cc, H, W, generator=gen) * 2 - 1).to(DEV) if cc else None
    w = ops.DyncaWeights(prm["w1.weight"], prm["w1.bias"], prm["w2.weight"], prm["w2.bias"], x)
    assert ops.lib().ncahip_dynca_nsteps_persist_workspace(B, C, H, W, fc, cc) > 0
    for Tn, mode in ((7, "u"), (12, "philox"), (2, "bits"), (33, "u")):
        us = None
        if mode != "philox":
            us = torch.rand(Tn, B, 1, H, W, generator=gen).to(DEV)
            if mode == "bits":
                us = ops.pack_fire_mask(us, 0.5, "dynca")
        ref, got = _both(ops, x, Tn, cond, us, w, pad, 0.5, seed=77, step0=5)
        assert torch.equal(ref, got), (pad, C, fc, shape, Tn, mode, float((ref - got).abs().max()))


@pytest.mark.parametrize("pad", ["replicate", "circular", "reflect", "constant"])
@pytest.mark.parametrize("C,fc,cc,shape", [(12, 96, 2, (1, 64, 64)), (16, 128, 3, (2, 32, 48)), (8, 64, 0, (1, 16, 16)), (12, 96, 2, (1, 256, 256))])
def test_persistent_two_scale_equals_per_step_bit_for_bit(ops, pad, C, fc, cc, shape):
    """perception_scales = [0, 1] (every shipped video model): the one-launch kernel against coarse-perceive + fused step launches."""
    B, H, W = shape
    gen = torch.Generator().manual_seed(C + H + len(pad) + 1)
    prm = _prm(C, fc, cc, seed=C + cc + 1, scale=2.0)
    x = (torch.rand(B, C, H, W, generator=gen) - 0.5).to(DEV)
    cond = (torch.rand(B, cc, H, W, generator=gen) * 2 - 1).to(DEV) if cc else None
    w = ops.DyncaWeights(prm["w1.weight"], prm["w1.bias"], prm["w2.weight"], prm["w2.bias"], x)
    for Tn, mode in ((1, "u"), (6, "philox"), (2, "bits"), (19, "u")):
        us = None
        if mode != "philox":
            us = torch.rand(Tn, B, 1, H, W, generator=gen).to(DEV)
            if mode == "bits":
                us = ops.pack_fire_mask(us, 0.5, "dynca")
        ref, got = _both(ops, x, Tn, cond, us, w, pad, 0.5, seed=78, step0=3, two_scale=True)
        assert torch.equal(ref, got), (pad, C, fc, shape, Tn, mode, float((ref - got).abs().max()))


def test_persistent_two_scale_vs_oracle(ops):
    gen = torch.Generator().manual_seed(13)
    prm = _prm(12, 96, 2, seed=4, scale=2.0)
    x = torch.rand(1, 12, 32, 48, generator=gen) - 0.5
    cond = O.cpe2d(1, 32, 48)
    us = torch.rand(7, 1, 1, 32, 48, generator=gen)
    for pad in ("circular", "replicate"):
        ref = O.dynca_nsteps(x, cond, list(us), prm, pad, 0.5, scales=(0, 1))
        w = ops.DyncaWeights(prm["w1.weight"], prm["w1.bias"], prm["w2.weight"], prm["w2.bias"], x.to(DEV))
        ops.persistent_steps = True
        got, _ = ops.dynca_nsteps(x.to(DEV), 7, cond.to(DEV), us.to(DEV), w, pad, 0.5, two_scale=True)
        ops.check_errors()
        assert rel_err(got, ref) < REL_TOL, pad


def test_persistent_vs_oracle_and_rates(ops):
    gen = torch.Generator().manual_seed(3)
    prm = _prm(12, 96, 3, seed=1, scale=2.0)
    x = torch.rand(1, 12, 16, 64, generator=gen) - 0.5
    cond = O.edge_extractor(torch.rand(1, 1, 16, 64, generator=gen) * 2 - 1, "tanh")
    for rate in (0.5, 0.25, 1.0, 0.0):
        us = torch.rand(9, 1, 1, 16, 64, generator=gen)
        ref = O.dynca_nsteps(x, cond, list(us), prm, "replicate", rate)
        w = ops.DyncaWeights(prm["w1.weight"], prm["w1.bias"], prm["w2.weight"], prm["w2.bias"], x.to(DEV))
        ops.persistent_steps = True
        got, _ = ops.dynca_nsteps(x.to(DEV), 9, cond.to(DEV), us.to(DEV), w, "replicate", rate)
        ops.check_errors()
        assert rel_err(got, ref) < REL_TOL, rate


def test_persistent_100_steps_with_trained_weights(ops):
    """The shipped vector-field model (G5's weights, decoded from docs/data/vec_field_models/large/starry-night.json) at B = 1,
    64 x 64, 100 steps: the persistent launch equals the per-step kernels bit for bit (and those reproduce the reference's own
    100-step run of this model: test_gpu_parity.py G5)."""
    g = load("g5_real_weights")
    prm = {"w1.weight": T(g["w1"]), "w1.bias": T(g["b1"]), "w2.weight": T(g["w2"]), "w2.bias": T(g["b2"])}
    gen = torch.Generator().manual_seed(8)
    yy, xx = torch.meshgrid(torch.arange(64.0), torch.arange(64.0), indexing="ij")
    disc = (((yy - 32) ** 2 + (xx - 32) ** 2) < 400).float()[None, None] * 2 - 1
    cond = O.edge_extractor(disc, "tanh").to(DEV)
    us = torch.rand(100, 1, 1, 64, 64, generator=gen).to(DEV)
    x = torch.zeros(1, 12, 64, 64, device=DEV)
    w = ops.DyncaWeights(prm["w1.weight"], prm["w1.bias"], prm["w2.weight"], prm["w2.bias"], x)
    ref, got = _both(ops, x, 100, cond, us, w, "circular", 0.5)
    assert torch.equal(ref, got) and float(got.abs().max()) > 0.1
    ref_o = O.dynca_nsteps(x.cpu(), cond.cpu(), list(us.cpu()), prm, "circular", 0.5)
    assert rel_err(got, ref_o) < REL_TOL


def test_uncovered_shapes_fall_back(ops):
    """H % 16 / W % 16 / C > 16 / keep_history: the per-step kernels run (same call, same results as before)."""
    assert ops.lib().ncahip_dynca_nsteps_persist_workspace(1, 12, 40, 48, 96, 3) == 0
    assert ops.lib().ncahip_dynca_nsteps_persist_workspace(1, 32, 64, 64, 256, 3) == 0
    gen = torch.Generator().manual_seed(4)
    prm = _prm(12, 96, 0, seed=2)
    x = (torch.rand(1, 12, 20, 40, generator=gen) - 0.5)
    us = torch.rand(4, 1, 1, 20, 40, generator=gen)
    w = ops.DyncaWeights(prm["w1.weight"], prm["w1.bias"], prm["w2.weight"], prm["w2.bias"], x.to(DEV))
    ops.persistent_steps = True
    got, _ = ops.dynca_nsteps(x.to(DEV), 4, None, us.to(DEV), w, "circular", 0.5)
    assert rel_err(got, O.dynca_nsteps(x, None, list(us), prm, "circular", 0.5)) < REL_TOL
    # too many tiles for one wave of workgroups: NCAHIP_ERANGE from the driver, transparent fallback
    xb = (torch.rand(4, 12, 256, 256, generator=gen) - 0.5).to(DEV)
    wb = ops.DyncaWeights(prm["w1.weight"], prm["w1.bias"], prm["w2.weight"], prm["w2.bias"], xb)
    ref, got = _both(ops, xb, 3, None, None, wb, "circular", 0.5, seed=3)
    assert torch.equal(ref, got)


def test_video_module_runs_the_persistent_kernel(ops):
    """DyNCA.forward_nsteps under no_grad at B = 1, 256^2 (the video loop's call) goes through the persistent launch."""
    from ncahip.models.dynca import DyNCA
    torch.manual_seed(5)
    m = DyNCA(12, 3, fc_dim=96, padding_mode="circular", conditioning="pos_emb", device=torch.device(DEV))
    x = m.seed(1, size=256)
    x = x + 0.1 * torch.randn_like(x)
    seen = []
    L = ops.lib()
    orig = L.ncahip_dynca_nsteps_fwd_persist_f32

    class Spy:
        def __call__(self, *a):
            rc = orig(*a)
            seen.append(rc)
            return rc
    L.ncahip_dynca_nsteps_fwd_persist_f32 = Spy()
    try:
        ops.persistent_steps = True
        with torch.no_grad():
            torch.manual_seed(9)
            a, rgb = m.forward_nsteps(x, 16)
            ops.persistent_steps = False
            torch.manual_seed(9)
            b, _ = m.forward_nsteps(x, 16)
    finally:
        L.ncahip_dynca_nsteps_fwd_persist_f32 = orig
        ops.persistent_steps = True
    assert seen == [0] and torch.equal(a, b) and rgb.shape == (1, 3, 256, 256)
    ops.check_errors()


def test_missing_neighbour_drains_and_reports(ops):
    """A workgroup that never runs (test hook: the launch leaves out its last tiles) must not hang the others: their bounded polls
    expire, every workgroup leaves its step loop, the sticky error word carries bit 1, ncahip_check_errors raises and clears it,
    and the next launch is healthy again."""
    from ncahip._capi import NcaHipError
    gen = torch.Generator().manual_seed(11)
    prm = _prm(12, 96, 0, seed=3)
    x = (torch.rand(1, 12, 64, 64, generator=gen) - 0.5).to(DEV)
    w = ops.DyncaWeights(prm["w1.weight"], prm["w1.bias"], prm["w2.weight"], prm["w2.bias"], x)
    ops.check_errors()
    ops.persistent_steps = True
    L = ops.lib()
    assert L.ncahip_debug_persist_drop_tiles(3) == 0
    try:
        ops.dynca_nsteps(x, 6, None, None, w, "circular", 0.5, seed=1)
        torch.cuda.synchronize()                       # returns: the launch drained
        with pytest.raises(NcaHipError, match="neighbour poll"):
            ops.check_errors()
    finally:
        L.ncahip_debug_persist_drop_tiles(0)
    ops.check_errors()                                  # cleared
    ref, got = _both(ops, x, 6, None, None, w, "circular", 0.5, seed=1)
    assert torch.equal(ref, got)
