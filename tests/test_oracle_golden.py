"""The oracle (oracle/nca_oracle.py) replayed against golden vectors captured from the
reference's own PyTorch modules (tests/golden/gen_golden.py).  CPU only, bit-exact
where the restatement uses the same aten ops (everything except the float64 numpy twin)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import nca_oracle as O

torch.set_num_threads(1)   # NOTE: process-wide (the GPU suite imports this module at collection): oracle summation order


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    return {k: z[k] for k in z.files}


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def sd(g, prefix="sd."):
    return {k[len(prefix):]: T(v) for k, v in g.items() if k.startswith(prefix)}


def same(a, b):
    a = a.numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return np.array_equal(a, np.asarray(b))


# ------------------------------------------------------------------ ConditionedNCA
def test_g1_cond_step_bit_exact(golden_dir):
    g = load(golden_dir, "g1_cond_step")
    prm = sd(g)
    r = O.cond_step(T(g["x"]), T(g["genc"]), T(g["u"]), prm, int(g["alive_ch"]), float(g["thr"]),
                    float(g["fire_rate"]), return_all=True)
    for k in ("pre", "rmask", "p", "out", "x1", "post", "x2"):
        assert same(r[k], g[k]), k
    # the fixture really exercises every branch
    assert 0 < g["pre"].mean() < 1 and 0 < g["rmask"].mean() < 1
    assert (np.abs(g["x2"]) == 10.0).any()
    assert (g["pre"] & ~g["post"]).any() or (g["post"] & ~g["pre"]).any()


def test_g1_cond_step_numpy_twin(golden_dir):
    g = load(golden_dir, "g1_cond_step")
    prm = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    x2 = O.cond_step_np(g["x"], g["genc"], g["u"], prm, int(g["alive_ch"]), float(g["thr"]), float(g["fire_rate"]))
    # float64 loops vs fp32 aten: cells whose alpha sits within 1e-6 of the threshold may flip
    bad = np.abs(x2 - g["x2"]) > 1e-4
    assert bad.mean() < 1e-3


@pytest.mark.parametrize("tag", ["seed", "rand"])
def test_g2_cond_grow(golden_dir, tag):
    g = load(golden_dir, "g2_cond_grow")
    prm = sd(g)
    C = g[f"{tag}_x0"].shape[1]
    genc = O.image_encoder(T(g["goal"]), prm)
    assert same(genc, g["genc"])
    gpad = O.cond_pad_goal(genc, C)
    us = [T(u) for u in g[f"{tag}_us"]]
    xT, states = O.cond_grow(T(g[f"{tag}_x0"]), gpad, us, prm, int(g["alive_ch"]), collect=True)
    assert same(torch.stack(states), g[f"{tag}_states"])
    # teacher-forced: each single step from the reference's own state
    ref_states = [T(g[f"{tag}_x0"])] + [T(s) for s in g[f"{tag}_states"]]
    for t, u in enumerate(us):
        assert same(O.cond_step(ref_states[t], gpad, u, prm, int(g["alive_ch"])), ref_states[t + 1])
    # global-RNG stream contract (one rand_like per step, nothing else)
    torch.manual_seed(55)
    assert same(O.cond_grow_rng(T(g[f"{tag}_x0"]), gpad, int(g["T"]), prm, int(g["alive_ch"])), g[f"{tag}_grow55"])
    if tag == "seed":  # the seed really grew
        assert (g["seed_states"][-1] != 0).sum() > (g["seed_x0"] != 0).sum()


def test_g2l_cfg1_exact_config(golden_dir):
    """BASELINE configs[0]: B=4 C=12 128x128 T=32, inputs regenerated from seeds."""
    g = load(golden_dir, "g2l_cfg1")
    prm = sd(g)
    torch.manual_seed(int(g["data_seed"]))
    x = torch.rand(4, 12, 128, 128)
    goal = torch.rand(4, 3, 128, 128)
    gpad = O.cond_pad_goal(O.image_encoder(goal, prm), 12)
    torch.manual_seed(int(g["rng_seed"]))
    sums, nal = [], []
    for t in range(int(g["T"])):
        x = O.cond_step(x, gpad, torch.rand_like(x[:, 0:1]), prm, 3)
        sums.append(float(x.double().sum())); nal.append(int(O.cond_alive(x, 3).sum()))
    assert same(x[:, :, 56:72, 56:72], g["crop"])
    assert sums == list(g["sums"]) and nal == list(g["nalive"])


def test_g9_seeds(golden_dir):
    g = load(golden_dir, "g9_seeds")
    assert same(O.cond_generate_seed(2, 12, 3, 16), g["cond_seed"])
    assert same(O.cond_generate_seed(1, 12, 3, 10), g["cond_seed_dev"])
    for mode in ("zeros", "center_on", "random"):
        assert same(O.dynca_seed(2, 6, (10, 6), mode), g[f"dynca_seed.{mode}"])


# ------------------------------------------------------------------ DyNCA
def _g3_cases(golden_dir):
    g = load(golden_dir, "g3_dynca")
    return g, json.loads(str(g["cases"]))


def test_g3_dynca_all_cases(golden_dir):
    g, cases = _g3_cases(golden_dir)
    assert len(cases) >= 18
    for c in cases:
        t = c["tag"]
        prm = {k: T(g[f"{t}.{k}"]) for k in ("w1.weight", "w1.bias", "w2.weight", "w2.bias")}
        x0 = T(g[f"{t}.x0"])
        if c["cond"] == "edges":
            cond = O.edge_extractor(T(g[f"{t}.cond_img"]), c["transform"])
            assert same(cond, g[f"{t}.cond"]), c
        elif c["cond"] == "pos_emb":
            cond = O.cpe2d(*[x0.shape[i] for i in (0, 2, 3)])
            assert same(cond, g[f"{t}.cond"]), c
        else:
            cond = None
        assert same(O.dynca_perceive(x0, c["pad"]), g[f"{t}.perc0"]), c
        us = [T(u) for u in g[f"{t}.us"]]
        xT, states = O.dynca_nsteps(x0, cond, us, prm, c["pad"], 0.5, collect=True)
        assert same(states[0], g[f"{t}.state_first"]), c
        assert same(states[-1], g[f"{t}.state_last"]), c
        torch.manual_seed(77)
        xn = O.dynca_nsteps_rng(x0, cond, c["T"], prm, c["pad"], 0.7)
        assert same(xn, g[f"{t}.nsteps77_rate07"]), c
        assert same(O.dynca_to_rgb(xn, 3), g[f"{t}.rgb77"]), c


def test_g4_perception_known_answers(golden_dir):
    g = load(golden_dir, "g4_perception")
    for pad in O.PAD_MODES:
        for name in ("ramp", "imp"):
            y = O.dynca_perceive(T(g[name]), pad)
            assert same(y, g[f"{name}.{pad}"]), (name, pad)
            y64 = O.dynca_perceive_np(g[name], pad)           # independent loops
            assert np.array_equal(y64.astype(np.float32), g[f"{name}.{pad}"]), (name, pad)
    # SURVEY 8c known answers on arange(25): sobel_x(centre)=8, sobel_y(centre)=40, lap(centre)=0, lap corner (replicate)=24
    r = g["ramp.replicate"][0]
    assert r[2, 2, 2] == 8 and r[4, 2, 2] == 40 and r[6, 2, 2] == 0 and r[6, 0, 0] == 24
    # ConditionedNCA perception channel map: input channel 2 -> output channels {6,7,8}
    y = O.cond_perceive(torch.zeros(1, 8, 8, 8).index_put_((torch.tensor(0), torch.tensor(2), torch.tensor(4), torch.tensor(4)),
                                                           torch.tensor(1.0)), T(g["cond_wp"]))
    assert same(y, g["cond_imp_ch2"])
    nz = np.nonzero(np.abs(g["cond_imp_ch2"][0]).sum(axis=(1, 2)))[0]
    assert list(nz) == [6, 7, 8]
    # multi-scale perception
    assert same(O.dynca_perceive_multiscale(T(g["ms_x"]), "replicate", (0, 1)), g["ms_y"])


def test_g5_trained_weights_trajectory(golden_dir):
    """Trained web-demo weights (docs/data/vec_field_models/large/starry-night.json), 100 steps."""
    g = load(golden_dir, "g5_real_weights")
    prm = {"w1.weight": T(g["w1"]), "w1.bias": T(g["b1"]), "w2.weight": T(g["w2"]), "w2.bias": T(g["b2"])}
    assert prm["w1.weight"].shape == (96, 51, 1, 1) and prm["w2.weight"].shape == (12, 96, 1, 1)
    cond = O.edge_extractor(T(g["cond_img"]), "tanh")
    x = O.dynca_seed(1, 12, (48, 48), "zeros")
    torch.manual_seed(int(g["rng_seed"]))
    amax = []
    for t in range(1, 101):
        x = O.dynca_step(x, cond, torch.rand(1, 1, 48, 48), prm, "circular", 0.5)
        amax.append(float(x.abs().max()))
        if t in (25, 50, 100):
            assert same(x, g[f"x_t{t}"]), t
    assert amax == list(g["absmax"]) and 0.1 < amax[-1] < 10.0  # bounded, non-trivial


def test_g6_extra_channels_variant(golden_dir):
    g = load(golden_dir, "g6_extra_channels")
    prm = {k: T(g[k]) for k in ("w1.weight", "w1.bias", "w2.weight", "w2.bias")}
    x0 = T(g["x0"])
    assert list(g["seed_shape"]) == [2, 12, 20, 28]        # seed() emits c_in-1 channels
    pe = O.cpe2d(2, 20, 28)
    assert same(pe, g["pos_emb"])
    xT, states = O.dynca_nsteps(x0, pe, [T(u) for u in g["us"]], prm, "replicate", 0.5, collect=True)
    assert same(torch.stack(states), g["states"])


def test_g7_encoders(golden_dir):
    g = load(golden_dir, "g7_encoders")
    prm = sd(g)
    assert same(O.gaussian_kernel_5x5(), g["sd.encoder.gaussian_blur.weight"])
    assert same(O.image_encoder(T(g["img"]), prm), g["enc_out"])
    assert same(O.edge_extractor(T(g["gray"]), "tanh"), g["edges_tanh"])
    assert same(O.edge_extractor(T(g["gray"]), None), g["edges_none"])
    assert same(O.cpe2d(2, 20, 24), g["cpe"])


# ------------------------------------------------------------------ gradients
def test_g8_cond_grads(golden_dir):
    g = load(golden_dir, "g8_cond_grads")
    prm = sd(g)
    xT, dx0, dg, grads = O.cond_grow_loss_grads(T(g["x0"]), T(g["gpad"]), [T(u) for u in g["us"]], prm,
                                                int(g["alive_ch"]), float(g["thr"]), float(g["fire_rate"]), T(g["cot"]))
    assert same(xT, g["xT"]) and same(dx0, g["d_x0"]) and same(dg, g["d_gpad"])
    for k, v in grads.items():
        assert same(v, g["grad." + k]), k
    assert set(grads) == {"perception_net.weight", "update_net.out.0.weight", "update_net.out.0.bias",
                          "update_net.out.2.weight", "update_net.out.2.bias", "update_net.out.4.weight"}


def test_g11_cond_grads_wide(golden_dir):
    """The reference's DEFAULT model (C = 20) and a C = 32 one: gradients of <cot, grow> through the reference's own autograd,
    encoder included (generated by importing the reference: tests/golden/gen_golden.py g11_cond_grads_wide)."""
    g = load(golden_dir, "g11_cond_grads_wide")
    for tag in ("c20", "c32"):
        prm = {k[len(tag) + 4:]: T(v) for k, v in g.items() if k.startswith(tag + ".sd.")}
        C = g[f"{tag}.x0"].shape[1]
        p = {k: v.clone().requires_grad_(k.startswith(("perception", "update")) or "embed" in k) for k, v in prm.items()}
        x0 = T(g[f"{tag}.x0"]).requires_grad_(True)
        genc = O.image_encoder(T(g[f"{tag}.goal_img"]), p)
        genc.retain_grad()
        assert same(genc.detach(), g[f"{tag}.goal_enc"]), tag
        x = O.cond_grow(x0, O.cond_pad_goal(genc, C), [T(u) for u in g[f"{tag}.us"]], p, int(g[f"{tag}.alive_ch"]),
                        float(g["thr"]), float(g["fire_rate"]))
        (x * T(g[f"{tag}.cot"])).sum().backward()
        assert same(x.detach(), g[f"{tag}.xT"]) and same(x0.grad, g[f"{tag}.d_x0"]) and same(genc.grad, g[f"{tag}.d_goal_enc"]), tag
        n = 0
        for k, v in p.items():
            if v.grad is not None:
                assert same(v.grad, g[f"{tag}.grad.{k}"]), (tag, k)
                n += 1
        assert n == 9, (tag, n)


def test_g8_dynca_grads(golden_dir):
    g = load(golden_dir, "g8_dynca_grads")
    for pad in O.PAD_MODES:
        prm = {k: T(g[f"{pad}.{k}"]) for k in ("w1.weight", "w1.bias", "w2.weight", "w2.bias")}
        cond = O.edge_extractor(T(g[f"{pad}.cond_img"]), "tanh")
        x0 = T(g[f"{pad}.x0"]).requires_grad_(True)
        p = {k: v.clone().requires_grad_(True) for k, v in prm.items()}
        x = O.dynca_nsteps(x0, cond, [T(u) for u in g[f"{pad}.us"]], p, pad, 0.5)
        ((x * T(g[f"{pad}.cot"])).sum() + (O.dynca_to_rgb(x, 3) * T(g[f"{pad}.cot_rgb"])).sum()).backward()
        assert same(x.detach(), g[f"{pad}.xT"]) and same(x0.grad, g[f"{pad}.d_x0"]), pad
        for k in p:
            assert same(p[k].grad, g[f"{pad}.g.{k}"]), (pad, k)


# ------------------------------------------------------------------ Philox known answers
def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    r = O.philox4x32_10(0, 0, 0, 0, 0, 0)
    assert [int(v) for v in r] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    r = O.philox4x32_10(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff)
    assert [int(v) for v in r] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    r = O.philox4x32_10(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0)
    assert [int(v) for v in r] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    u = O.philox_uniform(42, 3, 2, 8, 8)
    assert u.shape == (2, 1, 8, 8) and u.dtype == np.float32 and 0 <= u.min() and u.max() < 1
    assert not np.array_equal(u, O.philox_uniform(42, 4, 2, 8, 8))


# ------------------------------------------------------------------ two-scale perception (shipped video models)
def test_g10_two_scale_video_model_and_webgl_tables(golden_dir):
    """perception_scales = [0, 1]: the oracle against the reference's own forward with the trained weights of a shipped video
    model (docs/data/video_models/small/fountain_1.json) and four random-weight cases; and the WebGL importer against the raw
    layer tables of that JSON (data kept in the fixture)."""
    import sys
    from ncahip import webgl
    g = load(golden_dir, "g10_two_scale")
    assert int(g["n_perception_scales"]) == 2
    layers = []
    for i in range(2):
        meta = json.loads(str(g[f"json.l{i}.meta"]))
        meta["data_flatten"] = g[f"json.l{i}.data"].tolist()
        layers.append(meta)
    w = webgl.load_dynca_weights({"layers": layers, "n_perception_scales": 2})
    for k, n in (("w1.weight", "w1"), ("w1.bias", "b1"), ("w2.weight", "w2"), ("w2.bias", "b2")):
        assert same(w[k], g[n]), k
    assert w["pos_emb"] and not w["edge_conditioning"] and w["n_perception_scales"] == 2
    prm = {"w1.weight": T(g["w1"]), "w1.bias": T(g["b1"]), "w2.weight": T(g["w2"]), "w2.bias": T(g["b2"])}
    x, us = T(g["vid.x0"]), T(g["vid.us"])
    cond = O.cpe2d(1, x.shape[2], x.shape[3])
    assert same(O.dynca_perceive_multiscale(x, "circular", (0, 1), cond)[:, :, :12, -12:], g["vid.perc0_crop"])
    for t in range(us.shape[0]):
        x = O.dynca_step(x, cond, us[t], prm, "circular", 0.5, scales=(0, 1))
        if f"vid.x_t{t + 1}" in g:
            assert same(x, g[f"vid.x_t{t + 1}"]), t
    for c in json.loads(str(g["cases"])):
        t_ = c["tag"]
        p = {k: T(g[f"{t_}.{k}"]) for k in ("w1.weight", "w1.bias", "w2.weight", "w2.bias")}
        x0 = T(g[f"{t_}.x0"])
        cnd = O.edge_extractor(T(g[f"{t_}.cond_img"]), "tanh") if c["cond"] == "edges" else O.cpe2d(*[x0.shape[i] for i in (0, 2, 3)])
        us_ = T(g[f"{t_}.us"])
        assert same(O.dynca_step(x0, cnd, us_[0], p, c["pad"], 0.5, scales=(0, 1)), g[f"{t_}.first"]), c
        assert same(O.dynca_nsteps(x0, cnd, list(us_), p, c["pad"], 0.5, scales=(0, 1)), g[f"{t_}.last"]), c
