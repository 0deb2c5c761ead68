#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own PyTorch modules.

Runs only in the build container (needs /root/reference).  The .npz files hold
data only -- seeded inputs, weights drawn by the reference's own initialisers
and the outputs the reference produced for them -- never reference source.
The GPU box and CI replay them through tests/test_oracle_golden.py.

    python tests/golden/gen_golden.py            # rewrites every fixture

Reference entry points exercised (file:line in the reference checkout):
  EncoderConditioning/nca.py:61-215      ConditionedNCA.forward / grow / alive / generate_seed (default arguments: C = 20, G11)
  EncoderConditioning/encoder.py:5-64    ImageEncoder
  ConditioneDyNCA/models/dynca.py:7-253  DyNCA.forward / forward_nsteps / perceive_torch / seed, EdgeExtractor, CPE2D
  ExtraChannels/models/dynca.py:7-167    DyNCA (state-concat conditioning variant)
  docs/data/vec_field_models/large/starry-night.json   trained weights (data file)
  docs/data/video_models/small/fountain_1.json         trained two-scale video model (data file; n_perception_scales = 2)
"""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = os.environ.get("NCA_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(1)  # bit-stable across machines for the recorded outputs


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


sys.path.insert(0, os.path.join(REF, "EncoderConditioning"))
import nca as ref_nca  # noqa: E402  (reference module)

ref_dynca = _load(os.path.join(REF, "ConditioneDyNCA/models/dynca.py"), "ref_dynca_cond")
ref_dynca_x = _load(os.path.join(REF, "ExtraChannels/models/dynca.py"), "ref_dynca_extra")
CPU = torch.device("cpu")


def sd_np(module):
    return {"sd." + k: v.detach().cpu().numpy() for k, v in module.state_dict().items()}


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def rand_biases_(model):
    """The reference zero-inits biases; give them values so a dropped bias shows up."""
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("bias"):
                p.uniform_(-0.1, 0.1)


# ---------------------------------------------------------------- G1: one ConditionedNCA step
def g1_cond_step():
    torch.manual_seed(0)
    m = ref_nca.ConditionedNCA(target_shape=(3, 32, 32), num_hidden_channels=8, living_channel_dim=3)
    rand_biases_(m.update_net)
    C, a = m.num_channels, m.living_channel_dim
    torch.manual_seed(1)
    x = torch.randn(2, C, 32, 32) * 0.5
    x[:, a] = torch.rand(2, 32, 32) * 0.25  # alpha straddles the 0.1 threshold
    x[0, a, :8, :] = 0.0                     # a dead band (pre mask false, goal gated off)
    x[1, :, 20:24, 20:24] *= 30.0            # exercise the +-10 clamp
    genc = torch.randn(2, 8, 32, 32)
    genc = torch.nn.functional.pad(genc, (0, 0, 0, 0, C - 8, 0))
    with torch.no_grad():
        torch.manual_seed(7); u = torch.rand_like(x[:, 0:1])
        pre = m.alive(x)
        torch.manual_seed(7); rmask = m.get_stochastic_update_mask(x)
        p = m.perception_net(x + genc * pre)
        out = m.update(x, genc, pre)
        torch.manual_seed(7); x2, _ = m.forward((x, genc))
        x1 = x + rmask * out
        post = m.alive(x1)
    save("g1_cond_step", x=x, genc=genc, u=u, pre=pre, rmask=rmask, p=p, out=out, x1=x1, post=post, x2=x2,
         alive_ch=a, thr=m.alpha_living_threshold, fire_rate=m.cell_fire_rate, **sd_np(m))


# ---------------------------------------------------------------- G2: grow, per-step states
def g2_cond_grow():
    torch.manual_seed(0)
    m = ref_nca.ConditionedNCA(target_shape=(3, 32, 32), num_hidden_channels=8, living_channel_dim=3)
    rand_biases_(m.update_net)
    # the reference's default init barely moves a seed in 8 steps; scale the last layer so it grows
    with torch.no_grad():
        m.update_net.out[4].weight.mul_(3.0)
    C, a = m.num_channels, m.living_channel_dim
    torch.manual_seed(2)
    goal = torch.rand(2, 3, 32, 32)
    x_seed = m.generate_seed(2)
    x_rand = torch.rand(2, C, 32, 32)
    T = 8
    out = {}
    with torch.no_grad():
        genc = m.encoder(goal)
        gpad = torch.nn.functional.pad(genc, (0, 0, 0, 0, C - 8, 0))
        for tag, x0 in (("seed", x_seed), ("rand", x_rand)):
            us, states = [], []
            x = x0
            for t in range(T):
                torch.manual_seed(100 + t); us.append(torch.rand_like(x[:, 0:1]))
                torch.manual_seed(100 + t); x, _ = m.forward((x, gpad))
                states.append(x)
            out[f"{tag}_x0"] = x0
            out[f"{tag}_us"] = torch.stack(us)
            out[f"{tag}_states"] = torch.stack(states)
            torch.manual_seed(55)
            out[f"{tag}_grow55"] = m.grow(x0, T, goal)  # global-RNG stream contract
    save("g2_cond_grow", goal=goal, genc=genc, alive_ch=a, thr=0.1, fire_rate=0.5, T=T, **out, **sd_np(m))


# ---------------------------------------------------------------- G2L: BASELINE cfg1 exact
def g2l_cfg1():
    torch.manual_seed(0)
    m = ref_nca.ConditionedNCA(target_shape=(3, 128, 128), num_hidden_channels=8, living_channel_dim=3)
    C = m.num_channels
    torch.manual_seed(1234)
    x0 = torch.rand(4, C, 128, 128)
    goal = torch.rand(4, 3, 128, 128)
    T = 32
    sums, asums, nalive = [], [], []
    with torch.no_grad():
        genc = m.encoder(goal)
        gpad = torch.nn.functional.pad(genc, (0, 0, 0, 0, C - 8, 0))
        x = x0
        torch.manual_seed(99)
        for t in range(T):
            x, _ = m.forward((x, gpad))
            sums.append(float(x.double().sum())); asums.append(float(x.double().abs().sum()))
            nalive.append(int(m.alive(x).sum()))
    save("g2l_cfg1", crop=x[:, :, 56:72, 56:72], final_sum=sums[-1], sums=sums, asums=asums, nalive=nalive,
         T=T, data_seed=1234, rng_seed=99, **sd_np(m))


# ---------------------------------------------------------------- G3: DyNCA steps, pad x conditioning
def g3_dynca():
    arrs = {}
    cases = []
    k = 0
    for (C, fc) in ((12, 96), (16, 128)):
        for pad in ("replicate", "circular", "reflect", "constant"):
            for cond, tr in (("edges", "tanh"), ("edges", "None"), ("pos_emb", None), ("none", None)):
                if C == 16 and (pad in ("reflect", "constant") or cond == "pos_emb"):
                    continue  # keep the fixture small: C=16 covers replicate/circular x edges/none
                torch.manual_seed(10 + k)
                m = ref_dynca.DyNCA(C, 3, fc_dim=fc, padding_mode=pad, conditioning=cond,
                                    edge_transform=tr, device=CPU)
                rand_biases_(m)
                B, H, W = 2, 12, 16  # non-square on purpose
                x0 = torch.rand(B, C, H, W) - 0.5
                cimg = torch.rand(B, 1, H, W) * 2 - 1 if cond == "edges" else None
                T = 6
                us, states = [], []
                with torch.no_grad():
                    x = x0
                    for t in range(T):
                        torch.manual_seed(500 + t); us.append(torch.rand(B, 1, H, W))
                        torch.manual_seed(500 + t); x, rgb = m(x, update_rate=0.5, cond_img=cimg)
                        states.append(x)
                    torch.manual_seed(77)
                    xn, rgbn = m.forward_nsteps(x0, T, update_rate=0.7, cond_img=cimg)
                    y0 = m.perceive_torch(x0)
                tag = f"c{k}"
                cases.append(dict(tag=tag, C=C, fc=fc, pad=pad, cond=cond, transform=tr, T=T))
                arrs.update({f"{tag}.x0": x0, f"{tag}.us": torch.stack(us), f"{tag}.state_first": states[0], f"{tag}.state_last": states[-1],
                             f"{tag}.nsteps77_rate07": xn, f"{tag}.rgb77": rgbn, f"{tag}.perc0": y0,
                             f"{tag}.w1.weight": m.w1.weight.detach(), f"{tag}.w1.bias": m.w1.bias.detach(),
                             f"{tag}.w2.weight": m.w2.weight.detach(), f"{tag}.w2.bias": m.w2.bias.detach()})
                if cimg is not None:
                    arrs[f"{tag}.cond_img"] = cimg
                    with torch.no_grad():
                        arrs[f"{tag}.cond"] = m.cond_layer(cimg)
                elif cond == "pos_emb":
                    arrs[f"{tag}.cond"] = m.cond_layer(x0)
                k += 1
    save("g3_dynca", cases=json.dumps(cases), **arrs)


# ---------------------------------------------------------------- G4: perception only, ramps / impulses
def g4_perception():
    arrs = {}
    ramp = torch.arange(25, dtype=torch.float32).reshape(1, 1, 5, 5).repeat(1, 2, 1, 1)
    imp = torch.zeros(1, 2, 5, 5); imp[0, 0, 0, 0] = 1.0; imp[0, 1, 2, 4] = 1.0
    for pad in ("replicate", "circular", "reflect", "constant"):
        m = ref_dynca.DyNCA(2, 2, fc_dim=8, padding_mode=pad, conditioning="none", device=CPU)
        with torch.no_grad():
            arrs[f"ramp.{pad}"] = m.perceive_torch(ramp)
            arrs[f"imp.{pad}"] = m.perceive_torch(imp)
    torch.manual_seed(3)
    m = ref_nca.ConditionedNCA(target_shape=(3, 8, 8), num_hidden_channels=4, living_channel_dim=3)
    x = torch.zeros(1, m.num_channels, 8, 8); x[0, 2, 4, 4] = 1.0
    with torch.no_grad():
        arrs["cond_imp_ch2"] = m.perception_net(x)
    arrs["cond_wp"] = m.perception_net.weight.detach()
    # multi-scale perception (dynca.py:102-115), scales [0,1] as the 256^2 video models use
    torch.manual_seed(4)
    m2 = ref_dynca.DyNCA(4, 3, fc_dim=8, padding_mode="replicate", conditioning="none",
                         perception_scales=[0, 1], device=CPU)
    xs = torch.rand(1, 4, 16, 16) - 0.5
    with torch.no_grad():
        arrs["ms_x"] = xs
        arrs["ms_y"] = m2.perceive_multiscale(xs)
    save("g4_perception", ramp=ramp, imp=imp, **arrs)


# ---------------------------------------------------------------- G5: trained weights from the web demo
def decode_webgl_layer(layer):
    rows, cols = layer["shape"]
    a = np.asarray(layer["data_flatten"], dtype=np.float64).reshape(layer["data_shape"])
    a = a.reshape(rows, -1)[:, :cols]
    return ((a - layer["center"]) * layer["scale"]).astype(np.float32)


def g5_real_weights():
    path = os.path.join(REF, "docs/data/vec_field_models/large/starry-night.json")
    js = json.load(open(path))
    l1, l2 = (decode_webgl_layer(l) for l in js["layers"])
    w1 = torch.from_numpy(l1[:-1].T.copy())[:, :, None, None]; b1 = torch.from_numpy(l1[-1].copy())
    w2 = torch.from_numpy(l2[:-1].T.copy())[:, :, None, None]; b2 = torch.from_numpy(l2[-1].copy())
    m = ref_dynca.DyNCA(12, 3, fc_dim=96, padding_mode="circular", conditioning="edges",
                        edge_transform="tanh", device=CPU)
    with torch.no_grad():
        m.w1.weight.copy_(w1); m.w1.bias.copy_(b1); m.w2.weight.copy_(w2); m.w2.bias.copy_(b2)
    H = W = 48
    yy, xx = np.mgrid[0:H, 0:W]
    disc = ((yy - H / 2) ** 2 + (xx - W / 2) ** 2 < (H / 3) ** 2).astype(np.float32) * 2 - 1
    cimg = torch.from_numpy(disc)[None, None]
    x = m.seed(1, size=(W, H))
    crops = {}
    stats = []
    with torch.no_grad():
        torch.manual_seed(2024)
        for t in range(1, 101):
            x, rgb = m(x, update_rate=0.5, cond_img=cimg)
            if t in (25, 50, 100):
                crops[f"x_t{t}"] = x.clone()
            stats.append(float(x.abs().max()))
    save("g5_real_weights", w1=w1, b1=b1, w2=w2, b2=b2, cond_img=cimg, rng_seed=2024, absmax=stats, **crops)


# ---------------------------------------------------------------- G10: shipped two-scale video model (perception_scales = [0, 1])
def g10_two_scale_video_model():
    """DyNCA(perception_scales=[0, 1], conditioning='pos_emb') with the trained weights of a shipped video model
    (docs/data/video_models/small/fountain_1.json: C = 12, fc = 96, n_perception_scales = 2), the reference's own forward at an
    even, non-square size, plus a random-weight case at C = 16 / fc = 128 with each pad mode.  The raw layer tables of the JSON
    are kept in the fixture as data so the WebGL importer can be checked against the same numbers."""
    path = os.path.join(REF, "docs/data/video_models/small/fountain_1.json")
    js = json.load(open(path))
    l1, l2 = (decode_webgl_layer(l) for l in js["layers"])
    w1 = torch.from_numpy(l1[:-1].T.copy())[:, :, None, None]; b1 = torch.from_numpy(l1[-1].copy())
    w2 = torch.from_numpy(l2[:-1].T.copy())[:, :, None, None]; b2 = torch.from_numpy(l2[-1].copy())
    out = {"n_perception_scales": int(js["n_perception_scales"]), "w1": w1, "b1": b1, "w2": w2, "b2": b2}
    for i, l in enumerate(js["layers"]):     # raw tables (data): what ncahip.webgl.decode_layer must turn into w/b above
        out[f"json.l{i}.data"] = np.asarray(l["data_flatten"], dtype=np.float64)
        out[f"json.l{i}.meta"] = json.dumps({k: v for k, v in l.items() if k != "data_flatten"})
    m = ref_dynca.DyNCA(12, 3, fc_dim=96, padding_mode="circular", conditioning="pos_emb", perception_scales=[0, 1], device=CPU)
    with torch.no_grad():
        m.w1.weight.copy_(w1); m.w1.bias.copy_(b1); m.w2.weight.copy_(w2); m.w2.bias.copy_(b2)
    H, W, T = 32, 48, 24
    gen = torch.Generator().manual_seed(7)
    x0 = (torch.rand(1, 12, H, W, generator=gen) - 0.5) * 0.2
    x = x0.clone()
    with torch.no_grad():
        torch.manual_seed(77)
        us = torch.stack([torch.rand(1, 1, H, W) for _ in range(T)])
        torch.manual_seed(77)
        for t in range(T):
            x, rgb = m(x, update_rate=0.5)
            if t in (0, 7, 23):
                out[f"vid.x_t{t + 1}"] = x.clone()
        out["vid.perc0_crop"] = m.perceive_multiscale(x0, m.cond_layer(x0))[:, :, :12, -12:]
    out.update({"vid.x0": x0, "vid.us": us})
    cases = []
    for k, pad in enumerate(["replicate", "circular", "reflect", "constant"]):
        torch.manual_seed(40 + k)
        cond = "edges" if k % 2 == 0 else "pos_emb"
        mm = ref_dynca.DyNCA(16, 3, fc_dim=128, padding_mode=pad, conditioning=cond, edge_transform="tanh",
                             perception_scales=[0, 1], device=CPU)
        rand_biases_(mm)
        B, Hh, Ww, Tn = 2, 16, 24, 4
        xx0 = torch.rand(B, 16, Hh, Ww) - 0.5
        cimg = torch.rand(B, 1, Hh, Ww) * 2 - 1 if cond == "edges" else None
        with torch.no_grad():
            torch.manual_seed(900 + k)
            uu = torch.stack([torch.rand(B, 1, Hh, Ww) for _ in range(Tn)])
            torch.manual_seed(900 + k)
            xx, first = xx0, None
            for t in range(Tn):
                xx, _ = mm(xx, update_rate=0.5, cond_img=cimg)
                first = xx.clone() if t == 0 else first
        tag = f"ms{k}"
        cases.append({"tag": tag, "pad": pad, "cond": cond, "T": Tn})
        for kk, vv in mm.state_dict().items():
            if kk.startswith(("w1", "w2")):
                out[f"{tag}.{kk}"] = vv.detach()
        out.update({f"{tag}.x0": xx0, f"{tag}.us": uu, f"{tag}.first": first, f"{tag}.last": xx})
        if cimg is not None:
            out[f"{tag}.cond_img"] = cimg
    out["cases"] = json.dumps(cases)
    out.update({"g10_final": os.path.basename(path)})
    save("g10_two_scale", **out)


# ---------------------------------------------------------------- G6: ExtraChannels variant
def g6_extra_channels():
    torch.manual_seed(21)
    m = ref_dynca_x.DyNCA(13, 3, fc_dim=96, padding_mode="replicate", pos_emb="CPE", device=CPU)
    rand_biases_(m)
    B, H, W = 2, 20, 28
    with torch.no_grad():
        seed = m.seed(B, size=(W, H))              # 12 channels (c_in-1), ExtraChannels dynca.py:139-150
        cimg = torch.rand(B, 1, H, W)
        x0 = torch.cat([seed + 0.1 * torch.randn_like(seed), cimg], dim=1)
        us, states = [], []
        x = x0
        for t in range(5):
            torch.manual_seed(900 + t); us.append(torch.rand(B, 1, H, W))
            torch.manual_seed(900 + t); x, rgb = m(x)
            states.append(x)
        pe = m.pos_emb_2d(x0)
    save("g6_extra_channels", x0=x0, us=torch.stack(us), states=torch.stack(states), pos_emb=pe,
         seed_shape=np.array(seed.shape), **{"w1.weight": m.w1.weight.detach(), "w1.bias": m.w1.bias.detach(),
                                             "w2.weight": m.w2.weight.detach(), "w2.bias": m.w2.bias.detach()})


# ---------------------------------------------------------------- G7: conditioning encoders
def g7_encoders():
    torch.manual_seed(31)
    enc = sys.modules["encoder"].ImageEncoder(8, 3)
    img = torch.rand(2, 3, 20, 24)
    gray = torch.rand(2, 1, 20, 24) * 2 - 1
    with torch.no_grad():
        e = enc(img)
        ee_t = ref_dynca.EdgeExtractor("tanh")(gray)
        ee_n = ref_dynca.EdgeExtractor("None")(gray)
        pe = ref_dynca.CPE2D()(torch.zeros(2, 5, 20, 24))
    save("g7_encoders", img=img, gray=gray, enc_out=e, edges_tanh=ee_t, edges_none=ee_n, cpe=pe,
         **{"sd.encoder." + k: v.detach().numpy() for k, v in enc.state_dict().items()})


# ---------------------------------------------------------------- G8: gradients through T steps
def g8_grads():
    # ConditionedNCA: d<cot, grow(x0)> / d{x0, goal_enc, weights}
    torch.manual_seed(0)
    m = ref_nca.ConditionedNCA(target_shape=(3, 16, 16), num_hidden_channels=8, living_channel_dim=3)
    rand_biases_(m.update_net)
    with torch.no_grad():
        m.update_net.out[4].weight.mul_(3.0)
    C = m.num_channels
    torch.manual_seed(5)
    x0 = torch.rand(2, C, 16, 16).requires_grad_(True)
    x0.data[0, :, :5] = 0.0                     # some dead cells: life-mask gradient gating is exercised
    x0.data[1, :, 8:10, 8:10] *= 40.0           # clamp gradient gating
    gpad = torch.nn.functional.pad(torch.randn(2, 8, 16, 16), (0, 0, 0, 0, C - 8, 0)).requires_grad_(True)
    cot = torch.randn(2, C, 16, 16)
    T = 4
    us = []
    x = x0
    for t in range(T):
        torch.manual_seed(300 + t); us.append(torch.rand_like(x[:, 0:1]))
        torch.manual_seed(300 + t); x, _ = m.forward((x, gpad))
    (x * cot).sum().backward()
    cond = dict(x0=x0.detach(), gpad=gpad.detach(), cot=cot, us=torch.stack(us), xT=x.detach(),
                d_x0=x0.grad, d_gpad=gpad.grad)
    for n, p in m.named_parameters():
        if p.grad is not None:
            cond["grad." + n] = p.grad
    cond.update(sd_np(m))
    save("g8_cond_grads", alive_ch=3, thr=0.1, fire_rate=0.5, T=T, **cond)

    # DyNCA
    arrs = {}
    for k, pad in enumerate(("replicate", "circular", "reflect", "constant")):
        torch.manual_seed(40 + k)
        d = ref_dynca.DyNCA(12, 3, fc_dim=96, padding_mode=pad, conditioning="edges", edge_transform="tanh", device=CPU)
        rand_biases_(d)
        B, H, W = 2, 12, 16
        x0 = (torch.rand(B, 12, H, W) - 0.5).requires_grad_(True)
        cimg = torch.rand(B, 1, H, W) * 2 - 1
        cot = torch.randn(B, 12, H, W)
        cot_rgb = torch.randn(B, 3, H, W)
        us = []
        x = x0
        for t in range(T):
            torch.manual_seed(700 + t); us.append(torch.rand(B, 1, H, W))
            torch.manual_seed(700 + t); x, rgb = d(x, cond_img=cimg)
        ((x * cot).sum() + (rgb * cot_rgb).sum()).backward()
        arrs.update({f"{pad}.x0": x0.detach(), f"{pad}.cond_img": cimg, f"{pad}.cot": cot, f"{pad}.cot_rgb": cot_rgb,
                     f"{pad}.us": torch.stack(us), f"{pad}.xT": x.detach(), f"{pad}.d_x0": x0.grad,
                     f"{pad}.w1.weight": d.w1.weight.detach(), f"{pad}.w1.bias": d.w1.bias.detach(),
                     f"{pad}.w2.weight": d.w2.weight.detach(), f"{pad}.w2.bias": d.w2.bias.detach(),
                     f"{pad}.g.w1.weight": d.w1.weight.grad, f"{pad}.g.w1.bias": d.w1.bias.grad,
                     f"{pad}.g.w2.weight": d.w2.weight.grad, f"{pad}.g.w2.bias": d.w2.bias.grad})
    save("g8_dynca_grads", T=T, **arrs)


# ---------------------------------------------------------------- G11: gradients of the reference's DEFAULT model (C = 20) and C = 32
def g11_cond_grads_wide():
    """ConditionedNCA with its default arguments (nca.py:62-74: num_hidden_channels = 16 -> C = 3 + 16 + 1 = 20) and a
    28-hidden-channel one (C = 32): d<cot, grow(x0, T, goal)> / d{x0, goal, weights, encoder.embed} through the reference's
    own grow (nca.py:197-209, encoder included) under the reference's autograd."""
    arrs = {}
    for tag, hidden, T, seed in (("c20", 16, 4, 11), ("c32", 28, 3, 12)):
        torch.manual_seed(seed)
        m = ref_nca.ConditionedNCA(target_shape=(3, 16, 16), num_hidden_channels=hidden)
        rand_biases_(m.update_net)
        with torch.no_grad():
            m.update_net.out[4].weight.mul_(3.0)
        C = m.num_channels
        torch.manual_seed(seed + 100)
        x0 = torch.rand(2, C, 16, 16).requires_grad_(True)
        x0.data[0, :, :5] = 0.0                     # dead cells
        x0.data[1, :, 8:10, 8:10] *= 40.0           # clamp gating
        goal = torch.rand(2, 3, 16, 16)
        cot = torch.randn(2, C, 16, 16)
        genc = m.encode(goal)                       # [B, hidden, H, W]
        genc.retain_grad()
        gpad = torch.nn.functional.pad(genc, (0, 0, 0, 0, C - hidden, 0))
        us = []
        x = x0
        for t in range(T):
            torch.manual_seed(900 + t); us.append(torch.rand_like(x[:, 0:1]))
            torch.manual_seed(900 + t); x, _ = m.forward((x, gpad))
        # grow() is encode + pad + T x forward (nca.py:197-209): checked here against the loop above's pieces under one seed
        with torch.no_grad():
            torch.manual_seed(77); xa = m.grow(x0.detach(), T, goal)
            torch.manual_seed(77); xb = x0.detach()
            for t in range(T):
                xb, _ = m.forward((xb, gpad.detach()))
        assert torch.equal(xa, xb)
        (x * cot).sum().backward()
        arrs.update({f"{tag}.x0": x0.detach(), f"{tag}.goal_img": goal, f"{tag}.goal_enc": genc.detach(), f"{tag}.cot": cot,
                     f"{tag}.us": torch.stack(us), f"{tag}.xT": x.detach(), f"{tag}.d_x0": x0.grad, f"{tag}.d_goal_enc": genc.grad,
                     f"{tag}.T": T, f"{tag}.alive_ch": m.living_channel_dim})
        for n, p in m.named_parameters():
            if p.grad is not None:
                arrs[f"{tag}.grad." + n] = p.grad
        arrs.update({f"{tag}." + k: v for k, v in sd_np(m).items()})
    save("g11_cond_grads_wide", thr=0.1, fire_rate=0.5, **arrs)


# ---------------------------------------------------------------- G9: seeds
def g9_seeds():
    m = ref_nca.ConditionedNCA(target_shape=(3, 16, 16), num_hidden_channels=8, living_channel_dim=3)
    arrs = {"cond_seed": m.generate_seed(2), "cond_seed_dev": m.generate_seed(1, device=CPU, size=10)}
    for mode in ("zeros", "center_on", "random"):
        d = ref_dynca.DyNCA(6, 3, fc_dim=8, seed_mode=mode, conditioning="none", device=CPU)
        arrs[f"dynca_seed.{mode}"] = d.seed(2, size=(10, 6))
    save("g9_seeds", **arrs)


if __name__ == "__main__" and len(sys.argv) > 1:      # regenerate selected fixtures only:  gen_golden.py g10_two_scale_video_model
    for _n in sys.argv[1:]:
        globals()[_n]()
    sys.exit(0)

if __name__ == "__main__":
    with torch.no_grad():
        pass
    g1_cond_step(); g2_cond_grow(); g2l_cfg1(); g3_dynca(); g4_perception(); g5_real_weights()
    g6_extra_channels(); g7_encoders(); g8_grads(); g9_seeds(); g10_two_scale_video_model(); g11_cond_grads_wide()
