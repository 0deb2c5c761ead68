"""Drop-in class surface on CPU: constructor signatures, state_dict compatibility with the reference
(golden state dicts load strictly), conditioning encoders vs golden outputs, seeds, pool semantics."""
import inspect

import numpy as np
import pytest
import torch

from util import T, load, sd


def test_conditioned_nca_surface_and_state_dict():
    from ncahip.nca import ConditionedNCA, UpdateNet
    sig = inspect.signature(ConditionedNCA.__init__)
    assert list(sig.parameters)[1:] == ["encoder", "target_shape", "num_hidden_channels", "use_living_channel",
                                        "living_channel_dim", "alpha_living_threshold", "cell_fire_rate", "zero_bias"]
    assert list(inspect.signature(UpdateNet.__init__).parameters)[1:] == ["in_channels", "out_channels", "zero_bias"]
    g = load("g1_cond_step")
    m = ConditionedNCA(target_shape=(3, 32, 32), num_hidden_channels=8, living_channel_dim=3)
    assert m.num_channels == 12 and m.num_target_channels == 3 and m.image_size == 32
    m.load_state_dict(sd(g), strict=True)              # the reference's own state_dict
    assert sum(p.numel() for p in m.parameters()) == 8688      # 8 688 parameters (frozen filters included)
    for meth in ("encode", "generate_seed", "alive", "get_stochastic_update_mask", "update", "forward", "grow", "save", "load"):
        assert callable(getattr(m, meth))


def test_encoder_and_edge_extractor_match_golden():
    from ncahip.encoder import ImageEncoder
    from ncahip.models.dynca import CPE2D, EdgeExtractor
    g = load("g7_encoders")
    enc = ImageEncoder(8, 3)
    enc.load_state_dict({k[len("encoder."):]: v for k, v in sd(g).items()}, strict=True)
    with torch.no_grad():
        out = enc(T(g["img"]))
    assert torch.allclose(out, T(g["enc_out"]), rtol=1e-5, atol=1e-6)
    with torch.no_grad():
        assert torch.allclose(EdgeExtractor("tanh")(T(g["gray"])), T(g["edges_tanh"]), rtol=1e-6, atol=1e-6)
        assert torch.allclose(EdgeExtractor("None")(T(g["gray"])), T(g["edges_none"]), rtol=1e-6, atol=1e-6)
    assert torch.equal(CPE2D()(torch.zeros(2, 5, 20, 24)), T(g["cpe"]))
    with pytest.raises(RuntimeError):
        CPE2D()(torch.zeros(3, 4, 5))


def test_seeds_match_golden():
    from ncahip.models.dynca import DyNCA
    from ncahip.models.dynca_extra import DyNCA as DyNCAX
    from ncahip.nca import ConditionedNCA
    g = load("g9_seeds")
    m = ConditionedNCA(target_shape=(3, 16, 16), num_hidden_channels=8, living_channel_dim=3)
    assert torch.equal(m.generate_seed(2), T(g["cond_seed"]))
    s = m.generate_seed(1, device=torch.device("cpu"), size=10)
    assert torch.equal(s, T(g["cond_seed_dev"])) and s.device.type == "cpu"
    cpu = torch.device("cpu")
    for mode in ("zeros", "center_on", "random"):
        d = DyNCA(6, 3, fc_dim=8, seed_mode=mode, conditioning="none", device=cpu)
        assert torch.equal(d.seed(2, size=(10, 6)), T(g[f"dynca_seed.{mode}"])), mode
    dx = DyNCAX(13, 3, fc_dim=96, device=cpu)
    assert dx.seed(2, size=(28, 20)).shape == (2, 12, 20, 28) and dx.c_in == 13     # c_in-1 channels
    assert dx.w1.weight.shape == (96, 4 * 13 + 2, 1, 1) and dx.pos_emb_2d is not None


def test_dynca_surface():
    from ncahip.models.dynca import DyNCA
    cpu = torch.device("cpu")
    sig = list(inspect.signature(DyNCA.__init__).parameters)[1:]
    assert sig == ["c_in", "c_out", "fc_dim", "padding_mode", "seed_mode", "conditioning", "edge_transform",
                   "perception_scales", "device"]
    d = DyNCA(12, 3, fc_dim=96, device=cpu)
    assert d.w1.weight.shape == (96, 51, 1, 1) and d.w2.weight.shape == (12, 96, 1, 1) and d.c_cond == 3
    assert float(d.w2.bias.abs().sum()) == 0.0
    assert torch.equal(d.sobel_filter_y, d.sobel_filter_x.T) and float(d.laplacian_filter[1, 1]) == -12.0
    assert torch.equal(d.to_rgb(torch.ones(1, 12, 2, 2)), 2 * torch.ones(1, 3, 2, 2))
    with pytest.raises(AssertionError):
        DyNCA(12, 3, seed_mode="bogus", device=cpu)
    from ncahip._capi import NcaHipError
    with pytest.raises(NcaHipError):          # multi-scale models run the HIP stencil too: no CPU path
        DyNCA(12, 3, perception_scales=[0, 1], device=cpu).forward(torch.zeros(1, 12, 8, 8), cond_img=torch.zeros(1, 1, 8, 8))


def test_hot_path_has_no_cpu_fallback():
    from ncahip._capi import NcaHipError
    from ncahip.models.dynca import DyNCA
    from ncahip.nca import ConditionedNCA
    m = ConditionedNCA(target_shape=(3, 16, 16), num_hidden_channels=8, living_channel_dim=3)
    with pytest.raises(NcaHipError):
        m.grow(m.generate_seed(1), 2, torch.rand(1, 3, 16, 16))
    d = DyNCA(12, 3, conditioning="none", device=torch.device("cpu"))
    with pytest.raises(NcaHipError):
        d.forward_nsteps(torch.zeros(1, 12, 8, 8), 2)


def test_sample_pool_semantics():
    from ncahip.sample_pool import SamplePool
    p = SamplePool(6)
    assert len(p) == 6 and p[0] is None and p[[0, 3]] == [None, None]       # reference: list of None
    batch = torch.arange(2 * 3 * 4 * 4, dtype=torch.float32).reshape(2, 3, 4, 4)
    p[[4, 1]] = batch                                                       # iterable setitem: value[i] rows
    assert torch.equal(p[4], batch[0]) and torch.equal(p[1], batch[1]) and p[0] is None
    got = p[[1, 0, 4]]
    assert torch.equal(got[0], batch[1]) and got[1] is None and torch.equal(got[2], batch[0])
    p[2] = batch[0] * 2
    assert torch.equal(p[2], batch[0] * 2)
    seed = torch.full((3, 4, 4), -1.0)
    g = p.gather([0, 1, 5, 4], seed)                                        # empty slots -> seed, one index_select
    assert torch.equal(g[0], seed) and torch.equal(g[1], batch[1]) and torch.equal(g[2], seed) and torch.equal(g[3], batch[0])
    p.scatter([0, 5], g[:2] + 1)
    assert torch.equal(p[0], seed + 1) and torch.equal(p[5], batch[1] + 1)
    assert [x is None for x in p.pool] == [False, False, False, True, False, False]


def test_webgl_json_roundtrip_and_layout():
    """export_dynca_json -> load_dynca_weights reproduces the weights to the texture's float precision, for one model and
    for several tiled in one texture (convert_models_to_webgl.ipynb cell 1 / docs/dynca.js:827-872)."""
    import torch.nn as nn
    from ncahip import webgl

    class Stub(nn.Module):
        def __init__(self, seed, k1=51, fc=96, c=12):
            super().__init__()
            g = torch.Generator().manual_seed(seed)
            self.w1, self.w2 = nn.Conv2d(k1, fc, 1), nn.Conv2d(fc, c, 1)
            self.conditioning = "edges"
            with torch.no_grad():
                for p in self.parameters():
                    p.copy_(torch.randn(p.shape, generator=g) * 0.3)

    ms = [Stub(s) for s in range(3)]
    js = webgl.export_dynca_json(ms, ["a", "b", "c"])
    assert js["layers"][0]["shape"] == [52, 96] and js["layers"][1]["shape"] == [97, 12]      # the demo's own shapes
    assert js["layers"][0]["edge_conditioning"] and not js["layers"][1]["edge_conditioning"]
    for i, m in enumerate(ms):
        w = webgl.load_dynca_weights(js, index=i)
        for k in ("w1.weight", "w1.bias", "w2.weight", "w2.bias"):
            ref = dict(m.named_parameters())[k].detach()
            assert torch.allclose(w[k], ref, atol=2e-6), (i, k)
    with pytest.raises(IndexError):
        webgl.decode_layer(js["layers"][0], index=7)
