#!/usr/bin/env python3
"""Build-container-only check (needs /root/reference): the CPU restatement against the reference's own modules on seeded
RANDOM shapes, beyond the fixed golden vectors of tests/golden -- odd sizes, 1-pixel rows, every pad mode, alive on / off.
Prints the worst mismatch; exits non-zero above 0 (the restatement is meant to be bit-identical on CPU).

    python oracle/check_against_reference.py [cases]
"""
import importlib.util
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("NCA_REFERENCE", "/root/reference")
sys.path[:0] = [ROOT, os.path.join(REF, "EncoderConditioning")]
from oracle import nca_oracle as O  # noqa: E402
import nca as ref_nca  # noqa: E402  (reference module)


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


ref_dynca = _load(os.path.join(REF, "ConditioneDyNCA/models/dynca.py"), "ref_dynca_cond")
torch.set_num_threads(1)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.RandomState(2024)
worst = 0.0
for case in range(cases):
    # ---- ConditionedNCA.grow (nca.py:197-209)
    hid = int(rng.choice([4, 8, 12])); C = 4 + hid
    B = int(rng.randint(1, 3)); S = int(rng.randint(3, 30)); T = int(rng.randint(1, 5))
    use_alive = bool(rng.rand() < 0.7)
    torch.manual_seed(case)
    m = ref_nca.ConditionedNCA(target_shape=(3, S, S), num_hidden_channels=hid, living_channel_dim=3,
                               use_living_channel=use_alive, zero_bias=False)
    x = torch.rand(B, C, S, S) * 1.2 - 0.1
    x[:, 3] = torch.rand(B, S, S) * 0.5
    img = torch.rand(B, 3, S, S)
    prm = {k: v.detach().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        torch.manual_seed(1000 + case)
        ref = m.grow(x.clone(), T, img)
        enc = m.encoder(img)
        gpad = O.cond_pad_goal(enc, C)
        torch.manual_seed(1000 + case)
        _ = m.encoder(img) if False else None
        mine = O.cond_grow_rng(x.clone(), gpad, T, prm, 3, 0.1, 0.5) if use_alive else None
    if use_alive:
        e = float((mine - ref).abs().max())
        worst = max(worst, e)
        if e > 0:
            print("cond case", case, (B, C, S, T), "max abs diff", e)
    # ---- DyNCA.forward_nsteps (dynca.py:168-178)
    Cd, fc = [(12, 96), (16, 128), (8, 32)][int(rng.randint(0, 3))]
    H = int(rng.randint(2, 24)); W = int(rng.randint(2, 28)); Td = int(rng.randint(1, 4))
    pad = ["replicate", "circular", "reflect", "constant"][int(rng.randint(0, 4))]
    if pad == "reflect" and min(H, W) < 2:
        pad = "replicate"
    torch.manual_seed(case)
    d = ref_dynca.DyNCA(c_in=Cd, c_out=3, fc_dim=fc, padding_mode=pad, conditioning="edges", edge_transform="tanh",
                        device=torch.device("cpu"))
    xd = torch.rand(B, Cd, H, W) - 0.5
    ci = torch.rand(B, 1, H, W) * 2 - 1
    dp = {k: v.detach().clone() for k, v in d.state_dict().items()}
    with torch.no_grad():
        torch.manual_seed(2000 + case)
        refd = d.forward_nsteps(xd.clone(), Td, update_rate=0.5, cond_img=ci)
        refd = refd[0] if isinstance(refd, (tuple, list)) else refd
        torch.manual_seed(2000 + case)
        mined = O.dynca_nsteps_rng(xd.clone(), O.edge_extractor(ci, "tanh"), Td, dp, pad, 0.5)
    e = float((mined - refd).abs().max())
    worst = max(worst, e)
    if e > 0:
        print("dynca case", case, (B, Cd, fc, H, W, pad, Td), "max abs diff", e)
print("cases %d  worst |oracle - reference| = %.3e" % (cases, worst))
sys.exit(0 if worst <= 1e-6 else 1)
