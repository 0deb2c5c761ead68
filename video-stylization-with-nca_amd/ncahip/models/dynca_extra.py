"""DyNCA, "conditioning image as an extra state channel" variant
(reference: ExtraChannels/models/dynca.py:7-167).  Same fused step kernel with c_cond = 2 (CPE) or 0;
the caller concatenates the conditioning image to the state, and seed() emits c_in-1 channels."""
import numpy as np
import torch
import torch.nn as nn

from ..autograd import dynca_nsteps_autograd
from .dynca import CPE2D, DyNCA as _EdgeDyNCA


class DyNCA(_EdgeDyNCA):
    def __init__(self, c_in, c_out, fc_dim=96, padding_mode="replicate", seed_mode="zeros", pos_emb="CPE",
                 perception_scales=[0], device=torch.device("cuda:0")):
        super().__init__(c_in, c_out, fc_dim=fc_dim, padding_mode=padding_mode, seed_mode=seed_mode,
                         conditioning="pos_emb" if pos_emb == "CPE" else "none", edge_transform=None,
                         perception_scales=perception_scales, device=device)
        self.pos_emb = pos_emb
        self.pos_emb_2d = self.cond_layer  # reference attribute name (ExtraChannels dynca.py:50-54)
        if self.cond_layer is not None:
            del self.cond_layer
            self.cond_layer = None
        self.conditioning = "pos_emb" if self.pos_emb_2d is not None else "none"

    def _cond(self, x, cond_img=None):
        return None if self.pos_emb_2d is None else self.pos_emb_2d(x).float().contiguous()

    def forward(self, x, update_rate=0.5, return_perception=False):
        return super().forward(x, update_rate, return_perception, None)

    def forward_nsteps(self, input_state, step_n, update_rate=0.5, return_middle_feature=False):
        return super().forward_nsteps(input_state, step_n, update_rate, return_middle_feature, None)

    def seed(self, n, size=128):
        c, self.c_in = self.c_in, self.c_in - 1  # seed() makes c_in-1 channels (ExtraChannels dynca.py:139-150)
        try:
            return super().seed(n, size)
        finally:
            self.c_in = c
