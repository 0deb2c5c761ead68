"""ConditionedNCA / UpdateNet drop-ins (reference: EncoderConditioning/nca.py:29-215).

Same constructor signatures, attribute names, method names and state_dict keys as the reference, so
`from nca import ConditionedNCA` can be swapped for `from ncahip.nca import ConditionedNCA`
(INTEGRATION.md).  forward()/grow()/update()/alive() run on the hand-written HIP kernels of
libncahip.so (include/ncahip.h) -- there is no eager-PyTorch or CPU path for them.

Step semantics (nca.py:181-195), per cell, fp32:
    pre  = maxpool3x3(alpha) > thr ; z = x + goal*pre ; p = depthwise3x3(z) ; out = W3 relu(W2 relu(W1 p+b1)+b2)
    x'   = x + (u < fire_rate) * out ; post = maxpool3x3(alpha') > thr ; x'' = clamp(x' * (pre & post), -10, 10)
"""
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .autograd import cond_grow_autograd
from .encoder import ImageEncoder


class UpdateNet(nn.Module):
    """1x1 conv (in->64) ReLU 1x1 conv (64->64) ReLU 1x1 conv (64->out, no bias); `out` indices 0,2,4."""

    def __init__(self, in_channels: int, out_channels: int, zero_bias: bool = True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.out = nn.Sequential(nn.Conv2d(in_channels, 64, 1), nn.ReLU(), nn.Conv2d(64, 64, 1), nn.ReLU(),
                                 nn.Conv2d(64, out_channels, 1, bias=False))
        if zero_bias:
            with torch.no_grad():
                for m in self.out:
                    if isinstance(m, nn.Conv2d) and m.bias is not None:
                        m.bias.zero_()

    def forward(self, x):  # not used by ConditionedNCA.forward (fused in the step kernel); kept for API parity
        return self.out(x)


class ConditionedNCA(nn.Module):
    def __init__(self, encoder: Optional[nn.Module] = None, target_shape: Tuple[int] = (3, 64, 64),
                 num_hidden_channels=16, use_living_channel: bool = True, living_channel_dim: Optional[int] = None,
                 alpha_living_threshold: float = 0.1, cell_fire_rate: float = 0.5, zero_bias=True):
        super().__init__()
        n_target, hidden = target_shape[0], num_hidden_channels
        # geometry: state = [target channels | alpha | hidden]; the goal encoding rides on the hidden channels
        self.target_shape, self.image_size = target_shape, target_shape[-1]
        self.num_target_channels, self.num_hidden_channels = n_target, hidden
        self.num_channels = n_target + 1 + hidden
        self.living_channel_dim = n_target if living_channel_dim is None else living_channel_dim
        self.use_living_channel, self.alpha_living_threshold = use_living_channel, alpha_living_threshold
        self.cell_fire_rate, self.zero_bias = cell_fire_rate, zero_bias
        # parameters, under the reference's module names (state_dict compatibility, nca.py:99-110)
        width = self.num_channels
        self.perception_net = nn.Conv2d(width, 3 * width, kernel_size=3, padding=1, groups=width, bias=False)
        self.update_net = UpdateNet(3 * width, width, zero_bias)
        self.encoder = ImageEncoder(hidden, n_target) if encoder is None else encoder
        # fire-mask source: 'torch' = one torch.rand_like(x[:, 0:1]) per step on x's device (the reference's RNG contract,
        # nca.py:172); 'philox' = drawn inside the kernel, keyed by (mask_seed, running step counter, cell)
        self.mask_rng, self.mask_seed, self._mask_step = "torch", 0, 0

    # ------------------------------------------------------------------ helpers
    def _alive_ch(self) -> int:
        return self.living_channel_dim if self.use_living_channel else -1

    def _weights(self, like: torch.Tensor) -> "ops.CondWeights":
        u = self.update_net.out
        return ops.CondWeights(self.perception_net.weight, u[0].weight, u[0].bias, u[2].weight, u[2].bias,
                               u[4].weight, like)

    def _draw(self, x: torch.Tensor, steps: int) -> Optional[torch.Tensor]:
        if self.mask_rng == "philox":
            return None
        # one draw per step, as nca.py:207-208 (always float32: identical to the reference for float32 states)
        if not x.is_cuda:
            return torch.stack([torch.rand_like(x[:, 0:1], dtype=torch.float32) for _ in range(steps)])
        # ... evaluated to the fire mask right away and kept as BITS (what the backward re-reads: 1/32 of the float draws)
        return ops.draw_fire_masks(x.shape[0], x.shape[2], x.shape[3], steps, self.cell_fire_rate, "cond", x.device)

    def _split_goal(self, goal_encoding: torch.Tensor) -> torch.Tensor:
        """The kernels take the UNPADDED encoding; nca.py:199-203 pads zeros in front -- strip them."""
        C, hid = self.num_channels, self.num_hidden_channels
        if goal_encoding.size(1) == C and C != hid:
            return goal_encoding[:, C - hid:]
        return goal_encoding

    # ------------------------------------------------------------------ reference surface
    def encode(self, images: torch.Tensor):
        return self.encoder(images)

    def generate_seed(self, num_seeds, device: Optional[torch.device] = None, size: Optional[int] = None):
        # nca.py:130-150 -- note the reference forces CPU whenever a device is passed (:136-137); callers .to() it
        size = self.image_size if size is None else size
        seed = torch.zeros(num_seeds, self.num_channels, size, size)
        seed[:, self.living_channel_dim:, size // 2, size // 2] = 1.0
        return seed

    def alive(self, x):
        if not self.use_living_channel:
            return torch.ones_like(x, dtype=torch.bool)
        if x.is_cuda:
            return ops.cond_alive(x.float(), self.living_channel_dim, self.alpha_living_threshold)
        a = self.living_channel_dim  # host-side bookkeeping on CPU tensors (pool inspection), not the step path
        return F.max_pool2d(x[:, a:a + 1], 3, 1, 1) > self.alpha_living_threshold

    def get_stochastic_update_mask(self, x):
        return (torch.clamp(torch.rand_like(x[:, 0:1]), 0.0, 1.0).float() < self.cell_fire_rate).float()

    def update(self, x, goal_encoding, pre_life_mask):
        """perception + UpdateNet only (nca.py:176-179): the fused step with fire_rate=1, no alive logic."""
        w = self._weights(x)
        z = (x + goal_encoding * pre_life_mask).contiguous()
        xp, _ = ops.cond_step(z, None, None, torch.zeros_like(x[:, 0:1]), w, alive_ch=-1, fire_rate=1.0,
                              lo=-float("inf"), hi=float("inf"))
        return xp - z

    def forward(self, x):
        x, goal_encoding = x[0], x[1]
        out = cond_grow_autograd(self, x, self._split_goal(goal_encoding), 1)
        return out, goal_encoding

    def grow(self, x: torch.Tensor, num_steps: int, goal: torch.Tensor) -> torch.Tensor:
        goal_encoding = self.encoder(goal)  # once per grow (nca.py:198); padding (:199-203) is implicit in the kernel
        return cond_grow_autograd(self, x, goal_encoding, num_steps)

    def save(self, path: str):
        torch.save(self.state_dict(), path)

    def load(self, path: str):
        self.load_state_dict(torch.load(path, weights_only=True))
