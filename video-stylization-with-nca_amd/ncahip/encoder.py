"""ImageEncoder drop-in (reference: EncoderConditioning/encoder.py:5-64).

Conditioning precompute for ConditionedNCA.grow (runs ONCE per grow, nca.py:198): grayscale ->
Sobel-x / Sobel-y / Laplacian, per-channel 5x5 Gaussian blur (sigma 1), then a learned
3x3 conv -> ReLU -> 3x3 conv giving `embedding_dim` channels per pixel.  This is the "next" row f1
of SURVEY.md section 8 (adjacent to the hot path): on the device the fixed-filter front (gray, three 3x3
filters, per-channel blur) is ONE hand-written HIP pass (ncahip_image_encoder_front_f32, no gradient
needed: the target image is data); the learned `embed` convolutions stay on PyTorch-ROCm ops (autograd
flows into them).  CPU tensors (host-side tests) take the equivalent torch ops.  state_dict keys match the reference
(sobel_x/sobel_y/gaussian_blur/laplacian .weight frozen, embed.0.{weight,bias}, embed.2.weight).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class _FixedFilter(nn.Module):
    """Holds one frozen [1,1,k,k] filter under the attribute name `weight` (state_dict compatibility)."""

    def __init__(self, taps):
        super().__init__()
        w = torch.as_tensor(taps, dtype=torch.float32)
        self.weight = nn.Parameter(w.reshape(1, 1, *w.shape[-2:]), requires_grad=False)


def _gaussian_taps(size: int = 5, sigma: float = 1.0) -> torch.Tensor:
    # encoder.py:60-64 evaluates the Gaussian in float64 and normalises before casting to fp32
    c = size // 2
    k = torch.tensor([[(1.0 / (2 * math.pi * sigma ** 2)) * math.exp(-((i - c) ** 2 + (j - c) ** 2) / (2 * sigma ** 2))
                       for j in range(size)] for i in range(size)], dtype=torch.float64)
    return (k / k.sum()).float()


class ImageEncoder(nn.Module):
    def __init__(self, embedding_dim, channels):
        super().__init__()
        self.channels = channels
        self.sobel_x = _FixedFilter([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]])
        self.sobel_y = _FixedFilter([[-1, -2, -1], [0, 0, 0], [1, 2, 1]])
        self.gaussian_blur = _FixedFilter(_gaussian_taps(5, 1.0))
        self.laplacian = _FixedFilter([[1, 2, 1], [2, -12, 2], [1, 2, 1]])
        self.embed = nn.Sequential(
            nn.Conv2d(channels + 3, embedding_dim, kernel_size=3, padding=1),
            nn.ReLU(),
            nn.Conv2d(embedding_dim, embedding_dim, kernel_size=3, padding=1, bias=False),
        )

    def forward(self, x):
        if x.is_cuda and x.dtype == torch.float32 and self.channels <= 8 and not (torch.is_grad_enabled() and x.requires_grad):
            # the fixed-filter front as ONE HIP pass (ncahip_image_encoder_front_f32); the learned convolutions follow on MIOpen
            from . import ops
            k3 = torch.cat((self.sobel_x.weight, self.sobel_y.weight, self.laplacian.weight), dim=0)
            return self.embed(ops.image_encoder_front(x, k3, self.gaussian_blur.weight))
        gray = x.mean(dim=1, keepdim=True)
        edge_bank = torch.cat((self.sobel_x.weight, self.sobel_y.weight, self.laplacian.weight), dim=0)
        edges = F.conv2d(gray, edge_bank, padding=1)                                   # [B,3,H,W]
        blur = F.conv2d(x, self.gaussian_blur.weight.expand(self.channels, 1, 5, 5), padding=2,
                        groups=self.channels)                                          # [B,ch,H,W]
        return self.embed(torch.cat((edges, blur), dim=1))
