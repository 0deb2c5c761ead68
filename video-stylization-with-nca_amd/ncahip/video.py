"""Frame-conditioned video synthesis loop.

Reference: ConditioneDyNCA/utils/misc/video_utils.py:50-83 (`save_video`): seed once, then for every target frame run
`steps_per_frame` x `forward_nsteps(h, step_n, cond_img=gray(frame))` and emit `clip(rgb, -1, 1) * 0.5 + 0.5`.  Video
decoding / encoding (moviepy, cv2) is outside the hot path: frames come in and go out as tensors.
"""
from typing import Iterable, Iterator, Optional

import torch


def rgb_to_grayscale(x: torch.Tensor) -> torch.Tensor:
    """ITU-R 601 luma as torchvision's rgb_to_grayscale (the reference's RGBToGrayscale): [B,3,H,W] -> [B,1,H,W]."""
    r, g, b = x.unbind(dim=-3)
    return (0.2989 * r + 0.587 * g + 0.114 * b).unsqueeze(-3)


@torch.no_grad()
def synthesize_video(nca_model, frames: Iterable[torch.Tensor], step_n: int = 8, steps_per_frame: int = 1,
                     size=None, state: Optional[torch.Tensor] = None) -> Iterator[torch.Tensor]:
    """Yields one [3,H,W] image in [0,1] per (frame, k): video_utils.py:66-83.  `frames`: tensors [3,H,W] in [-1,1] on
    the model's device (the reference's preprocess_video output, one time slice each).  The NCA state persists across
    frames; pass `state` to continue a previous call."""
    h = state
    for frame in frames:
        frame = frame.to(nca_model.device)
        if h is None:
            hh, ww = frame.shape[-2:]
            h = nca_model.seed(1, size=size if size is not None else (ww, hh))
        cond = rgb_to_grayscale(frame.unsqueeze(0))
        for _ in range(int(steps_per_frame)):
            h, rgb = nca_model.forward_nsteps(h, step_n, cond_img=cond)
            yield (rgb[0].clamp(-1.0, 1.0) + 1.0) / 2.0
    synthesize_video.last_state = h
