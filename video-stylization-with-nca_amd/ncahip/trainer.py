"""NCATrainer drop-in (reference: EncoderConditioning/trainer.py:11-88): pool, logging, rgb helpers."""
import math
import os
from typing import Any, Optional, Tuple  # noqa

import torch

from .sample_pool import SamplePool


class _NullWriter:
    """Stands in for tensorboard's SummaryWriter when tensorboard is not installed."""

    def add_scalar(self, *a, **k):
        pass

    add_images = add_scalar


def _make_writer(path):
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(path, flush_secs=10)
    except Exception:
        return _NullWriter()


class NCATrainer:
    def __init__(self, pool_size: int = 256, num_damaged: int = 0, log_base_path: str = "test",
                 device: Optional[torch.device] = None):
        self.pool_size = pool_size
        self.pool = SamplePool(self.pool_size)
        self.num_damaged = num_damaged
        self.log_base_path = log_base_path
        self.log_path = os.path.join(log_base_path, "tensorboard")
        self.train_writer = _make_writer(self.log_path)
        self.sort_loss = self.loss  # bound before a subclass replaces self.loss (trainer.py:28)
        self.device = device if device is not None else torch.device("cpu")

    def to_alpha(self, x):
        return torch.clamp(x[:, 3:4, :, :], 0.0, 1.0)

    def to_rgb(self, x):
        if self.rgb:
            return torch.clamp(x[:, :3], 0.0, 1.0).detach().cpu().numpy()
        im = torch.clamp(1.0 - self.to_alpha(x) + x[:, :3, :, :], 0, 1)
        return im.detach().cpu().numpy()

    def sample_batch(self, sampled_indices, sample_pool) -> Tuple[Any, Any]:
        raise NotImplementedError("Sampled batch is not implemented!")

    def sample_targets(self, sampled_indices):
        raise NotImplementedError("Sampled targets not implemented!")

    def damage(self, batch):
        return batch

    def emit_metrics(self, i: int, batch, outputs, loss, metrics={}):
        with torch.no_grad():
            self.train_writer.add_scalar("loss", loss, i)
            self.train_writer.add_scalar("log10(loss)", math.log10(loss), i)

    def loss(self, batch, targets):
        raise NotImplementedError("loss not implemented!")

    def train_batch(self, batch, targets) -> Tuple[Any, Any]:
        raise NotImplementedError("train_batch not implemented!")

    def update_pool(self, idxs, outputs, targets):
        self.pool[idxs] = outputs

    def visualize(self, *args, **kwargs):
        raise NotImplementedError("Visualize is not implemented!")
