"""Base class of the pool-based NCA trainers.

Keeps the public surface of the reference's `NCATrainer` (EncoderConditioning/trainer.py:11-88: constructor arguments,
`pool`, `train_writer`, `sort_loss`, the rgb helpers and the overridable hooks) so that subclasses written against the
reference keep working, but is organised around two small collaborators of its own: a `_Telemetry` sink (tensorboard when it
is importable, a no-op otherwise) and the dense device-resident `SamplePool` of this package.
"""
import math
import os
from typing import Optional

import torch

from .sample_pool import SamplePool


class _Telemetry:
    """Scalar / image sink.  Falls back to a no-op when tensorboard is not installed (it is not in the offline image)."""

    def __init__(self, directory: str):
        self._sink = None
        try:
            from torch.utils.tensorboard import SummaryWriter
            self._sink = SummaryWriter(directory, flush_secs=10)
        except Exception:
            self._sink = None

    def add_scalar(self, tag, value, step):
        if self._sink is not None:
            self._sink.add_scalar(tag, value, step)

    def add_images(self, tag, images, step, dataformats="NCHW"):
        if self._sink is not None:
            self._sink.add_images(tag, images, step, dataformats=dataformats)

    def scalars(self, step, **values):
        for tag, value in values.items():
            self.add_scalar(tag, value, step)


def _unit_clamp(t: torch.Tensor) -> torch.Tensor:
    return t.clamp(0.0, 1.0)


def _hook(name: str):
    """An overridable hook that a concrete trainer has to provide (same exception type as the reference's stubs)."""

    def missing(self, *args, **kwargs):
        raise NotImplementedError(f"{type(self).__name__} does not implement `{name}`")

    missing.__name__ = name
    return missing


class NCATrainer:
    # hooks of the training loop; the conditioned trainer fills them in
    sample_batch = _hook("sample_batch")
    sample_targets = _hook("sample_targets")
    loss = _hook("loss")
    train_batch = _hook("train_batch")
    visualize = _hook("visualize")

    def __init__(self, pool_size: int = 256, num_damaged: int = 0, log_base_path: str = "test",
                 device: Optional[torch.device] = None):
        self.device = torch.device("cpu") if device is None else device
        self.num_damaged = num_damaged
        self.pool_size = pool_size
        self.pool = SamplePool(pool_size)
        self.log_base_path = log_base_path
        self.log_path = os.path.join(log_base_path, "tensorboard")
        self.train_writer = _Telemetry(self.log_path)
        # the reference binds this before subclasses overwrite `self.loss` with a module (trainer.py:28)
        self.sort_loss = self.loss

    # ---- image helpers (channel 3 is alpha; `self.rgb` is set by the subclass from the target's channel count) ----------
    def to_alpha(self, x):
        return _unit_clamp(x[:, 3:4])

    def to_rgb(self, x):
        colour = x[:, :3]
        if not self.rgb:                                   # RGBA target: composite over white
            colour = 1.0 - self.to_alpha(x) + colour
        return _unit_clamp(colour.float()).detach().cpu().numpy()      # (a bf16 pool: numpy has no bfloat16)

    # ---- default behaviour of the remaining hooks ------------------------------------------------------------------------
    def damage(self, batch):
        return batch                                       # no damage unless a subclass says so

    def update_pool(self, idxs, outputs, targets):
        self.pool[idxs] = outputs

    def emit_metrics(self, i: int, batch, outputs, loss, metrics={}):
        self.train_writer.scalars(i, **{"loss": loss, "log10(loss)": math.log10(loss)})
