"""ConditionedNCATrainer drop-in (reference: EncoderConditioning/conditioned_trainer.py:27-181).

Same constructor and method surface; the inner loop keeps the reference's semantics -- idxs from
`random.sample`, targets from `np.random.choice`, empty/dead pool slots reseeded, the first two batch
entries replaced by fresh seeds, TWO train_batch calls per iteration, T ~ random.randint(min,max) NCA steps,
per-parameter gradient L2 normalisation, Adam + MultiStepLR([5000], 0.3) stepped once per train_batch --
with the hot path on the HIP kernels and the host synchronisation points removed from it:
  * the batch is one index_select from the device-resident pool (no Python stack);
  * dead samples are found with one alive-mask kernel over the batch and replaced by torch.where
    (the reference syncs once per sample, conditioned_trainer.py:112);
  * per-parameter grad sums are fetched with one transfer instead of one .item() per parameter (:139-142);
  * with torch.distributed initialised (ncahip.dist), each rank trains on its own pool shard and the flat
    gradient bucket is all-reduced once per train_batch, before the normalisation.
"""
import math
import random
from typing import Any, Optional, Tuple  # noqa

import numpy as np
import torch

from . import dist as ncadist
from .loss import Loss
from .sample_pool import SamplePool
from .trainer import NCATrainer


class ConditionedNCATrainer(NCATrainer):
    def __init__(self, nca, target_dataset, target_style_image, nca_steps=[48, 96], lr: float = 2e-3,
                 pool_size: int = 512, num_damaged: int = 0, log_base_path: str = "test", damage_radius: int = 3,
                 appearance_loss_type: str = "OT", appearance_loss_weight: float = 1.0, content_loss_weight: float = 1.0,
                 overflow_loss_weight: float = 1.0, device: Optional[torch.device] = None, visualiser=None, loss=None):
        super().__init__(pool_size, num_damaged, log_base_path, device)
        self.target_dataset = target_dataset
        self.target_size = self.target_dataset.target_size
        self.nca = nca
        self.min_steps, self.max_steps = nca_steps[0], nca_steps[1]
        self.num_target_channels = self.target_size[0]
        self.image_size = self.target_size[-1]
        self.rgb = self.target_size[0] == 3
        self.damage_radius = damage_radius
        self.optimizer = torch.optim.Adam(self.nca.parameters(), lr=lr)
        self.lr_sched = torch.optim.lr_scheduler.MultiStepLR(self.optimizer, [5000], gamma=0.3)
        self.visualiser = visualiser
        # `loss=` (extension): any module mapping the reference's input dict to (loss, summary)
        self.loss = loss if loss is not None else Loss(
            device=self.device, content_loss_weight=content_loss_weight, overflow_loss_weight=overflow_loss_weight,
            appearance_loss_weight=appearance_loss_weight, appearance_loss_type=appearance_loss_type,
            target_style_image=target_style_image)
        self.pool_size = ncadist.shard_size(pool_size)   # this rank's shard of the global pool
        self.pool = SamplePool(self.pool_size)
        self.log_every = 1

    def emit_metrics(self, i: int, batch, outputs, targets, loss, metrics={}):
        with torch.no_grad():
            self.train_writer.add_scalar("loss", loss, i)
            self.train_writer.add_scalar("log10(loss)", math.log10(loss), i)
            self.train_writer.add_images("batch", self.to_rgb(batch), i, dataformats="NCHW")
            self.train_writer.add_images("outputs", self.to_rgb(outputs), i, dataformats="NCHW")
            self.train_writer.add_images("targets", self.to_rgb(targets), i, dataformats="NCHW")
            for k in metrics:
                self.train_writer.add_scalar(k, metrics[k], i)

    def damage(self, batch):
        size = batch.size(0)
        s = self.image_size
        yy, xx = np.ogrid[:s, :s]
        for i in range(self.num_damaged):
            cy, cx = np.random.randint(0, s, 2)
            mask = torch.from_numpy((yy - cy) ** 2 + (xx - cx) ** 2 <= self.damage_radius ** 2).to(batch.device)
            batch[max(size - i - 1, 0)][:, mask] *= 0.0
        return batch

    def sample_batch(self, sampled_indices, sample_pool) -> torch.Tensor:
        seed = self.nca.generate_seed(1)[0].to(self.device)
        batch = sample_pool.gather(sampled_indices, seed).to(self.device)
        dead = ~self.nca.alive(batch).flatten(1).any(dim=1)           # one kernel, no per-sample sync
        return torch.where(dead[:, None, None, None], seed[None], batch)

    def sample_targets(self, sampled_indices):
        random_indices = np.random.choice(len(self.target_dataset), len(sampled_indices), replace=True)
        return self.target_dataset[random_indices]

    def train_batch(self, batch, targets):
        num_steps = random.randint(self.min_steps, self.max_steps)
        batch = self.nca.grow(batch, num_steps=num_steps, goal=targets)
        loss_input_dict = {"target_images": targets, "nca_state": batch,
                           "generated_images": batch[:, : self.num_target_channels, :, :]}
        loss, loss_summary = self.loss(loss_input_dict)
        self.optimizer.zero_grad()
        loss.backward()
        params = [p for p in self.nca.parameters() if p.requires_grad]
        ncadist.allreduce_mean_grads(params)                         # one flat bucket, before the normalisation
        for p in params:
            if p.grad is not None:
                p.grad /= torch.norm(p.grad) + 1e-10
        self.optimizer.step()
        self.lr_sched.step()
        names = [n for n, W in self.nca.named_parameters() if W.grad is not None]
        sums = torch.stack([W.grad.sum() for n, W in self.nca.named_parameters() if W.grad is not None] + [loss.detach()])
        vals = sums.tolist()                                          # the only host sync of the step
        loss_v = vals[-1]
        grad_dict = {"{}_grad".format(n): v for n, v in zip(names, vals[:-1])}
        summary = {k: (float(v) if not isinstance(v, float) else v) for k, v in (loss_summary or {}).items()}
        return (batch.detach(), loss_v,
                {"loss": loss_v, **summary, "log10loss": math.log10(loss_v + 1e-5), **grad_dict})

    def update_pool(self, idxs, outputs, targets):
        self.pool[idxs] = outputs.detach()

    def train(self, batch_size, epochs, *args, **kwargs):
        try:
            import tqdm
            bar = tqdm.tqdm(range(epochs))
        except Exception:
            bar = range(epochs)
        self.pool = SamplePool(self.pool_size)
        for i in bar:
            idxs = random.sample(range(len(self.pool)), batch_size)
            with torch.no_grad():
                targets = self.sample_targets(idxs).to(self.device)
                batch = self.sample_batch(idxs, self.pool).to(self.device)
                batch[:2] = self.nca.generate_seed(2).to(self.device)
            outputs, loss, metrics = self.train_batch(batch, targets)
            outputs, loss, metrics = self.train_batch(outputs, targets)  # train more
            self.update_pool(idxs, outputs, targets)
            if hasattr(bar, "set_description"):
                bar.set_description(f"Epoch {i}/{epochs}: loss:{loss:.5f}")
            if i % self.log_every == 0:
                self.emit_metrics(i, batch, outputs, targets, loss, metrics)
            if self.visualiser is not None:
                self.visualiser.step(i, self.to_rgb(batch), self.to_rgb(outputs), self.to_rgb(targets), metrics)
