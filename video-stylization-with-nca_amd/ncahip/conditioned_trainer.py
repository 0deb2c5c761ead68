"""Pool-based training loop of the goal-conditioned NCA on the HIP hot path.

Public surface = the reference's `ConditionedNCATrainer` (EncoderConditioning/conditioned_trainer.py:27-181): constructor
arguments, `sample_batch / sample_targets / train_batch / update_pool / train / damage / emit_metrics`.  What an iteration
does is the reference's recipe, stated here once (line numbers are the reference's):

  idxs     = random.sample(range(pool), batch_size)                              (:160)
  targets  = dataset[np.random.choice(len(dataset), batch_size)]                 (:118-120)
  batch    = pool[idxs], empty or dead entries <- seed (:104-115), entries 0 and 1 <- fresh seeds (:167)
  twice:     T = random.randint(min_steps, max_steps); x = nca.grow(x, T, targets); loss; backward;
             every parameter's gradient /= its own L2 norm + 1e-10; Adam; MultiStepLR([5000], 0.3) step   (:122-151, :169-171)
  pool[idxs] = x                                                                  (:153-154)

How it is carried out differs: the pool is one device tensor (index_select / index_copy_ instead of a Python list and
`torch.stack`), dead samples are found by ONE alive-mask launch over the batch instead of a host round trip per sample,
the per-parameter gradient sums the reference logs are fetched with a single transfer, and with `torch.distributed`
initialised every rank owns pool_size/world slots and samples batch_size/world of them with a rank-offset generator, rank 0's
draw of T is broadcast so all ranks run the same number of steps, the 2 fresh seeds of an iteration are dealt over the ranks
(per GLOBAL batch), and the gradients travel in one flat all-reduce before they are normalised (ncahip.dist).  `loss=` accepts any module mapping the reference's loss-input dict to `(loss, summary)`.
"""
import math
import random
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import dist as ncadist
from .loss import Loss
from .sample_pool import SamplePool
from .trainer import NCATrainer

_GRAD_EPS = 1e-10


class PhaseTimer:
    """Optional per-phase device timing of train_batch (bench.py's `train` leg): `mark(name)` records an event on the current
    stream; the time between two consecutive marks is booked under the LATER mark's name.  Never on unless a caller sets
    `trainer.phase_timer = PhaseTimer()`; no host synchronisation until `summary()`."""

    def __init__(self):
        self.events = []

    def mark(self, name: str) -> None:
        if torch.cuda.is_available():
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.events.append((name, ev))

    def reset(self) -> None:
        self.events = []

    def summary(self) -> Dict[str, float]:
        """ms per phase, summed over everything recorded since the last reset ('start' marks open an interval)."""
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        out: Dict[str, float] = {}
        for (_, e0), (n1, e1) in zip(self.events, self.events[1:]):
            if n1 != "start":
                out[n1] = out.get(n1, 0.0) + e0.elapsed_time(e1)
        return out


class ConditionedNCATrainer(NCATrainer):
    def __init__(self, nca, target_dataset, target_style_image, nca_steps=[48, 96], lr: float = 2e-3,
                 pool_size: int = 512, num_damaged: int = 0, log_base_path: str = "test", damage_radius: int = 3,
                 appearance_loss_type: str = "OT", appearance_loss_weight: float = 1.0, content_loss_weight: float = 1.0,
                 overflow_loss_weight: float = 1.0, device: Optional[torch.device] = None, visualiser=None, loss=None,
                 sample_seed: int = 0, pool_dtype: torch.dtype = torch.float32):
        super().__init__(pool_size, num_damaged, log_base_path, device)
        self.nca, self.visualiser = nca, visualiser
        # data
        self.target_dataset = target_dataset
        self.target_size = target_dataset.target_size
        self.num_target_channels, self.image_size = self.target_size[0], self.target_size[-1]
        self.rgb = self.num_target_channels == 3
        # schedule
        self.min_steps, self.max_steps = nca_steps
        self.damage_radius = damage_radius
        self.log_every = 1
        ncadist.broadcast_parameters(nca)      # data parallel: every replica starts from rank 0's weights (no-op in one process)
        self.optimizer = torch.optim.Adam(nca.parameters(), lr=lr)
        self.lr_sched = torch.optim.lr_scheduler.MultiStepLR(self.optimizer, milestones=[5000], gamma=0.3)
        # objective
        if loss is None:
            loss = Loss(device=self.device, target_style_image=target_style_image, appearance_loss_type=appearance_loss_type,
                        appearance_loss_weight=appearance_loss_weight, content_loss_weight=content_loss_weight,
                        overflow_loss_weight=overflow_loss_weight)
        self.loss = loss
        self.phase_timer: Optional[PhaseTimer] = None
        # storage type of the pool and of the grow loop's history ring: float32 (the reference) or bfloat16 (BASELINE
        # configs[2]: half the saved-for-backward set; bf16-storage kernels forward, fp32 gradients)
        self.pool_dtype = pool_dtype
        # this rank's shard of the pool
        self.pool_size = ncadist.shard_size(pool_size)
        self.pool = SamplePool(self.pool_size)
        # sampling generators: the reference's global `random` / `np.random` streams in a single process; with data
        # parallelism rank-offset generators, so the shards put different samples into the global batch (T stays shared)
        self._py_rng, self._np_rng = random, np.random
        if ncadist.world_size() > 1:
            base = int(sample_seed)
            self._py_rng = random.Random(ncadist.rank_seed(base))
            self._np_rng = np.random.RandomState(ncadist.rank_seed(base) % (2 ** 32))
            torch.manual_seed(ncadist.rank_seed(base + 1))              # fire masks drawn with torch.rand_like (nca.py:172)
            if hasattr(nca, "mask_seed"):
                nca.mask_seed = ncadist.rank_seed(int(nca.mask_seed) + 1)   # ... or in-kernel Philox

    # ------------------------------------------------------------------------------------------------ sampling
    def sample_targets(self, sampled_indices):
        picks = self._np_rng.choice(len(self.target_dataset), size=len(sampled_indices), replace=True)
        return self.target_dataset[picks]

    def sample_batch(self, sampled_indices, sample_pool) -> torch.Tensor:
        fresh = self.nca.generate_seed(1)[0].to(self.device, self.pool_dtype)
        states = sample_pool.gather(sampled_indices, fresh).to(self.device)      # never-written slots come back as `fresh`
        any_alive = self.nca.alive(states).flatten(1).any(dim=1)
        return torch.where(any_alive.view(-1, 1, 1, 1), states, fresh.unsqueeze(0))

    def damage(self, batch):
        n, side = batch.size(0), self.image_size
        rows, cols = np.ogrid[:side, :side]
        for k in range(self.num_damaged):
            cy, cx = np.random.randint(0, side, 2)
            disc = (rows - cy) ** 2 + (cols - cx) ** 2 <= self.damage_radius ** 2
            batch[max(n - 1 - k, 0)][:, torch.from_numpy(disc).to(batch.device)] = 0.0
        return batch

    def update_pool(self, idxs, outputs, targets):
        self.pool[idxs] = outputs.detach()

    # ------------------------------------------------------------------------------------------------ one optimiser step
    def _normalise_and_step(self) -> List[torch.nn.Parameter]:
        live = [p for p in self.nca.parameters() if p.requires_grad]
        ncadist.allreduce_mean_grads(live)               # global-batch gradient first, then the per-tensor normalisation
        if self.phase_timer:
            self.phase_timer.mark("allreduce")
        for p in live:
            if p.grad is not None:
                p.grad.div_(p.grad.norm() + _GRAD_EPS)
        self.optimizer.step()
        self.lr_sched.step()
        if self.phase_timer:
            self.phase_timer.mark("normalise_adam")
        return live

    def _report(self, loss: torch.Tensor, parts: Optional[Dict]) -> Dict[str, float]:
        named = [(n, p.grad) for n, p in self.nca.named_parameters() if p.grad is not None]
        fetched = torch.stack([g.sum() for _, g in named] + [loss.detach()]).tolist()     # the step's only host sync
        if loss.is_cuda:
            from . import ops
            ops.check_errors()              # the stream is drained anyway: surface any recorded device-side failure now
        value = fetched.pop()
        report = {"loss": value}
        report.update({k: float(v) for k, v in (parts or {}).items()})
        report["log10loss"] = math.log10(value + 1e-5)
        report.update({f"{n}_grad": s for (n, _), s in zip(named, fetched)})
        return report

    def train_batch(self, batch, targets):
        steps = ncadist.shared_int(random.randint(self.min_steps, self.max_steps))   # every rank: rank 0's draw
        pt = self.phase_timer
        if pt:
            pt.mark("start")
        grown = self.nca.grow(batch, num_steps=steps, goal=targets)
        if pt:
            pt.mark("grow_fwd")
            if grown.requires_grad:       # fires when dL/d(grown) exists: the objective's forward + backward end here
                grown.register_hook(lambda g: (pt.mark("objective_fwd_bwd"), g)[1])
        gf = grown.float()                                    # a bf16 pool: the objective is evaluated in fp32
        loss, parts = self.loss({"generated_images": gf[:, :self.num_target_channels], "nca_state": gf,
                                 "target_images": targets})
        self.optimizer.zero_grad()
        loss.backward()
        if pt:
            pt.mark("grow_bwd")              # fused backward of the T steps + the conditioning encoder's backward
        self._normalise_and_step()
        report = self._report(loss, parts)
        if pt:
            pt.mark("report")
        return grown.detach(), report["loss"], report

    # ------------------------------------------------------------------------------------------------ loop
    def emit_metrics(self, i: int, batch, outputs, targets, loss, metrics={}):
        w = self.train_writer
        w.scalars(i, **{"loss": loss, "log10(loss)": math.log10(loss)})
        for tag, images in (("batch", batch), ("outputs", outputs), ("targets", targets)):
            w.add_images(tag, self.to_rgb(images), i, dataformats="NCHW")
        w.scalars(i, **metrics)

    def _iteration(self, i: int, batch_size: int):
        # `batch_size` is the GLOBAL batch: each rank takes its share from its own pool shard
        idxs: Sequence[int] = self._py_rng.sample(range(len(self.pool)), ncadist.local_batch(batch_size))
        with torch.no_grad():
            targets = self.sample_targets(idxs).to(self.device)
            batch = self.sample_batch(idxs, self.pool).to(self.device)
            fresh = ncadist.global_slots(2)               # 2 fresh seeds per global batch (conditioned_trainer.py:167)
            if fresh:
                batch[:fresh] = self.nca.generate_seed(fresh).to(self.device, batch.dtype)
        outputs = batch
        for _ in range(2):                                # the reference trains twice on every sampled batch
            outputs, loss, metrics = self.train_batch(outputs, targets)
        self.update_pool(idxs, outputs, targets)
        return batch, outputs, targets, loss, metrics

    def train(self, batch_size, epochs, *args, **kwargs):
        try:
            from tqdm import tqdm
            progress = tqdm(range(epochs))
        except Exception:
            progress = range(epochs)
        self.pool = SamplePool(self.pool_size)
        for i in progress:
            batch, outputs, targets, loss, metrics = self._iteration(i, batch_size)
            if hasattr(progress, "set_description"):
                progress.set_description(f"iteration {i}/{epochs}  loss {loss:.5f}")
            if i % self.log_every == 0:
                self.emit_metrics(i, batch, outputs, targets, loss, metrics)
            if self.visualiser is not None:
                self.visualiser.step(i, self.to_rgb(batch), self.to_rgb(outputs), self.to_rgb(targets), metrics)
