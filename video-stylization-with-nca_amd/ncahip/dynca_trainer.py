"""DyNCA training loop with a dense device-resident pool (reference: ConditioneDyNCA/experiments.py:158-159,
194-304; twins in ExtraChannels/experiments.py and fit_*_motion.py).  The reference keeps this loop inline in its
scripts; this class carries exactly the part SURVEY.md section 8 row a15 puts on the path:

    pool  = model.seed(pool_size, size)                       one [pool,C,H,W] tensor on the device (:158-159)
    every iteration i:
        reseed numpy / torch RNGs with i + 424                (:196-198)
        batch_idx = np.random.choice(pool, B, replace=False)  (:210)
        states    = pool[batch_idx]; every `inject_seed_step` iterations slot 0 <- model.seed(1)  (:213-216)
        T ~ np.random.randint(min_steps, max_steps)           (:224)
        states, rgb = model.forward_nsteps(states, T, cond_img=...)   (:226)   <- fused HIP steps, one autograd node
        loss(input_dict).backward(); p.grad /= ||p.grad|| + 1e-8 per parameter (:254-263); Adam; MultiStepLR(gamma 0.5)
        pool[batch_idx] = states                              (:269 writes back states[:, :12]; generalised to c_in)

The loss is supplied by the caller (the reference's appearance / motion losses need VGG / MSOE weights that are not
obtainable offline).  With torch.distributed initialised each rank owns pool_size/world slots, takes batch_size/world of
them per iteration under a rank-offset reseed ((i+424)*world + rank), adopts rank 0's T, and the gradients are all-reduced
through one flat bucket before the normalisation (ncahip.dist); `batch_size` is the GLOBAL batch.
"""
from typing import Callable, Optional, Sequence

import numpy as np
import torch

from . import dist as ncadist


class DyNCATrainer:
    def __init__(self, model, loss_fn: Callable, pool_size: int = 256, size=(128, 128), batch_size: int = 4,
                 nca_steps: Sequence[int] = (32, 128), lr: float = 1e-3, lr_decay_step: Sequence[int] = (1000, 2000),
                 inject_seed_step: int = 8, device: Optional[torch.device] = None, reseed_offset: int = 424):
        self.model, self.loss_fn = model, loss_fn
        self.device = device if device is not None else next(model.parameters()).device
        self.size, self.batch_size = size, batch_size
        self.min_steps, self.max_steps = nca_steps
        self.inject_seed_step, self.reseed_offset = inject_seed_step, reseed_offset
        self.pool_size = ncadist.shard_size(pool_size)
        ncadist.broadcast_parameters(model)    # data parallel: every replica starts from rank 0's weights (no-op in one process)
        with torch.no_grad():
            self.pool = model.seed(self.pool_size, size=size).to(self.device)
        self.optimizer = torch.optim.Adam(model.parameters(), lr=lr)
        self.lr_scheduler = torch.optim.lr_scheduler.MultiStepLR(self.optimizer, list(lr_decay_step), 0.5)
        self.iteration = 0

    def step(self, cond_img: Optional[torch.Tensor] = None, extra: Optional[dict] = None):
        """One training iteration; returns (loss value as a 0-d tensor, T)."""
        i = self.iteration
        # single process: the reference's reseed (experiments.py:196-198).  Data parallel: a rank-offset seed, so the shards
        # draw different slots and fire masks; T is rank 0's draw, broadcast; the seed injection into slot 0 is per GLOBAL batch
        seed = ncadist.rank_seed(i + self.reseed_offset)
        np.random.seed(seed % (2 ** 32))
        torch.manual_seed(seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(seed)
        if hasattr(self.model, "mask_seed") and ncadist.world_size() > 1:
            self.model.mask_seed = seed
        local = ncadist.local_batch(self.batch_size)
        with torch.no_grad():
            batch_idx = np.random.choice(self.pool_size, local, replace=False)
            idx = torch.as_tensor(batch_idx, device=self.pool.device)
            states = self.pool.index_select(0, idx)
            if i % self.inject_seed_step == 0 and ncadist.global_slots(1):
                states[:1] = self.model.seed(1, size=self.size).to(states.device)[:1]
        step_n = ncadist.shared_int(int(np.random.randint(self.min_steps, self.max_steps)))
        kw = {} if cond_img is None else {"cond_img": cond_img}
        states_after, rgb = self.model.forward_nsteps(states, step_n, **kw)
        input_dict = {"generated_image_list": [rgb], "nca_state": states_after, "step_n": step_n}
        if extra:
            input_dict.update(extra)
        loss = self.loss_fn(input_dict)
        loss = loss[0] if isinstance(loss, (tuple, list)) else loss
        self.optimizer.zero_grad()
        loss.backward()
        with torch.no_grad():
            params = [p for p in self.model.parameters() if p.requires_grad]
            ncadist.allreduce_mean_grads(params)
            for p in params:
                if p.grad is not None:
                    p.grad /= (p.grad.norm() + 1e-8)
            self.optimizer.step()
            self.lr_scheduler.step()
            self.pool.index_copy_(0, idx, states_after.detach()[:, : self.pool.shape[1]])
        self.iteration += 1
        return loss.detach(), step_n
