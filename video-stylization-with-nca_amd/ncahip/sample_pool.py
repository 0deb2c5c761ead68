"""SamplePool drop-in (reference: EncoderConditioning/sample_pool.py:14-33).

Same list-like API (`pool[i]`, `pool[idxs]` -> list, `pool[idxs] = batch`, `len(pool)`, entries start as
None), but the samples live in ONE device-resident [pool_size, C, H, W] tensor, allocated on first write,
so a batch is gathered / written back with a single index kernel (`gather` / `scatter`) instead of a Python
loop + torch.stack per iteration (conditioned_trainer.py:108-114).  With data parallelism each rank owns an
independent pool shard (ncahip.dist); samples never cross GPUs.
"""
from typing import Callable, Iterable, Optional

import torch
from torch.utils.data import Dataset


class SamplePool(Dataset):
    def __init__(self, pool_size: int = 256):
        self.pool_size = pool_size
        self._dense: Optional[torch.Tensor] = None
        self._valid = torch.zeros(pool_size, dtype=torch.bool)  # host-side: which slots hold a sample

    # ------------------------------------------------------------------ reference API
    def __len__(self):
        return self.pool_size

    def __getitem__(self, idx):
        if isinstance(idx, int):
            return self._dense[idx] if self._valid[idx] else None
        return [self[int(i)] for i in idx]

    def __setitem__(self, idx, value):
        if isinstance(idx, int):
            if value is None:
                self._valid[idx] = False
            else:
                self._ensure(value)
                self._dense[idx].copy_(value)
                self._valid[idx] = True
            return
        idx = [int(i) for i in idx]
        if isinstance(value, torch.Tensor):
            self.scatter(idx, value)
        else:
            for k, i in enumerate(idx):
                self[i] = value[k]

    # ------------------------------------------------------------------ batched, sync-free path
    @property
    def pool(self):
        """The reference exposes a python list; this view is built on demand for code that pokes at it."""
        return [self[i] for i in range(self.pool_size)]

    def _ensure(self, like: torch.Tensor):
        if self._dense is None:
            self._dense = torch.zeros((self.pool_size,) + tuple(like.shape[-3:]), dtype=like.dtype, device=like.device)

    def gather(self, idxs: Iterable[int], seed: torch.Tensor) -> torch.Tensor:
        """[B,C,H,W] batch; slots that are still None are filled with `seed` ([C,H,W]).  One index_select."""
        idxs = [int(i) for i in idxs]
        self._ensure(seed)
        sel = torch.as_tensor(idxs, device=self._dense.device)
        batch = self._dense.index_select(0, sel)
        empty = ~self._valid[idxs]
        if bool(empty.any()):
            batch[empty.to(batch.device)] = seed.to(batch.device, batch.dtype)
        return batch

    def scatter(self, idxs: Iterable[int], batch: torch.Tensor) -> None:
        idxs = [int(i) for i in idxs]
        self._ensure(batch)
        self._dense.index_copy_(0, torch.as_tensor(idxs, device=self._dense.device), batch.detach().to(self._dense.device, self._dense.dtype))
        self._valid[idxs] = True
