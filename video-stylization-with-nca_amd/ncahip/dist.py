"""Data parallelism for the NCA trainers: one process per GPU, torch.distributed over RCCL (backend "nccl" on
ROCm) or gloo for CPU tests.  The reference has no distributed code at all; this is the
one strategy BASELINE.json's north_star asks for:

  * the sample pool is sharded -- every rank owns pool_size/world slots in its own HBM and samples only from
    them; states never cross xGMI;
  * per optimiser step ONE collective: all-reduce (mean) of a single flat fp32 bucket holding every parameter
    gradient (~10.7k floats = 43 KB: latency-bound, so one bucket / one call), issued BEFORE the per-parameter
    gradient normalisation (conditioned_trainer.py:134-136) so the normalised direction equals the
    single-process large-batch one;
  * every rank draws the same number of NCA steps (shared `random` seed) so no rank idles at the collective.
"""
import os
from typing import Iterable, Optional

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> tuple:
    """Initialise from the torchrun environment (RANK/WORLD_SIZE/LOCAL_RANK/MASTER_*).  Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 0, 1
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            local = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, **kw)
    return dist.get_rank(), dist.get_world_size()


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_size(pool_size: int) -> int:
    """Slots of the global pool owned by this rank (contiguous shards, remainder to the low ranks)."""
    w, r = world_size(), rank()
    return pool_size // w + (1 if r < pool_size % w else 0)


def allreduce_mean_grads(params: Iterable[torch.nn.Parameter], group=None) -> int:
    """All-reduce (mean) every existing .grad through ONE flat bucket.  Returns the bucket size in floats.
    Parameters without a gradient on this rank contribute zeros (so every rank issues the same collective)."""
    params = [p for p in params if p.requires_grad]
    w = world_size()
    if w == 1 or not params:
        return 0
    grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in params]
    flat = torch.cat([g.reshape(-1).float() for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(w)
    off = 0
    for p, g in zip(params, grads):
        n = g.numel()
        if p.grad is None:
            p.grad = torch.empty_like(p)
        p.grad.copy_(flat[off:off + n].view_as(p))
        off += n
    return flat.numel()


def shared_randint_seed(base: int = 0) -> int:
    """A seed every rank agrees on (rank 0's choice), for the per-iteration step-count draw."""
    t = torch.tensor([base], dtype=torch.int64)
    if world_size() > 1:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        t = t.to(dev)
        dist.broadcast(t, src=0)
    return int(t.item())
