"""Data parallelism for the NCA trainers: one process per GPU, torch.distributed over RCCL (backend "nccl" on
ROCm) or gloo for CPU tests.  The reference has no distributed code at all; this is the
one strategy BASELINE.json's north_star asks for:

  * the sample pool is sharded -- every rank owns pool_size/world slots in its own HBM and samples only from
    them; states never cross xGMI;
  * per optimiser step ONE collective: all-reduce (mean) of a single flat fp32 bucket holding every parameter
    gradient (~10.7k floats = 43 KB: latency-bound, so one bucket / one call), issued BEFORE the per-parameter
    gradient normalisation (conditioned_trainer.py:134-136) so the normalised direction equals the
    single-process large-batch one;
  * the replicas START equal: `broadcast_parameters` sends rank 0's parameters and buffers to every rank (one flat bucket
    per dtype) when a trainer is constructed, before its optimiser exists -- each process builds its model under its own
    (rank-offset) RNG state, and Adam would keep differently initialised replicas apart silently;
  * every rank runs the same number of NCA steps: rank 0 draws T with the reference's own call and broadcasts it
    (`shared_int`), so no rank idles at the collective; everything else that is sampled (pool slots, targets, fire
    masks) comes from rank-offset generators (`rank_seed`) so the shards contribute DIFFERENT samples to the global batch;
  * the reference's per-iteration seed injections are per GLOBAL batch (`global_slots`): 2 fresh seeds per batch in
    ConditionedNCATrainer (conditioned_trainer.py:167), slot 0 every 8th iteration in the DyNCA loop (experiments.py:213-216).
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


def _coll_device() -> torch.device:
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


@torch.no_grad()
def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> int:
    """Make every rank's replica equal to rank `src`'s: all parameters and buffers of `module` travel in ONE flat bucket per
    dtype (a handful of KB for the NCA models: latency-bound, so one call) and are copied back in place.  Returns the number
    of elements sent.  A no-op in a single process."""
    if world_size() == 1:
        return 0
    tensors = [t for t in list(module.parameters()) + list(module.buffers()) if t.numel()]
    sent = 0
    for dtype in sorted({t.dtype for t in tensors}, key=str):
        group_t = [t for t in tensors if t.dtype == dtype]
        dev = _coll_device()
        flat = torch.cat([t.detach().reshape(-1).to(dev) for t in group_t])
        dist.broadcast(flat, src=src, group=group)
        off = 0
        for t in group_t:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t).to(t.device))
            off += n
        sent += flat.numel()
    return sent


def shared_int(value: int = 0) -> int:
    """Rank 0's `value` on every rank (one 8-byte broadcast): the per-iteration step count T, drawn by rank 0 with the
    reference's own RNG call, so that every rank runs the same number of NCA steps and none idles at the all-reduce."""
    if world_size() == 1:
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64).to(_coll_device())
    dist.broadcast(t, src=0)
    return int(t.item())


shared_randint_seed = shared_int      # earlier name


def rank_seed(base: int) -> int:
    """A seed that differs per rank and per `base` (base * world + rank): the sampling generators of the trainers."""
    return int(base) * world_size() + rank()


def local_batch(global_batch: int) -> int:
    """Per-rank share of a global batch (must divide evenly: the all-reduce averages equal shards)."""
    w = world_size()
    if global_batch % w != 0:
        raise ValueError(f"ncahip.dist: global batch {global_batch} is not divisible by the world size {w}")
    return global_batch // w


def global_slots(n: int) -> int:
    """How many of `n` per-global-batch items (fresh seeds) fall to this rank: item k goes to rank k % world."""
    w, r = world_size(), rank()
    return sum(1 for k in range(n) if k % w == r)
