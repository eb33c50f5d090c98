"""Multi-GPU plumbing: one process per GPU (torchrun), torch.distributed backend "nccl" (= RCCL over xGMI on ROCm).

The sampling path shards embarrassingly: every sample / (sample, view) is independent, there is NO collective
inside the denoise loop.  Collectives used:
  * start-up : rank 0's weights -> ONE flat buffer per dtype -> one `broadcast` each (the reference does one
    broadcast per parameter tensor: guided_diffusion/dist_util.py:122-133);
  * end      : `all_gather` of the final latents [B_local,12,32,32] f32 (49 KB / sample).
xGMI is point-to-point (7 links x ~153 GB/s per GPU); a 1.1 GB bf16 weight broadcast is start-up noise (~10 ms),
so the simple one-buffer broadcast is used rather than scatter+allgather.
"""
import os

import torch
import torch.distributed as dist

from . import _cache


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def setup_dist(backend=None):
    """env:// rendezvous (reference guided_diffusion/dist_util.py:57-73)."""
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, init_method="env://")
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    return rank, local_rank, world


def broadcast_flat(tensors, src=0):
    """Broadcast a list of same-device tensors as ONE flat buffer per dtype."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    for dtype, ts in by_dtype.items():
        flat = torch.cat([t.reshape(-1) for t in ts])
        dist.broadcast(flat, src)
        off = 0
        for t in ts:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n
    _cache.bump()          # in-place parameter writes: packed device copies of the weights are stale now


def shard_range(total, rank, world):
    """Consecutive shard of `total` independent units for `rank` (results independent of world size because
    noise is drawn for the full batch with the global seed and then sliced)."""
    per = (total + world - 1) // world
    lo = min(rank * per, total)
    return lo, min(lo + per, total)


def all_gather_cat(t):
    """Concatenate every rank's rows (dim 0).  Shards may be ragged or EMPTY (fewer samples than ranks): the row counts are
    gathered first, every rank pads to the largest shard, and the padding is dropped after the collective.  All ranks must
    call it (it is a collective) - also the ones whose shard is empty."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return t
    world = dist.get_world_size()
    n_loc = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    counts = [torch.zeros_like(n_loc) for _ in range(world)]
    dist.all_gather(counts, n_loc)
    counts = [int(c.item()) for c in counts]
    n_max = max(counts)
    if n_max == 0:
        return t
    pad = torch.zeros((n_max,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    return torch.cat([o[:c] for o, c in zip(outs, counts)], 0)


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(x):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return x
    dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
