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


# Process-group timeout.  The entry points have rank-0-only phases (conditioner checkpoint load + encoding, then broadcast_object)
# and unbalanced tails (a rank with a smaller shard waits in all_gather / agree_ok for the others' sampling, dopri5 or mesh
# export), so the default is generous: 30 min, LN3D_PG_TIMEOUT_S overrides it (the reference waits 15 h:
# guided_diffusion/dist_util.py:68).  bench.py and the tests, which have no such phase, pass their own short timeout so that a
# rank that died surfaces within minutes.
PG_TIMEOUT_S = int(os.environ.get("LN3D_PG_TIMEOUT_S", "1800"))
BENCH_PG_TIMEOUT_S = 180


def launched_by_torchrun():
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ and "MASTER_ADDR" in os.environ


def setup_dist(backend=None, timeout_s=None):
    """env:// rendezvous (reference guided_diffusion/dist_util.py:57-73).  The process group is created whenever the process was
    started by a launcher (RANK / WORLD_SIZE / MASTER_ADDR set) - also with ONE rank, so that a single-GPU box exercises the same
    RCCL initialisation, broadcast, all_gather and all_reduce calls as an 8-GPU node.  timeout_s: see PG_TIMEOUT_S."""
    import datetime
    rank, local_rank, world = env_rank()
    if (world > 1 or launched_by_torchrun()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, init_method="env://",
                                timeout=datetime.timedelta(seconds=timeout_s or PG_TIMEOUT_S))
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    return rank, local_rank, world


def ranks_seen(dev=None):
    """all_reduce of ones: how many ranks the collective layer actually reaches (1 without a process group)."""
    if not dist.is_initialized():
        return 1
    if dev is None:
        dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.ones(1, dtype=torch.float32, device=dev)
    dist.all_reduce(t)
    return int(round(float(t.item())))


def collective_info():
    """What carries the collectives: backend name and, for nccl (= RCCL on ROCm), the library version."""
    if not dist.is_initialized():
        return {"backend": None}
    info = {"backend": dist.get_backend()}
    if info["backend"] == "nccl":
        try:
            v = torch.cuda.nccl.version()
            info["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
        except Exception as e:                     # the version query is informational only
            info["rccl_version"] = "unknown (%s)" % type(e).__name__
    return info


def broadcast_flat(tensors, src=0):
    """Broadcast a list of same-device tensors as ONE flat buffer per dtype (also with one rank when a process group exists:
    the single-GPU launcher path runs the same RCCL calls as a node)."""
    if not dist.is_initialized():
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


def shard_pairs(n_samples, n_views, rank, world):
    """This rank's share of the n_samples x n_views independent (sample, view) render units, as [(sample, view_lo, view_hi)]:
    a consecutive range of the sample-major pair list, so with at least as many samples as ranks a rank renders the views of
    (mostly) its own samples, and with FEWER samples than ranks (--num_samples 4 on 8 GPUs) every rank still renders
    n_samples * n_views / world views instead of idling through the render (nsr/train_util_diffusion.py:262-283 renders one
    camera per call: any split is exact)."""
    lo, hi = shard_range(n_samples * n_views, rank, world)
    out, p = [], lo
    while p < hi:
        s, v0 = divmod(p, n_views)
        v1 = min(n_views, v0 + (hi - p))
        out.append((s, v0, v1))
        p += v1 - v0
    return out


def sharded_step(sample_fn, render_fn, n_samples, n_views, rank, world, gather_frames=False):
    """One pass of the hot path over a global batch on `world` ranks (bench.py's step and the entry points' body):
      1. sample_fn(lo, hi) -> latents [hi - lo, ...] of this rank's consecutive share of the samples (0 rows when it owns none);
      2. ONE all_gather of the latents (49 KB per sample) - the only collective of the step;
      3. render_fn(latent_all, pairs) -> {name: [P, ...]} for this rank's (sample, view) pairs (shard_pairs);
      4. optionally the frames of all ranks gathered in pair order ([n_samples * n_views, ...]).
    Every rank calls it (steps 2 and 4 are collectives), also a rank that owns no sample or no pair."""
    lo, hi = shard_range(n_samples, rank, world)
    latent_all = all_gather_cat(sample_fn(lo, hi))
    pairs = shard_pairs(n_samples, n_views, rank, world)
    frames = render_fn(latent_all, pairs)
    if gather_frames:
        frames = {k: all_gather_cat(v) for k, v in frames.items() if torch.is_tensor(v)}
    return latent_all, frames, pairs


def all_gather_cat(t):
    """Concatenate every rank's rows (dim 0).  Shards may be ragged or EMPTY (fewer samples than ranks): the row counts are
    gathered first, every rank pads to the largest shard, and the padding is dropped after the collective.  All ranks must
    call it (it is a collective) - also the ones whose shard is empty."""
    if not dist.is_initialized():
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
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(x):
    if not dist.is_initialized():
        return x
    dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def broadcast_object(obj, src=0):
    """A picklable object from `src` to every rank (conditioning tensors, error strings); identity without a process group."""
    if not dist.is_initialized():
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def agree_ok(err):
    """Collective: `err` is None or this rank's error message.  Raises RuntimeError on EVERY rank when any rank reported one - a
    failure on one rank must not leave the others waiting in their next collective."""
    if not dist.is_initialized():
        if err is not None:
            raise RuntimeError(err)
        return
    errs = [None] * dist.get_world_size()
    dist.all_gather_object(errs, err)
    bad = [f"rank {r}: {e}" for r, e in enumerate(errs) if e is not None]
    if bad:
        raise RuntimeError("; ".join(bad))


def shutdown():
    """Tear the process group down (quiet exit of launcher-started runs)."""
    if dist.is_initialized():
        dist.destroy_process_group()
