"""world_size-2 gloo tests (CPU) of the multi-GPU plumbing: flat weight broadcast, sharding that is independent of
the world size, latent all_gather, max-over-ranks timing."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from ln3diff_amd import parallel
    r, lr, w = parallel.setup_dist('gloo')
    assert (r, w) == (rank, world)
    # 1) one flat broadcast per dtype reproduces rank 0's tensors exactly on every rank
    g = torch.Generator().manual_seed(rank)           # different contents per rank before the broadcast
    ts = [torch.randn(7, 5, generator=g), torch.randn(3, generator=g).to(torch.bfloat16),
          torch.randn(2, 2, 2, generator=g), torch.randn(11, generator=g).to(torch.bfloat16)]
    parallel.broadcast_flat(ts, src=0)
    g0 = torch.Generator().manual_seed(0)
    ref = [torch.randn(7, 5, generator=g0), torch.randn(3, generator=g0).to(torch.bfloat16),
           torch.randn(2, 2, 2, generator=g0), torch.randn(11, generator=g0).to(torch.bfloat16)]
    ok_b = all(torch.equal(a, b) for a, b in zip(ts, ref))
    # 2) global-seed noise, sliced per rank, gathered back == the single-process batch
    gz = torch.Generator().manual_seed(41)
    z_all = torch.randn(6, 12, 4, 4, generator=gz)
    lo, hi = parallel.shard_range(6, rank, world)
    local = z_all[lo:hi] * 2.0                        # stand-in for "sample my shard"
    gathered = parallel.all_gather_cat(local)
    ok_g = torch.equal(gathered, z_all * 2.0)
    t = parallel.max_over_ranks(1.0 + rank)
    parallel.barrier()
    q.put((rank, ok_b, ok_g, t))


def test_world2_gloo_broadcast_shard_gather():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_b, ok_g, t in res:
        assert ok_b and ok_g and t == 2.0, (rank, ok_b, ok_g, t)


def test_shard_range_partitions():
    from ln3diff_amd.parallel import shard_range
    for total in (1, 7, 8, 64):
        for world in (1, 2, 4, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            cover = [i for lo, hi in spans for i in range(lo, hi)]
            assert cover == list(range(total))
