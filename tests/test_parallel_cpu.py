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
    # 3) ragged and EMPTY shards: totals that do not divide by the world size, and fewer samples than ranks (the gather is a
    #    collective - every rank calls it, also with zero rows; the entry script's and bench.py's sharding code)
    for total in (7, 5, 1):
        zt = torch.randn(total, 12, 4, 4, generator=torch.Generator().manual_seed(41))
        lo, hi = parallel.shard_range(total, rank, world)
        got = parallel.all_gather_cat(zt[lo:hi] * 2.0)
        ok_g = ok_g and got.shape[0] == total and torch.equal(got, zt * 2.0)
    t = parallel.max_over_ranks(1.0 + rank)
    ok_g = ok_g and parallel.ranks_seen() == world and parallel.collective_info()["backend"] == "gloo"
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


def _worker_world1(port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    from ln3diff_amd import parallel
    r, lr, w = parallel.setup_dist('gloo')
    ok = (r, w) == (0, 1) and dist.is_initialized()                    # launcher environment: a process group even with one rank
    ts = [torch.arange(6.0).reshape(2, 3), torch.ones(4, dtype=torch.bfloat16)]
    ref = [t.clone() for t in ts]
    parallel.broadcast_flat(ts, src=0)
    ok = ok and all(torch.equal(a, b) for a, b in zip(ts, ref))
    z = torch.randn(3, 12, 4, 4)
    ok = ok and torch.equal(parallel.all_gather_cat(z), z) and parallel.all_gather_cat(z[:0]).shape[0] == 0
    ok = ok and parallel.ranks_seen() == 1 and parallel.max_over_ranks(2.5) == 2.5
    parallel.barrier()
    q.put(ok)


def test_world1_process_group_runs_the_same_collectives():
    """r4: a launcher-started single rank (driver: `torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`) creates a process
    group and goes through every collective of the N-rank path (gloo here, RCCL in tests/test_dist_gpu.py)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_worker_world1, args=(_free_port(), q))
    p.start()
    assert q.get(timeout=120) is True
    p.join(timeout=60)
    assert p.exitcode == 0


def _worker_dies(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import time
    from ln3diff_amd import parallel
    parallel.setup_dist('gloo', timeout_s=8)
    parallel.barrier()
    if rank == 1:
        os._exit(3)                                  # a rank that dies between collectives
    t0 = time.time()
    try:
        parallel.all_gather_cat(torch.zeros(2, 3))
        q.put(('no error', time.time() - t0))
    except Exception as e:
        q.put((type(e).__name__, time.time() - t0))


def test_dead_rank_surfaces_as_an_error_not_a_hang():
    """The process group carries a timeout (parallel.PG_TIMEOUT_S = 30 min for the entry points with their rank-0-only phases, 180 s in bench.py, 8 s here; the reference sets 15 h,
    guided_diffusion/dist_util.py:68): the survivor of a rank that died gets an exception from its next collective."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_dies, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    what, dt = q.get(timeout=120)
    for p in ps:
        p.join(timeout=60)
    assert what != 'no error' and dt < 60, (what, dt)
    assert ps[1].exitcode == 3


def _stub_sample_fn(z_all):
    """Stand-in for "denoise my samples": a deterministic per-sample function of the globally seeded noise."""
    def fn(lo, hi):
        return torch.tanh(z_all[lo:hi] * 1.7) + 0.25
    return fn


def _stub_render_fn(n_views):
    """Stand-in for decode + render of (sample, view) pairs: frame = f(latent of that sample, view index)."""
    def fn(latent_all, pairs):
        fr = [latent_all[s].mean(0, keepdim=True) * (1.0 + 0.125 * v) for s, v0, v1 in pairs for v in range(v0, v1)]
        idx = torch.tensor([(s, v) for s, v0, v1 in pairs for v in range(v0, v1)], dtype=torch.int64).reshape(-1, 2)
        img = torch.stack(fr) if fr else torch.empty(0, 1, *latent_all.shape[2:])
        return {'image_raw': img, 'pair_index': idx}
    return fn


def _worker_step(rank, world, port, q, n_samples, n_views):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from ln3diff_amd import parallel
    parallel.setup_dist('gloo')
    z_all = torch.randn(n_samples, 12, 4, 4, generator=torch.Generator().manual_seed(41))
    lat, frames, pairs = parallel.sharded_step(_stub_sample_fn(z_all), _stub_render_fn(n_views), n_samples, n_views, rank, world,
                                               gather_frames=True)
    n_mine = sum(v1 - v0 for _, v0, v1 in pairs)
    parallel.barrier()
    q.put((rank, lat.numpy(), frames['image_raw'].numpy(), frames['pair_index'].numpy(), n_mine))      # numpy: pickled by value


def _run_world(world, n_samples, n_views):
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_step, args=(r, world, port, q, n_samples, n_views)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res, key=lambda r: r[0])


def test_sharded_step_samples_and_views_over_2_4_8_ranks():
    """bench.py's step (parallel.sharded_step) on stub sample / render functions: 4 samples x 40 views on 8 ranks (fewer samples
    than ranks: the render is shared out by (sample, view) pairs), and ragged cases on 2 and 4 ranks.  Every pair is rendered
    exactly once, every rank has work, and latents and frames equal the single-process result bit for bit."""
    from ln3diff_amd import parallel
    for world, n_samples, n_views in ((8, 4, 40), (4, 6, 5), (2, 3, 7)):
        z_all = torch.randn(n_samples, 12, 4, 4, generator=torch.Generator().manual_seed(41))
        lat1, fr1, pairs1 = parallel.sharded_step(_stub_sample_fn(z_all), _stub_render_fn(n_views), n_samples, n_views, 0, 1)
        assert pairs1 == [(s, 0, n_views) for s in range(n_samples)]
        res = _run_world(world, n_samples, n_views)
        counts = [r[4] for r in res]
        assert sum(counts) == n_samples * n_views and min(counts) >= (n_samples * n_views) // world - 1 and min(counts) > 0, counts
        for rank, lat, img, idx, _ in res:
            assert torch.equal(torch.from_numpy(lat), lat1)                  # one all_gather: every rank holds all latents
            assert idx.tolist() == [[s, v] for s in range(n_samples) for v in range(n_views)]      # each pair once, in order
            assert torch.equal(torch.from_numpy(img), fr1['image_raw'])


def test_shard_pairs_cover_every_pair_once():
    from ln3diff_amd.parallel import shard_pairs
    for n_samples, n_views in ((1, 1), (4, 40), (8, 40), (3, 24), (16, 24)):
        for world in (1, 2, 3, 4, 8):
            got = [(s, v) for r in range(world) for s, v0, v1 in shard_pairs(n_samples, n_views, r, world) for v in range(v0, v1)]
            assert got == [(s, v) for s in range(n_samples) for v in range(n_views)]
            sizes = [sum(v1 - v0 for _, v0, v1 in shard_pairs(n_samples, n_views, r, world)) for r in range(world)]
            assert max(sizes) - min(s for s in sizes if s) <= (n_samples * n_views + world - 1) // world


def test_shard_range_partitions():
    from ln3diff_amd.parallel import shard_range
    for total in (1, 7, 8, 64):
        for world in (1, 2, 4, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            cover = [i for lo, hi in spans for i in range(lo, hi)]
            assert cover == list(range(total))


def test_checkpoint_loader_prefixes_and_strictness(tmp_path):
    """ln3diff_amd.checkpoint: reference-named tensors under the released prefixes load strictly; a shape mismatch raises."""
    import torch
    from safetensors.torch import save_file
    from ln3diff_amd.checkpoint import load_checkpoint
    from ln3diff_amd.dit.dit_trilatent import DiT_TriLatent
    from ln3diff_amd.synth import synth_state_dict
    m = DiT_TriLatent(input_size=32, patch_size=2, in_channels=4, hidden_size=128, depth=1, num_heads=2, context_dim=768,
                      num_classes=0, learn_sigma=False, roll_out=True)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert 'blocks.0.mlp.mlp.0.weight' in shapes and 'blocks.0.mlp.mlp.1.bias' in shapes      # xformers FusedMLP naming
    sd = synth_state_dict({k: v for k, v in shapes.items() if 'pos_embed' not in k}, 3)
    sd.update({k: v.clone() for k, v in m.state_dict().items() if 'pos_embed' in k})     # computed buffer, saved as is
    f = str(tmp_path / 'ck.safetensors')
    save_file({'ddpm_model.' + k: v.contiguous() for k, v in sd.items()}, f)
    rep = load_checkpoint(f, dit=m)
    assert rep['dit'] == {'ddpm_model.': len(shapes)}
    assert all(torch.equal(m.state_dict()[k], sd[k]) for k in sd)
    bad = dict(sd); bad['final_layer.linear.bias'] = torch.zeros(3)
    f2 = str(tmp_path / 'bad.pt')
    torch.save({'module.' + k: v for k, v in bad.items()}, f2)
    import pytest
    with pytest.raises(RuntimeError):
        load_checkpoint(f2, dit=m)


def test_weight_cache_epoch_bumps_on_every_weight_writer():
    """ln3diff_amd._cache: packed device copies of the weights are keyed on a global epoch that every in-place weight writer bumps
    (load_state_dict on a parent or a child, the checkpoint loader, the synthetic filler, invalidate_weight_caches)."""
    import ln3diff_amd
    from ln3diff_amd import _cache
    from ln3diff_amd.dit.dit_trilatent import DiT_TriLatent
    from ln3diff_amd.synth import fill_module_random_
    m = DiT_TriLatent(input_size=32, patch_size=2, in_channels=4, hidden_size=128, depth=1, num_heads=2, context_dim=768, num_classes=0,
                      learn_sigma=False, roll_out=True)
    e0 = _cache.EPOCH[0]
    m.load_state_dict(m.state_dict())
    e1 = _cache.EPOCH[0]
    assert e1 > e0
    _cache.stamp({}, m)                                             # what building the packed copies does (first forward)
    m.blocks[0].load_state_dict(m.blocks[0].state_dict())          # a CHILD's load must invalidate the parent's packed copies too
    e2 = _cache.EPOCH[0]
    assert e2 > e1
    fill_module_random_(m, 3)
    e3 = _cache.EPOCH[0]
    assert e3 > e2
    ln3diff_amd.invalidate_weight_caches()
    assert _cache.EPOCH[0] > e3
    stamp = _cache.stamp({'device': 'cpu'})
    assert _cache.fresh(stamp, 'cpu')
    _cache.bump()
    assert not _cache.fresh(stamp, 'cpu')
