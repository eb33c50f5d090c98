"""GPU parity of the HIP DiT_TriLatent forward against (a) the committed golden outputs produced by the
REFERENCE in the build container and (b) the CPU oracle on the same synthetic weights.
Tolerance: bf16 GEMM operands with fp32 accumulation / residual stream vs an fp32 CPU reference:
rel-L2 <= 2e-2 per network forward."""
import pytest
import torch

from conftest import golden, load_synth, manifest, rel_l2

pytestmark = pytest.mark.gpu
TOL = 2e-2


def _build(hidden, depth, heads):
    from ln3diff_amd.dit.dit_trilatent import DiT_TriLatent
    from ln3diff_amd.dit.dit_models_xformers import TextCondDiTBlock
    return DiT_TriLatent(input_size=32, patch_size=2, in_channels=4, hidden_size=hidden, depth=depth, num_heads=heads,
                         num_classes=0, learn_sigma=False, context_dim=768, roll_out=True, vit_blk=TextCondDiTBlock)


def test_t23d_tiny_vs_golden_and_oracle(hip_lib):
    from ln3diff_amd.synth import synth_input
    from oracle import dit as odit
    g = golden('t23d_tiny')
    m = _build(128, 2, 2)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == manifest(g)
    sd, _ = load_synth(m, 0)
    x = synth_input('x', (2, 12, 32, 32), 0)
    t = torch.tensor([999., 37.])
    ctx = synth_input('ctx', (2, 77, 768), 0)
    m = m.cuda()
    y = m(x.cuda(), t.cuda(), ctx.cuda()).cpu()
    y_or = odit.t23d_forward(sd, x, t, ctx, 2)
    assert rel_l2(y_or, g['y']) < 1e-5                 # oracle == reference (CPU)
    e = rel_l2(y, g['y'])
    print('tiny rel-l2 vs reference golden', e)
    assert e < TOL, e


@pytest.mark.parametrize("arch,B,tag", [('DiT-B/2', 1, 't23d_dit_b2'), ('DiT-L/2', 2, 't23d_dit_l2'),
                                        ('DiT-XL/2', 1, 't23d_dit_xl2'), ('DiT-B/1', 1, 't23d_dit_b1')])
def test_t23d_full_vs_golden(hip_lib, arch, B, tag):
    from ln3diff_amd.dit.dit_trilatent import DiT_models
    from ln3diff_amd.dit.dit_models_xformers import TextCondDiTBlock
    from ln3diff_amd.synth import synth_input
    g = golden(tag)
    m = DiT_models[arch](input_size=32, num_classes=0, learn_sigma=False, in_channels=4, context_dim=768, roll_out=True,
                         vit_blk=TextCondDiTBlock)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == manifest(g)
    load_synth(m, 0)
    x = synth_input('x', (B, 12, 32, 32), 0)
    t = torch.from_numpy(g['t']).float()
    ctx = synth_input('ctx', (B, 77, 768), 0)
    m = m.cuda()
    y = m(x.cuda(), t.cuda(), ctx.cuda()).cpu()
    e = rel_l2(y, g['y'])
    print(arch, 'rel-l2 vs reference golden', e)
    assert e < TOL, e
    # context cache + in_scale + batch replication give the same numbers
    cc = m.prepare_context(torch.cat([ctx, ctx]).cuda())
    y2 = m(x.cuda() * 2.0, torch.cat([t, t]).cuda(), context_cache=cc,
           in_scale=torch.full((2 * B,), 0.5, device='cuda')).cpu()
    assert rel_l2(y2[:B], y) < 1e-3 and rel_l2(y2[B:], y) < 1e-3


@pytest.mark.parametrize("tag", ['tiny', 'l2'])
def test_t23d_pixart_forward_with_cfg_vs_reference_golden(hip_lib, tag):
    """DiT_TriLatent_PixelArt (dit_trilatent.py:146-270; 'DiT-PixelArt-L/2'): the T23D denoiser of the flow-matching engine."""
    from ln3diff_amd.dit.dit_trilatent import DiT_models
    from ln3diff_amd.dit.dit_i23d import DiT_TriLatent_PixelArt
    from ln3diff_amd.synth import synth_input
    g = golden(f't23d_pixart_{tag}')
    kw = dict(input_size=32, num_classes=0, learn_sigma=False, in_channels=4, context_dim=768, roll_out=True)
    m = DiT_models['DiT-PixelArt-L/2'](**kw) if tag == 'l2' else DiT_TriLatent_PixelArt(hidden_size=128, depth=2, num_heads=2, patch_size=2, **kw)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == manifest(g)
    load_synth(m, 0)
    m = m.cuda()
    B2 = int(g['y'].shape[0])
    x = synth_input('x', (B2, 12, 32, 32), 3).cuda()
    ctx = {'crossattn': synth_input('c', (B2, 77, 768), 3).cuda(), 'vector': synth_input('v', (B2, 768), 3).cuda()}
    y = m.forward_with_cfg(x, torch.from_numpy(g['t']).cuda(), ctx, 4.0).cpu()
    e = rel_l2(y, g['y'])
    print('t23d pixart', tag, e)
    assert e < 2e-2, e


@pytest.mark.parametrize("arch,B", [('DiT-L/2', 8), ('DiT-B/2', 3), ('DiT-XL/2', 2)])
def test_unconditional_branch_fold_is_exact_algebra(hip_lib, arch, B, monkeypatch):
    """Samples whose context rows are all identical (the ZERO embeddings of the unconditional CFG half, force_uc_zero_embeddings in
    sgm_DiffusionEngine.py:448-452) have a query-independent cross-attention output: prepare_context() computes that constant per
    (layer, sample) and forward() adds it in the proj GEMM's epilogue instead of running to_q / attention / to_out on those rows.
    Checked against the same network with the fold switched off (every sample through the attention kernels), against the CPU
    oracle, and on a batch whose 'unconditional' half is NOT uniform (nothing may be folded)."""
    from ln3diff_amd.dit.dit_trilatent import DiT_models
    from ln3diff_amd.dit.dit_models_xformers import TextCondDiTBlock
    from ln3diff_amd.synth import synth_input
    from oracle import dit as odit
    m = DiT_models[arch](input_size=32, num_classes=0, learn_sigma=False, in_channels=4, context_dim=768, roll_out=True, vit_blk=TextCondDiTBlock)
    sd, _ = load_synth(m, 0)
    m = m.cuda()
    heads = m.num_heads
    x = synth_input('x', (B, 12, 32, 32), 5).cuda()
    x2 = torch.cat([x, x])
    t = torch.full((2 * B,), 617.0).cuda()
    c = synth_input('c', (B, 77, 768), 5).cuda()
    ctx = torch.cat([torch.zeros_like(c), c])                         # VanillaCFG order [uc, c]
    cc = m.prepare_context(ctx)
    assert cc['fold'] == B and cc['const'].shape == (m.depth, 2 * B, m.embed_dim) and float(cc['const'][:, B:].abs().max()) == 0.0
    y_fold = m(x2, t, context_cache=cc)
    monkeypatch.setenv('LN3D_NO_UC_FOLD', '1')
    cc0 = m.prepare_context(ctx)
    assert cc0['fold'] == 0
    y_full = m(x2, t, context_cache=cc0)
    monkeypatch.delenv('LN3D_NO_UC_FOLD')
    e = rel_l2(y_fold, y_full)
    print(arch, 'fold vs no fold', e, 'uncond half', rel_l2(y_fold[:B], y_full[:B]), 'cond half', rel_l2(y_fold[B:], y_full[B:]))
    assert e < 1e-3, e                                                # the same arithmetic up to GEMM tile shape / summation order
    if arch != 'DiT-L/2':                                             # oracle forward on the CPU: the smaller cases only
        y_or = odit.t23d_forward(sd, x2[[0, B]].cpu(), t[:2].cpu(), ctx[[0, B]].cpu(), heads)       # sample 0: its uncond and cond rows
        assert rel_l2(y_fold[[0, B]].cpu(), y_or) < TOL
    # a different prompt in the first half: rows differ inside a sample, nothing is folded, same result as the unfolded path
    ctx2 = torch.cat([synth_input('c', (B, 77, 768), 6).cuda(), c])
    cc2 = m.prepare_context(ctx2)
    assert cc2['fold'] == 0 and 'const' not in cc2
    # only a LEADING run is folded: zeros in the second half stay on the attention path
    cc3 = m.prepare_context(torch.cat([c, torch.zeros_like(c)]))
    assert cc3['fold'] == 0


def test_schedule_modulation_cache_forms(hip_lib):
    """prepare_timesteps(): the timestep-only sub-network evaluated once for the whole schedule.  One row per step when all samples of
    a step share the timestep (read with a sample stride of 0), Bn rows per step otherwise, None above the size guard - every form
    gives the network output of the uncached forward."""
    from ln3diff_amd.dit.dit_trilatent import DiT_TriLatent
    from ln3diff_amd.dit.dit_models_xformers import TextCondDiTBlock
    from ln3diff_amd.synth import synth_input
    m = DiT_TriLatent(input_size=32, patch_size=2, in_channels=4, hidden_size=128, depth=2, num_heads=2, num_classes=0, learn_sigma=False,
                      context_dim=768, roll_out=True, vit_blk=TextCondDiTBlock)
    load_synth(m, 0)
    m = m.cuda()
    Bn = 4
    x = synth_input('x', (Bn, 12, 32, 32), 2).cuda()
    ctx = synth_input('c', (Bn, 77, 768), 2).cuda()
    cc = m.prepare_context(ctx)
    uniform = torch.tensor([900., 500., 17.])[:, None].expand(3, Bn)
    mc = m.prepare_timesteps(uniform)
    assert mc['rows'] == 1 and mc['mod'].shape[0] == 3
    for step in range(3):
        want = m(x, uniform[step].cuda(), context_cache=cc).clone()
        got = m(x, uniform[step].cuda(), context_cache=cc, mod_cache=(mc, step))
        assert rel_l2(got, want) < 1e-5, step
    ragged = torch.tensor([[900., 800., 700., 600.], [5., 6., 7., 8.]])
    mr = m.prepare_timesteps(ragged)
    assert mr['rows'] == Bn and mr['mod'].shape[0] == 2 * Bn
    for step in range(2):
        want = m(x, ragged[step].cuda(), context_cache=cc).clone()
        got = m(x, ragged[step].cuda(), context_cache=cc, mod_cache=(mr, step))
        assert rel_l2(got, want) < 1e-5, step
    m.MODCACHE_MAX_BYTES = 1024
    assert m.prepare_timesteps(uniform) is None          # the samplers then run the modulation GEMMs per step


@pytest.mark.parametrize("folded", [True, False])
def test_cfg_twins_block0_dedup_is_exact_algebra(hip_lib, monkeypatch, folded):
    """r5: with `cfg_twins=True` (the samplers' statement that the two halves of the CFG batch enter with the same latent, timestep
    and input scale) block 0's norm / QKV / self-attention run on one half.  Same network output as the plain forward on the same
    inputs up to summation order and the tile chosen for the halved row count; with and without the zero-context fold."""
    from ln3diff_amd.dit.dit_trilatent import DiT_TriLatent
    from ln3diff_amd.dit.dit_models_xformers import TextCondDiTBlock
    from ln3diff_amd.synth import synth_input
    if not folded:
        monkeypatch.setenv('LN3D_NO_UC_FOLD', '1')
    m = DiT_TriLatent(input_size=32, patch_size=2, in_channels=4, hidden_size=256, depth=3, num_heads=4, num_classes=0, learn_sigma=False,
                      context_dim=768, roll_out=True, vit_blk=TextCondDiTBlock)
    load_synth(m, 0)
    m = m.cuda()
    B = 2
    x = synth_input('x', (B, 12, 32, 32), 2).cuda()                       # Bx = B latents, network batch 2B = [uc ; c]
    c = synth_input('c', (B, 77, 768), 2).cuda()
    cc = m.prepare_context(torch.cat([torch.zeros_like(c), c]))
    assert cc['fold'] == (B if folded else 0)
    sched = torch.tensor([900., 500.])[:, None].expand(2, 2 * B)
    mc = m.prepare_timesteps(sched)
    assert mc['rows'] == 1
    sc = torch.full((2 * B,), 0.37, device='cuda')
    for step in range(2):
        t = sched[step].cuda()
        want = m(x, t, context_cache=cc, in_scale=sc, mod_cache=(mc, step)).clone()
        got = m(x, t, context_cache=cc, in_scale=sc, mod_cache=(mc, step), cfg_twins=True)
        e = rel_l2(got, want)
        print('cfg twins dedup, folded', folded, 'step', step, e)
        assert e < 1e-3, e
    # not applied when the modulation rows differ per sample (per-sample timesteps): the flag alone does not force it
    ragged = torch.tensor([[900., 800., 700., 600.]])
    mr = m.prepare_timesteps(ragged)
    a = m(x, ragged[0].cuda(), context_cache=cc, mod_cache=(mr, 0)).clone()
    b = m(x, ragged[0].cuda(), context_cache=cc, mod_cache=(mr, 0), cfg_twins=True)
    assert torch.equal(a, b)
