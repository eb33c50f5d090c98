"""smoke(): one small invocation of the whole hot path on cuda:0, checked against the CPU oracle.
(The oracle is test infrastructure: it is imported here only as the checker.)"""
import torch


def smoke():
    if not torch.cuda.is_available():
        raise RuntimeError("smoke() needs an MI355X: the product path has no CPU fallback")
    from . import _lib
    _lib.check_symbols()
    from .dit.dit_trilatent import DiT_TriLatent
    from .dit.dit_models_xformers import TextCondDiTBlock
    from .nsr.triplane import Triplane, draw_render_noise
    from .sgm.sampling import DiscreteDenoiser, EulerEDMSampler, VanillaCFG
    from .synth import synth_state_dict, synth_input, orbit_cameras
    from oracle import dit as odit, samplers as osamp, render as orender

    dev = torch.device('cuda', 0)
    # -- tiny T23D DiT, 4 EulerEDM steps with CFG
    m = DiT_TriLatent(input_size=32, patch_size=2, in_channels=4, hidden_size=128, depth=2, num_heads=2, num_classes=0,
                      learn_sigma=False, context_dim=768, roll_out=True, vit_blk=TextCondDiTBlock)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = synth_state_dict(shapes, 0, {'pos_embed': m.pos_embed.data})
    m.load_state_dict(sd)
    m = m.to(dev)
    z = synth_input('z', (1, 12, 32, 32), 41)
    cond = {'crossattn': synth_input('c', (1, 77, 768), 41)}
    uc = {'crossattn': torch.zeros(1, 77, 768)}
    y = EulerEDMSampler(num_steps=4, guider=VanillaCFG(6.5))(
        DiscreteDenoiser().bind(m), z.to(dev), {k: v.to(dev) for k, v in cond.items()}, {k: v.to(dev) for k, v in uc.items()})
    y_ref = osamp.edm_euler_sample(lambda x, t, c: odit.t23d_forward(sd, x, t, c, 2), z, cond, uc, 4, 6.5)
    e1 = float((y.cpu() - y_ref).norm() / y_ref.norm())
    # -- render 2 views at 16^2
    tp = Triplane(img_resolution=16)
    dshapes = {'net.0.weight': (64, 32), 'net.0.bias': (64,), 'net.2.weight': (4, 64), 'net.2.bias': (4,)}
    dsd = synth_state_dict(dshapes, 0)
    dsd['net.2.bias'] = dsd['net.2.bias'] + torch.tensor([4., 0, 0, 0])
    tp.decoder.load_state_dict(dsd)
    tp = tp.to(dev)
    planes = synth_input('planes', (2, 96, 128, 128), 3, 4.0)
    cams = orbit_cameras(8)[[1, 6]]
    j, u = draw_render_noise(2, 256, 64, generator=torch.Generator().manual_seed(0))
    out = tp(planes.to(dev), cams.to(dev), jitter=j, u_fine=u)
    ref = orender.triplane_render(planes, dsd, cams, 16, j.unsqueeze(-1), u)
    e2 = float((out['image_raw'].cpu() - ref['image_raw']).norm() / ref['image_raw'].norm())
    print(f"smoke: EDM-4 latent rel-L2 {e1:.2e} (tol 2e-2), render rel-L2 {e2:.2e} (tol 2e-3)")
    assert e1 < 2e-2 and e2 < 2e-3, (e1, e2)
    return e1, e2
