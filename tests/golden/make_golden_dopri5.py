"""Fixture of the adaptive dopri5 solver on the registry-size I23D network (tests/test_i23d_gpu.py): the fp32 CPU restatement
(oracle/samplers.py::flow_ode_dopri5 over oracle/dit.py::i23d_forward_with_cfg, DiT-PixArt-L/2, (name, shape, seed) weights and
inputs, CFG 4, atol 1e-6, rtol 1e-3, num_steps 50) takes ~4 minutes of CPU per run, so its final latent and its step sequence are
stored instead of recomputed on the GPU box.  torchdiffeq is absent from the reference tree and the image: the oracle restates its
published algorithm (parity unpinned against the package).   python tests/golden/make_golden_dopri5.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from conftest import load_synth                      # noqa: E402
from oracle import dit as odit, samplers as osamp    # noqa: E402
from ln3diff_amd.synth import synth_input            # noqa: E402
from ln3diff_amd.dit.dit_i23d import DiT_models      # noqa: E402

m = DiT_models['DiT-PixArt-L/2'](input_size=32, num_classes=0, learn_sigma=False, in_channels=4, context_dim=1024, roll_out=True,
                                 pooling_ctx_dim=768)
sd, _ = load_synth(m, 0)
z = synth_input('z', (1, 12, 32, 32), 42)
cond = {'crossattn': synth_input('ca', (1, 256, 2048), 42), 'vector': synth_input('v', (1, 768), 42)}
ctx = {k: torch.cat([v, torch.zeros_like(v)], 0) for k, v in cond.items()}
zz = torch.cat([z, z])
st = {}
with torch.no_grad():
    y = osamp.flow_ode_dopri5(lambda x, t, **kw: odit.i23d_forward_with_cfg(sd, x, t, kw['context'], kw['cfg_scale'], 16), zz, 50, 1e-6,
                              1e-3, st, context=ctx, cfg_scale=4.0)
print(st)
np.savez_compressed(os.path.join(HERE, 'dopri5_pixartl2_oracle.npz'), final=y.numpy().astype(np.float32), nfe=st['nfe'], steps=st['steps'],
                    accepted=st['accepted'], t_end=st['t_end'], h0=st['h0'], trace=np.array(st['trace'], dtype=np.float64))
