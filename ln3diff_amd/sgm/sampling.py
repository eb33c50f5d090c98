"""EDM-style sampling stack of the released T23D checkpoint, device-resident.

Mirrors the reference's sgm surface for this path (same class names / call signatures):
  LegacyDDPMDiscretization   sgm/modules/diffusionmodules/discretizer.py:42-69
  DiscreteDenoiser+EpsScaling sgm/modules/diffusionmodules/denoiser.py:45-78, denoiser_scaling.py:29-37
  VanillaCFG                 sgm/modules/diffusionmodules/guiders.py:24-42
  EulerEDMSampler            sgm/modules/diffusionmodules/sampling.py:82-130,211-215
The sigma tables are built on the host in fp64/fp32 exactly like the reference; the per-step work is one
network call on [uc ; c] (2B) with the c_in scale folded into the patch-embed kernel, and ONE fused
elementwise kernel for denoiser-combine + CFG + Euler update (ln3d_edm_euler_step).  Context K/V are
computed once per call, not once per step.
"""
import numpy as np
import os

import torch

from .. import ops


def make_beta_schedule(n_timestep=1000, linear_start=0.00085, linear_end=0.0120):
    betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2
    return betas.numpy()


def generate_roughly_equally_spaced_steps(num_substeps, max_step):
    return np.linspace(max_step - 1, 0, num_substeps, endpoint=False).astype(int)[::-1]


class LegacyDDPMDiscretization:
    def __init__(self, linear_start=0.00085, linear_end=0.0120, num_timesteps=1000):
        self.num_timesteps = num_timesteps
        self.alphas_cumprod = np.cumprod(1.0 - make_beta_schedule(num_timesteps, linear_start, linear_end), axis=0)

    def get_sigmas(self, n, device="cpu"):
        if n < self.num_timesteps:
            ac = self.alphas_cumprod[generate_roughly_equally_spaced_steps(n, self.num_timesteps)]
        elif n == self.num_timesteps:
            ac = self.alphas_cumprod
        else:
            raise ValueError
        sig = torch.tensor((1 - ac) / ac, dtype=torch.float32, device=device) ** 0.5
        return torch.flip(sig, (0,))

    def __call__(self, n, do_append_zero=True, device="cpu", flip=False):
        s = self.get_sigmas(n, device=device)
        if do_append_zero:
            s = torch.cat([s, s.new_zeros([1])])
        return s if not flip else torch.flip(s, (0,))


class VanillaCFG:
    def __init__(self, scale):
        self.scale = scale


class DiscreteDenoiser:
    """EpsScaling + index quantisation; `sigmas` is the ascending 1000-entry table."""

    def __init__(self, num_idx=1000, discretization=None):
        self.discretization = discretization or LegacyDDPMDiscretization()
        self.sigmas = self.discretization(num_idx, do_append_zero=False, flip=True)
        self.num_idx = num_idx

    def sigma_to_idx(self, sigma):
        return int((self.sigmas - float(sigma)).abs().argmin())

    def quantize(self, sigma):
        i = self.sigma_to_idx(sigma)
        s = float(self.sigmas[i])
        return s, self.sigma_to_idx(s)


class EulerEDMSampler:
    def __init__(self, num_steps=250, guider=None, discretization=None, s_churn=0.0, use_graph=None, **_):
        assert s_churn == 0.0, "released config: gamma = 0 (deterministic)"
        self.num_steps = num_steps
        self.use_graph = use_graph            # None: follow LN3D_GRAPH
        self.guider = guider or VanillaCFG(6.5)
        self.discretization = discretization or LegacyDDPMDiscretization()

    @torch.no_grad()
    def __call__(self, denoiser, network, x, cond, uc=None, num_steps=None, trace=None):
        """x [B,12,32,32] f32 device noise (consumed in place: returns the final latent).
        network(x, t, context_cache=..., in_scale=...) is a ln3diff_amd DiT; cond/uc dicts with 'crossattn'."""
        n = self.num_steps if num_steps is None else num_steps
        sigmas = self.discretization(n, device="cpu")
        B = x.shape[0]
        dev = x.device
        uc = cond if uc is None else uc
        ctx = torch.cat((uc['crossattn'], cond['crossattn']), 0).to(dev)      # VanillaCFG: [uc, c]
        cache = network.prepare_context(ctx)
        x = x * float(torch.sqrt(1.0 + sigmas[0] ** 2.0))
        t_dev = torch.empty(2 * B, device=dev, dtype=torch.float32)
        s_dev = torch.empty(2 * B, device=dev, dtype=torch.float32)
        quant = [denoiser.quantize(sigmas[i]) for i in range(n)]
        mod_all = None
        if hasattr(network, 'prepare_timesteps') and not os.environ.get('LN3D_NO_MODCACHE'):   # timestep-only sub-network for the whole schedule in one pass
            t_table = torch.tensor([float(q[1]) for q in quant], dtype=torch.float32)[:, None].expand(n, 2 * B)
            mod_all = network.prepare_timesteps(t_table)
        # Optional HIP-graph replay of the network evaluation (LN3D_GRAPH=1 or use_graph=True): the ~220 launches of a forward are
        # captured once and replayed per step; what changes between steps goes through fixed device buffers (x in place, t_dev,
        # s_dev, the step's modulation rows copied into mod_step).  The loop is already 99 % kernel-busy without it.
        graph, eps_g, mod_step = None, None, None
        want_graph = self.use_graph if self.use_graph is not None else bool(os.environ.get('LN3D_GRAPH'))
        if want_graph and mod_all is not None and n > 2:
            x = x.contiguous()
            mrows = mod_all['rows']
            mod_step = {'mod': torch.empty_like(mod_all['mod'][:mrows]), 'rows': mrows}
            mod_step['mod'].copy_(mod_all['mod'][:mrows])
            t_dev.fill_(float(quant[0][1]))
            s_dev.fill_(1.0)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):                     # warm-up outside the capture: workspaces, kernel attributes
                network(x, t_dev, context_cache=cache, in_scale=s_dev, mod_cache=(mod_step, 0))
            torch.cuda.current_stream(dev).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                eps_g = network(x, t_dev, context_cache=cache, in_scale=s_dev, mod_cache=(mod_step, 0))
        for i in range(n):
            sig, idx = quant[i]
            c_in = float(1.0 / (torch.tensor(sig, dtype=torch.float32) ** 2 + 1.0) ** 0.5)
            t_dev.fill_(float(idx))
            s_dev.fill_(c_in)
            if graph is not None:
                mod_step['mod'].copy_(mod_all['mod'][i * mrows:(i + 1) * mrows])
                graph.replay()
                eps2 = eps_g
            elif mod_all is not None:
                eps2 = network(x, t_dev, context_cache=cache, in_scale=s_dev, mod_cache=(mod_all, i))
            else:
                eps2 = network(x, t_dev, context_cache=cache, in_scale=s_dev)
            ops.edm_euler_step(x, eps2, sig, float(sigmas[i + 1]), float(self.guider.scale))
            if trace is not None:
                trace.append(x.clone())
        return x
