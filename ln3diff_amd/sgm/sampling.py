"""EDM-style sampling stack of the released T23D checkpoint, device-resident.

Mirrors the reference's sgm surface for this path (same class names / call signatures):
  LegacyDDPMDiscretization, EDMDiscretization   sgm/modules/diffusionmodules/discretizer.py:27-69
  Denoiser / DiscreteDenoiser + EpsScaling / VScaling / VScalingWithEDMcNoise / EDMScaling   sgm/modules/diffusionmodules/denoiser.py:13-78, denoiser_scaling.py:14-59
  VanillaCFG                 sgm/modules/diffusionmodules/guiders.py:24-42
  EulerEDMSampler            sgm/modules/diffusionmodules/sampling.py:82-130,211-215
  HeunEDMSampler, EulerAncestralSampler, DPMPP2SAncestralSampler, DPMPP2MSampler, LinearMultistepSampler (r6)   sampling.py:133-365
The sigma tables are built on the host in fp64/fp32 exactly like the reference; the per-step work is one
network call on [uc ; c] (2B) with the c_in scale folded into the patch-embed kernel, and ONE fused
elementwise kernel for denoiser-combine + CFG + Euler update (ln3d_edm_euler_step).  Context K/V are
computed once per call, not once per step.
"""
import math
import os

import numpy as np
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


class EDMDiscretization(LegacyDDPMDiscretization):
    """discretizer.py:27-39: Karras' rho schedule, sigma_i = (sigma_max^(1/rho) + i / (n - 1) (sigma_min^(1/rho) - sigma_max^(1/rho)))^rho."""

    def __init__(self, sigma_min=0.002, sigma_max=80.0, rho=7.0):
        self.sigma_min, self.sigma_max, self.rho = sigma_min, sigma_max, rho

    def get_sigmas(self, n, device="cpu"):
        ramp = torch.linspace(0, 1, n, device=device)
        lo, hi = self.sigma_min ** (1 / self.rho), self.sigma_max ** (1 / self.rho)
        return (hi + ramp * (lo - hi)) ** self.rho


class VanillaCFG:
    """guiders.py:24-42: prepare_inputs doubles the batch as [uc ; c]; __call__ = x_u + scale * (x_c - x_u)."""

    def __init__(self, scale):
        self.scale = scale

    def prepare_inputs(self, x, s, c, uc):
        c_out = {}
        for k in c:
            if k in ("vector", "crossattn", "concat"):
                c_out[k] = torch.cat((uc[k], c[k]), 0)
            else:
                assert c[k] == uc[k]
                c_out[k] = c[k]
        return torch.cat([x] * 2), torch.cat([s] * 2), c_out


class IdentityGuider:
    """guiders.py:45-57 (the reference's default when no guider_config is given): no guidance, the batch is not doubled."""
    scale = None

    def prepare_inputs(self, x, s, c, uc):
        return x, s, {k: c[k] for k in c}


def _guided_terms(guider, den, B, coef):
    """(tensors, coefficients) of coef * guider(denoised) for one ln3d_lincomb: VanillaCFG's x_u + s (x_c - x_u) as two weighted halves,
    IdentityGuider's single batch as is."""
    if getattr(guider, 'scale', None) is None:
        return [den], [coef]
    sc = float(guider.scale)
    return [den[:B], den[B:]], [coef * (1.0 - sc), coef * sc]


def _f32(v):
    return torch.tensor(float(v), dtype=torch.float32)


class EpsScaling:
    """denoiser_scaling.py:29-37 (the released configuration): c_skip 1, c_out -sigma, c_in 1 / sqrt(sigma^2 + 1), c_noise sigma - evaluated in
    fp32 like the reference's tensors."""

    def __call__(self, sigma):
        s = _f32(sigma)
        return 1.0, float(-s), float(1.0 / (s ** 2 + 1.0) ** 0.5), float(s)


class VScaling:
    """denoiser_scaling.py:40-48"""

    def __call__(self, sigma):
        s = _f32(sigma)
        return float(1.0 / (s ** 2 + 1.0)), float(-s / (s ** 2 + 1.0) ** 0.5), float(1.0 / (s ** 2 + 1.0) ** 0.5), float(s)


class VScalingWithEDMcNoise(VScaling):
    """denoiser_scaling.py:51-59"""

    def __call__(self, sigma):
        c_skip, c_out, c_in, _ = super().__call__(sigma)
        return c_skip, c_out, c_in, float(0.25 * _f32(sigma).log())


class EDMScaling:
    """denoiser_scaling.py:14-26"""

    def __init__(self, sigma_data=0.5):
        self.sigma_data = float(sigma_data)

    def __call__(self, sigma):
        s, d = _f32(sigma), _f32(self.sigma_data)
        return (float(d ** 2 / (s ** 2 + d ** 2)), float(s * d / (s ** 2 + d ** 2) ** 0.5), float(1 / (s ** 2 + d ** 2) ** 0.5),
                float(0.25 * s.log()))


class Denoiser:
    """denoiser.py:13-42: c_skip * input + c_out * network(c_in * input, c_noise, cond) with the scaling's four factors; no quantisation
    (the network sees c_noise as a float).  One sigma per call is assumed to be shared by the batch (every sampler of this path)."""

    def __init__(self, scaling=None):
        self.scaling = scaling or EpsScaling()

    def quantize(self, sigma):
        """(sigma the scalings are evaluated at, None): the continuous denoiser passes sigma through"""
        return float(sigma), None

    def noise_label(self, c_noise):
        return float(c_noise)

    @torch.no_grad()
    def __call__(self, network, input, sigma, cond, **additional_model_inputs):
        """`network` is a ln3diff_amd DiT (c_in rides on its patch-embed kernel) or any callable (x, t, cond) -> output on the device."""
        sig, _ = self.quantize(float(sigma.reshape(-1)[0]))
        c_skip, c_out, c_in, c_noise = self.scaling(sig)
        n = input.shape[0]
        t = torch.full((n,), self.noise_label(c_noise), device=input.device, dtype=torch.float32)
        if hasattr(network, 'prepare_context'):
            eps = network(input, t, context=cond, in_scale=torch.full((n,), c_in, device=input.device, dtype=torch.float32),
                          **additional_model_inputs)
        else:
            eps = network(input * c_in, t, cond, **additional_model_inputs)
        out = torch.empty_like(input, dtype=torch.float32)
        xin = input.contiguous().float()
        if c_skip == 1.0:
            ops.lincomb(xin, [eps.contiguous().float()], [c_out], out)
        else:
            ops.lincomb(None, [xin, eps.contiguous().float()], [c_skip, c_out], out)
        return out

    def bind(self, network, **additional_model_inputs):
        """The closure DiffusionEngineLSGM.sample hands to its sampler (sgm_DiffusionEngine.py:401-403):
        `lambda input, sigma, c: self.denoiser(self.model, input, sigma, c, **kwargs)` - as an object the sampler can look into."""
        return BoundDenoiser(self, network, **additional_model_inputs)


class DiscreteDenoiser(Denoiser):
    """denoiser.py:45-78: sigma snapped to the ascending 1000-entry table before the scalings are evaluated, c_noise replaced by its table index
    (quantize_c_noise).  With EpsScaling (the released configuration) the network's timestep is the index of the snapped sigma."""

    def __init__(self, num_idx=1000, discretization=None, scaling=None, quantize_c_noise=True):
        super().__init__(scaling)
        self.discretization = discretization or LegacyDDPMDiscretization()
        self.sigmas = self.discretization(num_idx, do_append_zero=False, flip=True)
        self.num_idx = num_idx
        self.quantize_c_noise = bool(quantize_c_noise)

    def sigma_to_idx(self, sigma):
        return int((self.sigmas - float(sigma)).abs().argmin())

    def quantize(self, sigma):
        i = self.sigma_to_idx(sigma)
        s = float(self.sigmas[i])
        return s, self.sigma_to_idx(s)

    def noise_label(self, c_noise):
        return float(self.sigma_to_idx(c_noise)) if self.quantize_c_noise else float(c_noise)

    @property
    def fused_ok(self):
        """the fused network loop of EulerEDMSampler bakes EpsScaling + the index timestep in"""
        return type(self.scaling) is EpsScaling and self.quantize_c_noise

class BoundDenoiser:
    def __init__(self, denoiser, network, **additional_model_inputs):
        self.denoiser, self.network, self.kw = denoiser, network, additional_model_inputs

    def __call__(self, input, sigma, c):
        return self.denoiser(self.network, input, sigma, c, **self.kw)


_REF_LAMBDA_OPS = {'LOAD_DEREF', 'LOAD_ATTR', 'LOAD_METHOD', 'LOAD_FAST', 'BUILD_TUPLE', 'BUILD_MAP', 'DICT_MERGE', 'CALL_FUNCTION_EX',
                   'CALL_FUNCTION', 'CALL_METHOD', 'CALL_FUNCTION_KW', 'RETURN_VALUE', 'RESUME', 'COPY_FREE_VARS', 'PUSH_NULL', 'PRECALL', 'CALL',
                   'LOAD_CONST', 'KW_NAMES'}
_CALL_OPS = {'CALL_FUNCTION_EX', 'CALL_FUNCTION', 'CALL_METHOD', 'CALL_FUNCTION_KW', 'CALL'}


def _is_reference_lambda(fn):
    """True when `fn`'s code is the engine's closure and nothing else (sgm_DiffusionEngine.py:401-403):
    `lambda input, sigma, c: self.denoiser(self.model, input, sigma, c, **additional_model_inputs)` - three positional arguments, ONE call,
    the attribute names `denoiser` and `model` only, every argument passed through untouched.  A closure that post-processes the
    denoised output, wraps the network or adds inputs has other opcodes / names and is run as written (generic loop)."""
    import dis
    code = getattr(fn, '__code__', None)
    if code is None or code.co_argcount != 3 or code.co_kwonlyargcount or (code.co_flags & 0x0C):      # no *args / **kwargs of its own
        return False
    if set(code.co_names) != {'denoiser', 'model'}:
        return False
    ins = list(dis.get_instructions(code))
    if any(i.opname not in _REF_LAMBDA_OPS for i in ins) or sum(i.opname in _CALL_OPS for i in ins) != 1:
        return False
    fast = [i.argval for i in ins if i.opname == 'LOAD_FAST']
    return fast == list(code.co_varnames[:3])                  # input, sigma, c: each once, in order


def _find_pair(denoiser):
    """(DiscreteDenoiser, ln3diff_amd network) behind the sampler's `denoiser` argument, or (None, None).
    Recognised (nothing heuristic): `BoundDenoiser` without extra inputs; a callable that opts in with
    `_ln3d_pair = (denoiser, network)`; a closure whose CODE is exactly the reference's lambda over an engine object with
    `.denoiser` / `.model` and an empty `additional_model_inputs` (_is_reference_lambda).  Anything else runs the generic loop."""
    if isinstance(denoiser, BoundDenoiser):
        return (denoiser.denoiser, denoiser.network) if not denoiser.kw and hasattr(denoiser.network, 'prepare_context') else (None, None)
    pair = getattr(denoiser, '_ln3d_pair', None)
    if pair is not None:
        den, net = pair
        return (den, net) if isinstance(den, DiscreteDenoiser) and hasattr(net, 'prepare_context') else (None, None)
    if not _is_reference_lambda(denoiser):
        return None, None
    den = net = None
    for cell in getattr(denoiser, '__closure__', None) or ():
        try:
            o = cell.cell_contents
        except ValueError:
            return None, None
        if isinstance(getattr(o, 'denoiser', None), DiscreteDenoiser) and hasattr(getattr(o, 'model', None), 'prepare_context'):
            if den is not None:
                return None, None
            den, net = o.denoiser, o.model
        elif isinstance(o, dict):
            if o:
                return None, None                             # **additional_model_inputs forwarded to the network: not the plain pair
        else:
            return None, None                                 # a cell the reference lambda does not have
    return (den, net) if den is not None else (None, None)


class EulerEDMSampler:
    """sampling.py:82-130,211-215: EDMSampler + the Euler step.  The released configuration is s_churn = 0 (gamma = 0: deterministic);
    s_churn / s_tmin / s_tmax / s_noise > 0 (r6) add the reference's noise injection: where s_tmin <= sigma_i <= s_tmax,
    gamma = min(s_churn / (num_sigmas - 1), sqrt 2 - 1), sigma_hat = sigma_i (1 + gamma), x += randn * s_noise * sqrt(sigma_hat^2 -
    sigma_i^2), and the denoiser runs at sigma_hat (quantised to the 1000-entry table for c_out / c_in / the timestep, the Euler step
    itself on sigma_hat).  The draw of step i comes from `step_noise(i)` (a [B, ...] tensor; parity runs feed the recorded stream) or
    torch.randn on the device."""

    def __init__(self, num_steps=250, guider=None, discretization=None, s_churn=0.0, s_tmin=0.0, s_tmax=float('inf'), s_noise=1.0,
                 use_graph=None, **_):
        self.num_steps = num_steps
        self.use_graph = use_graph            # None: follow LN3D_GRAPH
        self.s_churn, self.s_tmin, self.s_tmax, self.s_noise = float(s_churn), float(s_tmin), float(s_tmax), float(s_noise)
        self.guider = guider or VanillaCFG(6.5)
        self.discretization = discretization or LegacyDDPMDiscretization()

    def _gammas(self, sigmas):
        """gamma per step (sampling.py:114-118): num_sigmas = len(sigmas) counts the appended zero."""
        n = len(sigmas)
        g = min(self.s_churn / (n - 1), 2 ** 0.5 - 1)
        return [g if (self.s_churn > 0 and self.s_tmin <= float(sigmas[i]) <= self.s_tmax) else 0.0 for i in range(n - 1)]

    @torch.no_grad()
    def __call__(self, denoiser, x, cond, uc=None, num_steps=None, *, network=None, trace=None, step_noise=None):
        """The reference's call (sampling.py:109): denoiser = the engine's closure (input, sigma, c) -> denoised; x [B,12,32,32]
        f32 device noise (consumed in place: returns the final latent); cond / uc dicts with 'crossattn'.
        Fast path (context K/V, the timestep sub-network of the whole schedule and the CFG + Euler update fused): taken when the
        closure is recognisably this package's DiscreteDenoiser over one of its networks (`DiscreteDenoiser.bind(network)`, the
        reference's own lambda over an engine, or `network=` with a DiscreteDenoiser).  Any other callable runs the generic loop
        with the same arithmetic, one closure call per step."""
        if network is not None and isinstance(denoiser, Denoiser):
            den, net = denoiser, network
        else:
            den, net = _find_pair(denoiser)
        if net is None or isinstance(self.guider, IdentityGuider) or not getattr(den, 'fused_ok', False):   # the fused loop: CFG-doubled, EpsScaling
            gen = denoiser if net is None else (lambda x_, s_, c_: den(net, x_, s_, c_))
            return self._generic(gen, x, cond, uc, num_steps, trace, step_noise)
        uc = cond if uc is None else uc
        return self._fast(den, net, x, cond, uc, num_steps, trace, step_noise)

    # ------------------------------------------------------------------ generic: the reference loop, one closure call per step
    def _churn(self, x, sig, gamma, i, step_noise):
        """x += randn * s_noise * sqrt(sigma_hat^2 - sigma^2) in place; returns sigma_hat (sampling.py:96-99)."""
        sig_hat = sig * (gamma + 1.0)
        eps = step_noise(i).to(x.device).float().contiguous() if step_noise is not None else torch.randn(x.shape, device=x.device)
        ops.lincomb(x, [eps], [self.s_noise * (sig_hat ** 2 - sig ** 2) ** 0.5], x)
        return sig_hat

    def _generic(self, denoiser, x, cond, uc, num_steps, trace, step_noise=None):
        n = self.num_steps if num_steps is None else num_steps
        sigmas = self.discretization(n, device="cpu")
        gammas = self._gammas(sigmas)
        uc = cond if uc is None else uc
        x = (x * float(torch.sqrt(1.0 + sigmas[0] ** 2.0))).contiguous()
        B = x.shape[0]
        s_in = x.new_ones([B])
        for i in range(n):
            sig, nxt = float(sigmas[i]), float(sigmas[i + 1])
            if gammas[i] > 0:
                sig = self._churn(x, sig, gammas[i], i, step_noise)
            den = denoiser(*self.guider.prepare_inputs(x, s_in * sig, cond, uc)).contiguous().float()
            # guider + to_d + euler_step in one combination: x + dt/sigma * (x - (x_u + s (x_c - x_u)))
            r = (nxt - sig) / sig
            dk, dcf = _guided_terms(self.guider, den, B, -r)
            ops.lincomb(x, [x] + dk, [r] + dcf, x)
            if trace is not None:
                trace.append(x.clone())
        return x

    # ------------------------------------------------------------------ fast path pieces
    def _prepare(self, den, network, x, cond, uc, n):
        sigmas = self.discretization(n, device="cpu")
        B, dev = x.shape[0], x.device
        ctx = torch.cat((uc['crossattn'], cond['crossattn']), 0).to(dev)      # VanillaCFG: [uc, c]
        gammas = self._gammas(sigmas)
        st = {'cache': network.prepare_context(ctx), 'sigmas': sigmas, 'B': B, 'gammas': gammas,
              't_dev': torch.empty(2 * B, device=dev, dtype=torch.float32), 's_dev': torch.empty(2 * B, device=dev, dtype=torch.float32),
              # the denoiser sees sigma_hat = sigma (1 + gamma), quantised to its table (denoiser.py:66-78)
              'quant': [den.quantize(float(sigmas[i]) * (gammas[i] + 1.0)) for i in range(n)]}
        st['x'] = x * float(torch.sqrt(1.0 + sigmas[0] ** 2.0))
        return st

    def _mod_all(self, network, quant, n, B):
        if not hasattr(network, 'prepare_timesteps'):
            return None
        # timestep-only sub-network for the whole schedule in one pass
        t_table = torch.tensor([float(q[1]) for q in quant], dtype=torch.float32)[:, None].expand(n, 2 * B)
        return network.prepare_timesteps(t_table)

    def _step(self, network, st, i, mod_all, step_noise=None):
        sig, idx = st['quant'][i]
        gamma = st['gammas'][i]
        if gamma > 0:
            sig_hat = self._churn(st['x'], float(st['sigmas'][i]), gamma, i, step_noise)
        c_in = float(1.0 / (torch.tensor(sig, dtype=torch.float32) ** 2 + 1.0) ** 0.5)
        st['t_dev'].fill_(float(idx))
        st['s_dev'].fill_(c_in)
        g = st.get('graph')
        if g is not None:
            g['mod_step']['mod'].copy_(mod_all['mod'][i * g['mrows']:(i + 1) * g['mrows']])
            g['graph'].replay()
            eps2 = g['eps']
        elif mod_all is not None:      # cfg_twins: [uc ; c] are the same latents, timestep and c_in twice (VanillaCFG.prepare_inputs)
            eps2 = network(st['x'], st['t_dev'], context_cache=st['cache'], in_scale=st['s_dev'], mod_cache=(mod_all, i), cfg_twins=True)
        else:
            eps2 = network(st['x'], st['t_dev'], context_cache=st['cache'], in_scale=st['s_dev'], cfg_twins=True)
        if gamma > 0:
            # denoised = x - sig_q (eps_u + s (eps_c - eps_u)); d = (x - denoised) / sigma_hat; x += d (sigma_next - sigma_hat): sig_q (the table
            # entry c_out uses) and sigma_hat differ here, which the fused kernel's single sigma cannot express - one linear combination instead
            B, sc = st['B'], float(self.guider.scale)
            r = (float(st['sigmas'][i + 1]) - sig_hat) / sig_hat * sig
            e2 = eps2.reshape(2, -1)
            ops.lincomb(st['x'], [e2[0].reshape(st['x'].shape), e2[1].reshape(st['x'].shape)], [r * (1.0 - sc), r * sc], st['x'])
        else:
            ops.edm_euler_step(st['x'], eps2, sig, float(st['sigmas'][i + 1]), float(self.guider.scale))

    def _capture(self, network, st, mod_all, dev):
        """Optional HIP-graph replay of the network evaluation (LN3D_GRAPH=1 or use_graph=True): the ~220 launches of a forward
        are captured once and replayed per step; what changes between steps goes through fixed device buffers (x in place,
        t_dev, s_dev, the step's modulation rows copied into mod_step)."""
        st['x'] = st['x'].contiguous()
        mrows = mod_all['rows']
        mod_step = {'mod': torch.empty_like(mod_all['mod'][:mrows]), 'rows': mrows}
        mod_step['mod'].copy_(mod_all['mod'][:mrows])
        st['t_dev'].fill_(float(st['quant'][0][1]))
        st['s_dev'].fill_(1.0)
        side = _capture_stream(dev)                      # ONE stream per device for warm-up and capture
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                     # warm-up outside the capture: workspaces (allocated and zeroed here, once), kernel attributes
            network(st['x'], st['t_dev'], context_cache=st['cache'], in_scale=st['s_dev'], mod_cache=(mod_step, 0), cfg_twins=True)
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            eps_g = network(st['x'], st['t_dev'], context_cache=st['cache'], in_scale=st['s_dev'], mod_cache=(mod_step, 0), cfg_twins=True)
        st['graph'] = {'graph': graph, 'eps': eps_g, 'mod_step': mod_step, 'mrows': mrows}

    def _fast(self, den, network, x, cond, uc, num_steps, trace, step_noise=None):
        n = self.num_steps if num_steps is None else num_steps
        st = self._prepare(den, network, x, cond, uc, n)
        mod_all = self._mod_all(network, st['quant'], n, st['B'])
        want_graph = self.use_graph if self.use_graph is not None else bool(os.environ.get('LN3D_GRAPH'))
        if want_graph and mod_all is not None and n > 2:
            self._capture(network, st, mod_all, x.device)
        for i in range(n):
            self._step(network, st, i, mod_all, step_noise)
            if trace is not None:
                trace.append(st['x'].clone())
        return st['x']


_CAPTURE_STREAMS = {}


# ---------------------------------------------------------------------------------------------------------------------------------
# r6: the other samplers of sgm/modules/diffusionmodules/sampling.py (HeunEDMSampler 218-236, EulerAncestralSampler 133-170 / 239-246,
# DPMPP2SAncestralSampler 249-287, DPMPP2MSampler 290-365).  None of the released launchers selects them; they are here so that a sampler
# config of the reference's family resolves.  Host loops over the same device pieces as EulerEDMSampler's generic path: one closure call per
# network evaluation (the CFG-doubled batch), every update ONE ln3d_lincomb launch with the guidance x_u + s (x_c - x_u) folded into its
# coefficients (all of these updates are linear in x and the denoised halves).  sigma is one number per step (s_in * sigma in the reference),
# so the step sizes are host scalars.  LinearMultistepSampler's weights are the closed-form integrals of the Lagrange basis (the reference integrates the same polynomials numerically).
def _closure(denoiser, network):
    """(input, sigma, c) -> denoised [2B, ...]: the reference's lambda / BoundDenoiser as given, or DiscreteDenoiser + network= bound here."""
    if network is not None and isinstance(denoiser, Denoiser):
        return lambda x, s, c: denoiser(network, x, s, c)
    return denoiser


class _LoopSampler:
    def __init__(self, num_steps=250, guider=None, discretization=None, **_):
        self.num_steps = num_steps
        self.guider = guider or VanillaCFG(6.5)
        self.discretization = discretization or LegacyDDPMDiscretization()

    def _setup(self, denoiser, x, cond, uc, num_steps, network):
        n = self.num_steps if num_steps is None else num_steps
        sigmas = [float(v) for v in self.discretization(n, device="cpu")]
        x = (x * float((1.0 + sigmas[0] ** 2.0) ** 0.5)).contiguous()
        return n, sigmas, _closure(denoiser, network), (cond if uc is None else uc), x

    def _den(self, call, x, sig, cond, uc, keep=False):
        """the denoised batch at noise level sig as the guider laid it out ([uc ; c] for VanillaCFG); keep = it must survive the next call"""
        B = x.shape[0]
        out = call(*self.guider.prepare_inputs(x, x.new_ones([B]) * sig, cond, uc)).contiguous().float()
        return out.clone() if keep else out

    def _g(self, den, x, coef):
        return _guided_terms(self.guider, den, x.shape[0], coef)

    @staticmethod
    def _noise(x, i, step_noise):
        return step_noise(i).to(x.device).float().contiguous() if step_noise is not None else torch.randn(x.shape, device=x.device)


def _ancestral(sig, nxt, eta):
    """sampling_utils.get_ancestral_step (sampling_utils.py:22-31) on scalars"""
    if not eta:
        return nxt, 0.0
    up = min(nxt, eta * (nxt ** 2 * (sig ** 2 - nxt ** 2) / sig ** 2) ** 0.5)
    return (nxt ** 2 - up ** 2) ** 0.5, up


class HeunEDMSampler(EulerEDMSampler):
    """sampling.py:218-236: EDMSampler's step (noise injection included) + the trapezoidal correction, a second network evaluation per step
    except onto sigma = 0.  Always the generic loop (the fused network loop of EulerEDMSampler is a one-evaluation-per-step schedule)."""

    @torch.no_grad()
    def __call__(self, denoiser, x, cond, uc=None, num_steps=None, *, network=None, trace=None, step_noise=None):
        n = self.num_steps if num_steps is None else num_steps
        sig_t = self.discretization(n, device="cpu")
        gammas = self._gammas(sig_t)
        sigmas = [float(v) for v in sig_t]
        call = _closure(denoiser, network)
        uc = cond if uc is None else uc
        x = (x * float((1.0 + sigmas[0] ** 2.0) ** 0.5)).contiguous()
        B = x.shape[0]
        xe = torch.empty_like(x)
        den = _LoopSampler._den
        for i in range(n):
            sig, nxt = sigmas[i], sigmas[i + 1]
            if gammas[i] > 0:
                sig = self._churn(x, sig, gammas[i], i, step_noise)
            d1 = den(self, call, x, sig, cond, uc, keep=True)
            r = (nxt - sig) / sig
            k1, c1 = _guided_terms(self.guider, d1, B, -r)
            ops.lincomb(x, [x] + k1, [r] + c1, xe)                                               # the Euler step
            if nxt < 1e-14:
                x.copy_(xe)
            else:                                                                                # x + dt ((x - D) / sig + (x_e - D2) / nxt) / 2
                d2 = den(self, call, xe, nxt, cond, uc)
                a, b = (nxt - sig) / (2.0 * sig), (nxt - sig) / (2.0 * nxt)
                k1, c1 = _guided_terms(self.guider, d1, B, -a)
                k2, c2 = _guided_terms(self.guider, d2, B, -b)
                ops.lincomb(x, [x] + k1 + [xe] + k2, [a] + c1 + [b] + c2, x)
            if trace is not None:
                trace.append(x.clone())
        return x


class EulerAncestralSampler(_LoopSampler):
    """sampling.py:133-170, 239-246: Euler step to sigma_down, then s_noise * sigma_up of fresh noise where the next sigma is not 0.
    step_noise(i) ([B, ...] tensor) replaces the device draw of step i (parity runs; the reference draws at every step)."""

    def __init__(self, eta=1.0, s_noise=1.0, **kw):
        super().__init__(**kw)
        self.eta, self.s_noise = float(eta), float(s_noise)

    def _ancestral_noise(self, x, i, nxt, up, step_noise):
        if nxt > 0.0:
            ops.lincomb(x, [self._noise(x, i, step_noise)], [self.s_noise * up], x)

    @torch.no_grad()
    def __call__(self, denoiser, x, cond, uc=None, num_steps=None, *, network=None, trace=None, step_noise=None):
        n, sigmas, call, uc, x = self._setup(denoiser, x, cond, uc, num_steps, network)
        for i in range(n):
            sig, nxt = sigmas[i], sigmas[i + 1]
            down, up = _ancestral(sig, nxt, self.eta)
            d1 = self._den(call, x, sig, cond, uc)
            r = (down - sig) / sig
            k1, c1 = self._g(d1, x, -r)
            ops.lincomb(x, [x] + k1, [r] + c1, x)
            self._ancestral_noise(x, i, nxt, up, step_noise)
            if trace is not None:
                trace.append(x.clone())
        return x


class DPMPP2SAncestralSampler(EulerAncestralSampler):
    """sampling.py:249-287: exponential-integrator midpoint step in t = -log sigma towards sigma_down (second evaluation at sigma(t + h / 2)),
    the Euler step when sigma_down is 0, then the ancestral noise."""

    @torch.no_grad()
    def __call__(self, denoiser, x, cond, uc=None, num_steps=None, *, network=None, trace=None, step_noise=None):
        n, sigmas, call, uc, x = self._setup(denoiser, x, cond, uc, num_steps, network)
        x2 = torch.empty_like(x)
        for i in range(n):
            sig, nxt = sigmas[i], sigmas[i + 1]
            down, up = _ancestral(sig, nxt, self.eta)
            d1 = self._den(call, x, sig, cond, uc)
            if down < 1e-14:
                r = (down - sig) / sig
                k1, c1 = self._g(d1, x, -r)
                ops.lincomb(x, [x] + k1, [r] + c1, x)
            else:
                t, t_next = -math.log(sig), -math.log(down)
                h = t_next - t
                sm = t + 0.5 * h
                m1, m2 = math.exp(-sm) / math.exp(-t), math.expm1(-0.5 * h)
                m3, m4 = math.exp(-t_next) / math.exp(-t), math.expm1(-h)
                k1, c1 = self._g(d1, x, -m2)
                ops.lincomb(None, [x] + k1, [m1] + c1, x2)                                         # x2 = m1 x - m2 D
                d2 = self._den(call, x2, math.exp(-sm), cond, uc)
                k2, c2 = self._g(d2, x, -m4)
                ops.lincomb(None, [x] + k2, [m3] + c2, x)                                          # x = m3 x - m4 D2
            self._ancestral_noise(x, i, nxt, up, step_noise)
            if trace is not None:
                trace.append(x.clone())
        return x


class DPMPP2MSampler(_LoopSampler):
    """sampling.py:290-365: second-order multistep - the previous step's denoised output extrapolates the current one by the ratio of the two
    log-sigma steps; the first step and the step onto sigma = 0 are first order (that last step returns the denoised sample itself)."""

    @torch.no_grad()
    def __call__(self, denoiser, x, cond, uc=None, num_steps=None, *, network=None, trace=None, **_):
        n, sigmas, call, uc, x = self._setup(denoiser, x, cond, uc, num_steps, network)
        old = None
        for i in range(n):
            sig, nxt = sigmas[i], sigmas[i + 1]
            d1 = self._den(call, x, sig, cond, uc, keep=True)
            t = -math.log(sig)
            if nxt < 1e-14:                                   # t_next = +inf: mult1 = 0, mult2 = expm1(-inf) = -1
                m1, m2, h = 0.0, -1.0, float('inf')
            else:
                h = -math.log(nxt) - t
                m1, m2 = nxt / sig, math.expm1(-h)
            if old is None or nxt < 1e-14:
                k1, c1 = self._g(d1, x, -m2)
                ops.lincomb(None, [x] + k1, [m1] + c1, x)
            else:
                r = (t + math.log(sigmas[i - 1])) / h
                m3, m4 = 1.0 + 1.0 / (2.0 * r), 1.0 / (2.0 * r)
                k1, c1 = self._g(d1, x, -m2 * m3)
                k0, c0 = self._g(old, x, m2 * m4)
                ops.lincomb(None, [x] + k1 + k0, [m1] + c1 + c0, x)
            old = d1
            if trace is not None:
                trace.append(x.clone())
        return x


def linear_multistep_coeff(order, t, i, j):
    """The Adams-Bashforth weight of derivative j steps back for the step t[i] -> t[i + 1]: the integral over that interval of the Lagrange
    basis polynomial through the last `order` nodes t[i], t[i - 1], ... (sampling_utils.py:7-19 evaluates the same integral with scipy's
    adaptive quadrature at epsrel 1e-4; the integrand is a polynomial of degree < order, so the closed form below is that value exactly)."""
    if order - 1 > i:
        raise ValueError(f"Order {order} too high for step {i}")
    nodes = [float(t[i - k]) for k in range(order)]
    basis = np.poly1d([1.0])
    for k in range(order):
        if k != j:
            basis = basis * np.poly1d([1.0, -nodes[k]]) / (nodes[j] - nodes[k])
    prim = basis.integ()
    return float(prim(float(t[i + 1])) - prim(float(t[i])))


class LinearMultistepSampler(_LoopSampler):
    """sampling.py:172-208: Adams-Bashforth in sigma over the last `order` derivatives d = (x - denoised) / sigma; one evaluation per step.
    Each d is one ln3d_lincomb (guidance folded in), the update another."""

    def __init__(self, order=4, **kw):
        super().__init__(**kw)
        self.order = int(order)

    @torch.no_grad()
    def __call__(self, denoiser, x, cond, uc=None, num_steps=None, *, network=None, trace=None, **_):
        n, sigmas, call, uc, x = self._setup(denoiser, x, cond, uc, num_steps, network)
        sig_np = np.asarray(self.discretization(n, device="cpu"), dtype=np.float32)     # the reference's fp32 sigma table (sigmas.cpu().numpy())
        ds = []
        for i in range(n):
            d1 = self._den(call, x, sigmas[i], cond, uc)
            d = torch.empty_like(x) if len(ds) < self.order else ds.pop(0)
            inv = 1.0 / sigmas[i]
            k1, c1 = self._g(d1, x, -inv)
            ops.lincomb(None, [x] + k1, [inv] + c1, d)
            ds.append(d)
            cur = min(i + 1, self.order)
            coeffs = [linear_multistep_coeff(cur, sig_np, i, j) for j in range(cur)]
            ops.lincomb(x, list(reversed(ds))[:cur], coeffs, x)
            if trace is not None:
                trace.append(x.clone())
        return x


def _capture_stream(dev):
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _CAPTURE_STREAMS:
        _CAPTURE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _CAPTURE_STREAMS[key]
