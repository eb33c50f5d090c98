"""Flow-matching (SiT transport) ODE / SDE sampling on the HIP path.

Surface of the reference's transport package for the sampling path: create_transport(...),
Sampler(transport).sample_ode(sampling_method, num_steps, atol, rtol, reverse)(x, model_fn, **kw) -> [T, ...]
(transport/__init__.py:3-71, transport/transport.py:374-420, transport/integrators.py:78-120).
Linear path + velocity prediction: dx/dt = model(x, t), t from 0 (noise) to 1 (data).  Fixed-grid
'euler', 'heun', 'midpoint' and 'rk4' (torchdiffeq's fixed-grid definitions: its RK4 is the 3/8 rule) plus adaptive Dormand-Prince 5(4) with dense output ('dopri5', the reference's default; restated from the
published algorithm of the absent third-party torchdiffeq 0.2.3: parity unpinned against the package - SURVEY.md §8c; pinned to
oracle/samplers.py's restatement step for step, and validated by convergence to the fixed-step solution).
"""
import torch

from .. import ops


class Transport:
    def __init__(self, path_type='Linear', prediction='velocity', train_eps=0, sample_eps=0, snr_type='uniform'):
        assert path_type == 'Linear' and prediction == 'velocity'
        self.train_eps, self.sample_eps, self.snr_type = train_eps, sample_eps, snr_type

    def check_interval(self, *a, **k):
        return 0, 1


def create_transport(path_type='Linear', prediction='velocity', loss_weight=None, train_eps=None, sample_eps=None,
                     snr_type='uniform'):
    return Transport(path_type, prediction, 0, 0, snr_type)


class Sampler:
    def __init__(self, transport):
        self.transport = transport

    def sample_ode(self, *, sampling_method="dopri5", num_steps=50, atol=1e-6, rtol=1e-3, reverse=False, cfg=False):
        if sampling_method == "dopri5":
            return _dopri5_sampler(num_steps, atol, rtol)
        if sampling_method not in ("euler", "heun", "midpoint", "rk4"):
            raise NotImplementedError(f"ODE method '{sampling_method}' (fixed grid: euler, heun, midpoint, rk4; adaptive: dopri5)")
        assert not reverse
        ts = torch.linspace(0.0, 1.0, num_steps)

        @torch.no_grad()
        def sample(x, model_fn, return_trajectory=True, **model_kwargs):
            dev = x.device
            x = x.clone().float()
            t_dev = torch.empty(x.shape[0], device=dev, dtype=torch.float32)
            traj = [x.clone()] if return_trajectory else None
            for i in range(num_steps - 1):
                t0, t1 = float(ts[i]), float(ts[i + 1])
                dt = float(ts[i + 1] - ts[i])
                t_dev.fill_(t0)
                k1 = model_fn(x, t_dev, **model_kwargs)
                if sampling_method == "euler":
                    ops.axpby(k1, x, dt, 1.0)
                elif sampling_method == "midpoint":      # torchdiffeq Midpoint._step_func: dt * f(t0 + dt/2, y0 + dt/2 f0)
                    xm = torch.empty_like(x)
                    ops.lincomb(x, [k1], [0.5 * dt], xm)
                    t_dev.fill_(t0 + 0.5 * dt)
                    k2 = model_fn(xm, t_dev, **model_kwargs)
                    ops.axpby(k2, x, dt, 1.0)
                elif sampling_method == "rk4":           # torchdiffeq RK4 = rk4_alt_step_func, the 3/8 rule (rk_common.py)
                    xs = torch.empty_like(x)
                    ops.lincomb(x, [k1], [dt / 3], xs)
                    t_dev.fill_(t0 + dt / 3)
                    k2 = model_fn(xs, t_dev, **model_kwargs)
                    xs = torch.empty_like(x)
                    ops.lincomb(x, [k2, k1], [dt, -dt / 3], xs)
                    t_dev.fill_(t0 + dt * 2 / 3)
                    k3 = model_fn(xs, t_dev, **model_kwargs)
                    xs = torch.empty_like(x)
                    ops.lincomb(x, [k1, k2, k3], [dt, -dt, dt], xs)
                    t_dev.fill_(t1)
                    k4 = model_fn(xs, t_dev, **model_kwargs)
                    ops.lincomb(x, [k1, k2, k3, k4], [dt / 8, 3 * dt / 8, 3 * dt / 8, dt / 8], x)
                else:
                    xe = x.clone()
                    ops.axpby(k1, xe, dt, 1.0)
                    t_dev.fill_(t1)
                    k2 = model_fn(xe, t_dev, **model_kwargs)
                    ops.axpby(k1, x, 0.5 * dt, 1.0)
                    ops.axpby(k2, x, 0.5 * dt, 1.0)
                if return_trajectory:
                    traj.append(x.clone())
            return torch.stack(traj, 0) if return_trajectory else x[None]
        return sample


    def sample_sde(self, *, sampling_method="Euler", diffusion_form="SBDM", diffusion_norm=1.0, last_step="Mean",
                   last_step_size=0.04, num_steps=250):
        """transport/transport.py:312-372 + integrators.sde (integrators.py:9-76), Linear path / velocity prediction:
        drift = v + D(t) * score, score = (t v - x) / (1 - t), t = linspace(0, 1 - last_step_size, num_steps).
        Every update is an affine combination of device tensors with host-computed scalar coefficients (t is uniform over
        the batch), i.e. one ln3d_lincomb launch; the Wiener increments are drawn like the reference
        (torch.randn(x.size()) from the global CPU generator, then moved to the device).  Returns the list of states."""
        import math
        if last_step is None:
            last_step_size = 0.0
        t1 = 1.0 if last_step_size == 0 else 1 - last_step_size
        ts = torch.linspace(0.0, t1, num_steps)
        dt = float(ts[1] - ts[0])

        def D(t):
            if diffusion_form == "constant":       # the reference hands a Python float to th.sqrt (integrators.py:37)
                raise TypeError("sqrt(): argument 'input' (position 1) must be Tensor, not float")
            if diffusion_form == "SBDM":
                return diffusion_norm * ((1 - t) ** 2 / t + (1 - t)) if t > 0 else float('inf')
            if diffusion_form in ("sigma", "linear"):
                return diffusion_norm * (1 - t)
            if diffusion_form == "decreasing":
                return 0.25 * (diffusion_norm * math.cos(math.pi * t) + 1) ** 2
            if diffusion_form == "inccreasing-decreasing":
                return diffusion_norm * math.sin(math.pi * t) ** 2
            raise NotImplementedError(f"Diffusion form {diffusion_form} not implemented")

        if sampling_method not in ("Euler", "Heun"):
            raise NotImplementedError("Smapler type not implemented.")
        if last_step not in (None, "Mean", "Tweedie", "Euler"):
            raise NotImplementedError()

        @torch.no_grad()
        def _sample(init, model_fn, **model_kwargs):
            dev = init.device
            x = init.clone().float().contiguous()
            t_dev = torch.empty(x.shape[0], device=dev, dtype=torch.float32)

            def vel(xx, t):
                t_dev.fill_(t)
                return model_fn(xx, t_dev, **model_kwargs).contiguous()

            def drift_coefs(t):                     # sde_drift(x, t) = cv * v + cx * x
                d = D(t)
                return 1.0 + d * t / (1 - t), -d / (1 - t)

            xs = []
            for ti in ts[:-1]:
                t = float(ti)
                w = torch.randn(x.size()).to(dev)
                cw = math.sqrt(2 * D(t)) * math.sqrt(dt)
                if sampling_method == "Euler":
                    cv, cx = drift_coefs(t)
                    v = vel(x, t)
                    xn = torch.empty_like(x)
                    ops.lincomb(None, [x, v, w], [1.0 + cx * dt, cv * dt, cw], xn)
                else:
                    xhat = torch.empty_like(x)
                    ops.lincomb(x, [w], [cw], xhat)
                    cv1, cx1 = drift_coefs(t)
                    v1 = vel(xhat, t)
                    xp = torch.empty_like(x)
                    ops.lincomb(None, [xhat, v1], [1.0 + dt * cx1, dt * cv1], xp)
                    cv2, cx2 = drift_coefs(t + dt)
                    v2 = vel(xp, t + dt)
                    xn = torch.empty_like(x)
                    # xhat + 0.5 dt (K1 + K2), K1 = cv1 v1 + cx1 xhat, K2 = cv2 v2 + cx2 xp
                    ops.lincomb(None, [xhat, v1, v2, xp], [1.0 + 0.5 * dt * cx1, 0.5 * dt * cv1, 0.5 * dt * cv2, 0.5 * dt * cx2], xn)
                x = xn
                xs.append(x)
            xl = xs[-1]
            if last_step is None:
                x = xl
            else:
                v = vel(xl, t1)
                x = torch.empty_like(xl)
                if last_step == "Mean":
                    cv, cx = drift_coefs(t1)
                    ops.lincomb(None, [xl, v], [1.0 + cx * last_step_size, cv * last_step_size], x)
                elif last_step == "Euler":
                    ops.lincomb(xl, [v], [last_step_size], x)
                else:                                   # Tweedie: x / alpha + sigma^2 / alpha * score
                    a, sg = t1, 1 - t1
                    ops.lincomb(None, [xl, v], [1.0 / a - sg * sg / a / (1 - t1), sg * sg / a * t1 / (1 - t1)], x)
            xs.append(x)
            assert len(xs) == num_steps, "Samples does not match the number of steps"
            return xs
        return _sample


# ----------------------------------------------------------------------------- adaptive Dormand-Prince 5(4), torchdiffeq semantics
# Restated from torchdiffeq 0.2.3's published algorithm (the package is absent here: PARITY UNPINNED against it; the call site is
# transport/integrators.py:112-119 `odeint(drift, x, t, method='dopri5', atol, rtol)`, defaults atol 1e-6 / rtol 1e-3 from
# transport/transport.py:377-380):
#   * Dormand-Prince-Shampine tableau with FSAL; error estimate dt * sum_i (b5_i - b4_i) k_i;
#   * error ratio = rms( err / (atol + rtol * max(|y0|, |y1|)) ) over the WHOLE state tensor; accept when <= 1;
#   * next step = dt * min(10, max(0.9 / ratio^(1/5), 0.2)) (lower bound 1 instead of 0.2 after an accepted step; x 10 when ratio = 0);
#   * first step from Hairer's heuristic with scale = atol + rtol |y0| and exponent 1/5;
#   * steps are NOT clipped to the output times: the solver steps past an output time t_i and evaluates the 4th-order
#     interpolant fitted through (y0, y_mid, y1, f0, f1) of the step that contains it - also for the last output t = 1, so the
#     model can be evaluated slightly beyond t = 1, as in the reference.
_DP_C = [0.0, 1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0]
_DP_A = [[], [1 / 5], [3 / 40, 9 / 40], [44 / 45, -56 / 15, 32 / 9],
         [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
         [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
         [35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84]]
_DP_E = [35 / 384 - 1951 / 21600, 0.0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720, -2187 / 6784 + 12231 / 42400,
         11 / 84 - 649 / 6300, -1.0 / 60.0]
_DP_MID = [6025192743 / 30085553152 / 2, 0.0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
           187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2]


def dopri5_next_step(dt, ratio, safety=0.9, ifactor=10.0, dfactor=0.2):
    """torchdiffeq's _optimal_step_size (order 5): the lower clamp is lifted to 1 only for error_ratio < 1 (a step with ratio == 1
    exactly is accepted - `error_ratio <= 1` - but may still shrink by the safety factor)."""
    if ratio == 0:
        return dt * ifactor
    return dt * min(ifactor, max(safety / ratio ** 0.2, 1.0 if ratio < 1.0 else dfactor))


def _dopri5_sampler(num_steps, atol, rtol, max_steps=100000):
    """Device implementation: every stage / error / interpolant is one ln3d_lincomb launch, the weighted error norm one
    ln3d_err_ratio_sq launch; the accept / step-size decision is host-side (one scalar read-back per attempted step).
    Times are Python floats (fp64), the state fp32 - torchdiffeq's own mixed precision."""
    ts = [float(v) for v in torch.linspace(0.0, 1.0, num_steps)]

    @torch.no_grad()
    def sample(x, model_fn, return_trajectory=True, **model_kwargs):
        dev = x.device
        y = x.clone().float().contiguous()
        n = y.numel()
        t_dev = torch.empty(y.shape[0], device=dev, dtype=torch.float32)
        acc = torch.zeros(1, device=dev)

        def f(t, yy):
            t_dev.fill_(t)
            return model_fn(yy, t_dev, **model_kwargs).contiguous()

        def rms(err, y0, y1):
            ops.err_ratio_sq(err, y0, y1, atol, rtol, acc)
            return (float(acc.item()) / n) ** 0.5

        k = [None] * 7
        k[0] = f(ts[0], y)
        # _select_initial_step(func, t0, y0, order = 4, rtol, atol, norm, f0)
        d0, d1 = rms(y, y, None), rms(k[0], y, None)
        h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
        y1 = torch.empty_like(y)
        ops.lincomb(y, [k[0]], [h0], y1)
        f1 = f(ts[0] + h0, y1)
        df = torch.empty_like(y)
        ops.lincomb(None, [f1, k[0]], [1.0, -1.0], df)
        d2 = rms(df, y, None) / h0
        h1 = max(1e-6, h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else (0.01 / max(d1, d2)) ** (1.0 / 5.0)
        dt = min(100 * h0, h1)
        nfe, steps, accepted = 2, 0, 0
        trace = []                                          # (t, dt, error ratio) of every attempted step
        t0 = t1 = ts[0]                                    # the interval the interpolant covers (empty before the first step)
        y0i, f0i = y.clone(), k[0]                          # left end of that interval
        y1b, ymid = torch.empty_like(y), torch.empty_like(y)
        ca, cb, cc = torch.empty_like(y), torch.empty_like(y), torch.empty_like(y)
        err, ystage = torch.empty_like(y), torch.empty_like(y)
        dti = 0.0
        out = [y.clone()]
        for t_out in ts[1:]:
            while t_out > t1:                               # _advance: step until the accepted interval reaches the output time
                if steps >= max_steps:
                    raise RuntimeError("dopri5: max_steps exceeded")
                ta = t1                                     # attempt a step from the end of the last accepted one
                for s_ in range(1, 7):
                    ops.lincomb(y, k[:s_], [dt * a for a in _DP_A[s_]], ystage)
                    if s_ < 6:
                        k[s_] = f(ta + _DP_C[s_] * dt, ystage)
                    else:
                        y1b.copy_(ystage)                   # FSAL: the last stage's input is the 5th-order solution
                        k[6] = f(ta + dt, y1b)
                nfe += 6
                ops.lincomb(None, k, [dt * e for e in _DP_E], err)
                ratio = rms(err, y, y1b)
                steps += 1
                ok = ratio <= 1.0
                trace.append((ta, dt, ratio))
                if ok:
                    accepted += 1
                    # _interp_fit(y0, y1, k, dt)
                    ops.lincomb(y, k, [dt * m for m in _DP_MID], ymid)
                    ops.lincomb(None, [k[6], k[0], y1b, y, ymid], [2 * dt, -2 * dt, -8.0, -8.0, 16.0], ca)
                    ops.lincomb(None, [k[0], k[6], y, y1b, ymid], [5 * dt, -3 * dt, 18.0, 14.0, -32.0], cb)
                    ops.lincomb(None, [k[6], k[0], y, y1b, ymid], [dt, -4 * dt, -11.0, -5.0, 16.0], cc)
                    y0i.copy_(y)
                    f0i, dti = k[0], dt
                    t0, t1 = ta, ta + dt
                    y.copy_(y1b)
                    k[0] = k[6]
                dt = dopri5_next_step(dt, ratio)
            xq = (t_out - t0) / (t1 - t0)
            yo = torch.empty_like(y)
            ops.lincomb(None, [y0i, f0i, cc, cb, ca], [1.0, xq * dti, xq ** 2, xq ** 3, xq ** 4], yo)     # _interp_evaluate
            out.append(yo)
        sample.last_stats = {'nfe': nfe, 'steps': steps, 'accepted': accepted, 't_end': t1, 'h0': min(100 * h0, h1), 'trace': trace}
        return torch.stack(out, 0) if return_trajectory else out[-1][None]
    return sample
