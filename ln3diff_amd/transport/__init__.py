"""Flow-matching (SiT transport) ODE sampling on the HIP path.

Surface of the reference's transport package for the sampling path: create_transport(...),
Sampler(transport).sample_ode(sampling_method, num_steps, atol, rtol, reverse)(x, model_fn, **kw) -> [T, ...]
(transport/__init__.py:3-71, transport/transport.py:374-420, transport/integrators.py:78-120).
Linear path + velocity prediction: dx/dt = model(x, t), t from 0 (noise) to 1 (data).  Fixed-grid
'euler' and 'heun' are built; the reference's default adaptive 'dopri5' lives in torchdiffeq (absent,
parity unpinned - SURVEY.md §8c) and is a listed next item.
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
        if sampling_method not in ("euler", "heun"):
            raise NotImplementedError(f"ODE method '{sampling_method}': only fixed-grid euler/heun are built "
                                      "(adaptive dopri5 is a third-party solver absent from the reference tree)")
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
