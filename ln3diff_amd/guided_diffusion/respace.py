"""Timestep respacing (reference guided_diffusion/respace.py:8-136)."""
import numpy as np

from .gaussian_diffusion import GaussianDiffusion


def space_timesteps(num_timesteps, section_counts):
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            desired = int(section_counts[len("ddim"):])
            for i in range(1, num_timesteps):
                if len(range(0, num_timesteps, i)) == desired:
                    return set(range(0, num_timesteps, i))
            raise ValueError(f"cannot create exactly {desired} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per = num_timesteps // len(section_counts)
    extra = num_timesteps % len(section_counts)
    start, steps = 0, []
    for i, cnt in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < cnt:
            raise ValueError(f"cannot divide section of {size} steps into {cnt}")
        frac = 1 if cnt <= 1 else (size - 1) / (cnt - 1)
        cur = 0.0
        for _ in range(cnt):
            steps.append(start + round(cur))
            cur += frac
        start += size
    return set(steps)


class SpacedDiffusion(GaussianDiffusion):
    def __init__(self, use_timesteps, **kwargs):
        self.use_timesteps = set(use_timesteps)
        base = GaussianDiffusion(**kwargs)
        last, new_betas, tmap = 1.0, [], []
        for i, ac in enumerate(base.alphas_cumprod):
            if i in self.use_timesteps:
                new_betas.append(1 - ac / last)
                last = ac
                tmap.append(i)
        orig = len(kwargs["betas"])
        kwargs["betas"] = np.array(new_betas)
        super().__init__(**kwargs)
        self.timestep_map = tmap
        self.original_num_steps = orig

    def _model_t(self, i):
        # _WrappedModel: map to the original index, then divide by the original step count (respace.py:126-131)
        t = float(self.timestep_map[i])
        if self.rescale_timesteps:
            t = t * (1000.0 / self.original_num_steps)
        return t / self.original_num_steps
