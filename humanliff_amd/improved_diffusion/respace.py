"""Timestep respacing with the reference's API (human_diffusion/improved_diffusion/respace.py).

space_timesteps (:7-60) -> kept original-schedule steps; SpacedDiffusion (:63-110) re-derives betas for
the kept steps; _WrappedModel (:113-122) maps loop indices back to ORIGINAL timesteps before the UNet
sees them.  Here the map lives on the device once (the reference rebuilds the tensor every call).
"""
import weakref

import numpy as np
import torch as th

from .gaussian_diffusion import GaussianDiffusion


def space_timesteps(num_timesteps, section_counts):
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[len("ddim"):])
            for stride in range(1, num_timesteps):
                steps = range(0, num_timesteps, stride)
                if len(steps) == want:
                    return set(steps)
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    base, extra = divmod(num_timesteps, len(section_counts))
    kept, start = [], 0
    for i, count in enumerate(section_counts):
        size = base + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        pos = 0.0
        for _ in range(count):
            kept.append(start + round(pos))
            pos += stride
        start += size
    return set(kept)


class SpacedDiffusion(GaussianDiffusion):
    def __init__(self, use_timesteps, **kwargs):
        self.use_timesteps = set(use_timesteps)
        self.original_num_steps = len(kwargs["betas"])
        base_acp = np.cumprod(1.0 - np.array(kwargs["betas"], dtype=np.float64), axis=0)
        self.timestep_map, new_betas, last = [], [], 1.0
        for i, acp in enumerate(base_acp):
            if i in self.use_timesteps:
                new_betas.append(1 - acp / last)
                last = acp
                self.timestep_map.append(i)
        kwargs["betas"] = np.array(new_betas)
        self._wrapped = {}
        super().__init__(**kwargs)

    def p_mean_variance(self, model, *args, **kwargs):
        return super().p_mean_variance(self._wrap_model(model), *args, **kwargs)

    def p_sample(self, model, *args, **kwargs):
        return super().p_sample(self._wrap_model(model), *args, **kwargs)

    def ddim_sample(self, model, *args, **kwargs):
        return super().ddim_sample(self._wrap_model(model), *args, **kwargs)

    def training_losses(self, model, *args, **kwargs):
        return super().training_losses(self._wrap_model(model), *args, **kwargs)

    def _loop_model(self, model):
        return self._wrap_model(model)

    def _wrap_model(self, model):
        """One wrapper per model object (the reference builds a new one - and a new device tensor of the timestep map - on every call,
        respace.py:97-122).  The cache holds the model WEAKLY: a model carries its parameters plus the packed HIP copy and workspaces
        (several GB on the device), and deleting it must free them while the diffusion object lives on.  Callables that cannot be weakly
        referenced (a bare function object can, a bound C callable cannot) get a throwaway wrapper like in the reference."""
        if isinstance(model, _WrappedModel):
            return model
        hit = self._wrapped.get(id(model))
        if hit is not None and hit[0]() is model:
            return hit[1]
        wrapper = _WrappedModel(model, self.timestep_map, self.rescale_timesteps, self.original_num_steps, weak=True)
        if wrapper._ref is None:
            return wrapper
        key = id(model)
        self._wrapped[key] = (weakref.ref(model, lambda _, k=key, d=self._wrapped: d.pop(k, None)), wrapper)
        return wrapper

    def _scale_timesteps(self, t):
        return t  # scaling is done by the wrapped model


class _WrappedModel:
    def __init__(self, model, timestep_map, rescale_timesteps, original_num_steps, weak=False):
        # weak=True (the cached wrappers of SpacedDiffusion): the wrapper must not keep the model alive
        self._ref = self._strong = None
        if weak:
            try:
                self._ref = weakref.ref(model)
            except TypeError:
                self._strong = model
        else:
            self._strong = model
        self.timestep_map = timestep_map
        self.rescale_timesteps = rescale_timesteps
        self.original_num_steps = original_num_steps
        self._maps = {}

    @property
    def model(self):
        return self._strong if self._ref is None else self._ref()

    def parameters(self):
        return self.model.parameters()

    def __call__(self, x, ts, x_cond, **kwargs):
        key = (str(ts.device), ts.dtype)
        m = self._maps.get(key)
        if m is None:
            m = self._maps[key] = th.tensor(self.timestep_map, device=ts.device, dtype=ts.dtype)
        new_ts = m[ts]
        if self.rescale_timesteps:
            new_ts = new_ts.float() * (1000.0 / self.original_num_steps)
        # plain call: a DDP / DataParallel wrapper keeps its forward hooks (gradient sync), any callable works
        return self.model(x, new_ts, x_cond, **kwargs)
