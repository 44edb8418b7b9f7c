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
        """A wrapper per call, like the reference (respace.py:97-122) - but the device tensor of the timestep map it would rebuild every
        time is shared: the cache maps a model object to the per-(device, dtype) map tensors.  The cache holds the model WEAKLY (a model
        carries its parameters plus the packed HIP copy and workspaces - several GB on the device - and deleting it must free them while
        the diffusion object lives on); the wrapper handed to the caller holds it STRONGLY, so a temporary callable (a lambda, a
        functools.partial, a bound method) passed to a sampling loop lives as long as the loop does.  Callables that cannot be weakly
        referenced get a wrapper with private map tensors."""
        if isinstance(model, _WrappedModel):
            return model
        hit = self._wrapped.get(id(model))
        if hit is not None and hit[0]() is model:
            return _WrappedModel(model, self.timestep_map, self.rescale_timesteps, self.original_num_steps, maps=hit[1])
        try:
            key = id(model)
            ref = weakref.ref(model, lambda _, k=key, d=self._wrapped: d.pop(k, None))
        except TypeError:
            return _WrappedModel(model, self.timestep_map, self.rescale_timesteps, self.original_num_steps)
        maps = {}
        self._wrapped[key] = (ref, maps)
        return _WrappedModel(model, self.timestep_map, self.rescale_timesteps, self.original_num_steps, maps=maps)

    def _scale_timesteps(self, t):
        return t  # scaling is done by the wrapped model


class _WrappedModel:
    def __init__(self, model, timestep_map, rescale_timesteps, original_num_steps, maps=None):
        self.model = model                       # strong: the wrapper keeps its model alive (the cache of SpacedDiffusion does not)
        self.timestep_map = timestep_map
        self.rescale_timesteps = rescale_timesteps
        self.original_num_steps = original_num_steps
        self._maps = {} if maps is None else maps   # (device, dtype) -> device tensor of the timestep map, shared between wrappers of one model

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
