"""ORACLE (test infrastructure, not product code) — CPU restatement of the sampler arithmetic.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Pinned against tests/golden/diffusion_steps.npz and diffusion_loops.npz (reference outputs).

    linear_betas / cosine_betas   gaussian_diffusion.py:18-62
    kept_timesteps                respace.py:7-60
    Schedule                      gaussian_diffusion.py:118-169 + respace.py:72-86
    p_sample_step / ddim_step     gaussian_diffusion.py:232-333, 356-388, 484-529
Tables are float64 numpy exactly like the reference; per-step coefficients are cast to fp32
(gaussian_diffusion.py:850-863) before touching the tensors.
"""
import math

import numpy as np
import torch


def linear_betas(T):
    s = 1000 / T
    return np.linspace(s * 0.0001, s * 0.02, T, dtype=np.float64)


def cosine_betas(T, max_beta=0.999):
    f = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2  # noqa: E731
    return np.array([min(1 - f((i + 1) / T) / f(i / T), max_beta) for i in range(T)])


def kept_timesteps(T, spec):
    if isinstance(spec, str):
        if spec.startswith("ddim"):
            want = int(spec[4:])
            for stride in range(1, T):
                if len(range(0, T, stride)) == want:
                    return sorted(range(0, T, stride))
            raise ValueError("no integer stride")
        spec = [int(v) for v in spec.split(",")]
    per, extra = divmod(T, len(spec))
    start, keep = 0, []
    for i, n in enumerate(spec):
        size = per + (1 if i < extra else 0)
        if size < n:
            raise ValueError("section too small")
        stride = 1 if n <= 1 else (size - 1) / (n - 1)
        cur = 0.0
        for _ in range(n):
            keep.append(start + round(cur))
            cur += stride
        start += size
    return sorted(set(keep))


class Schedule:
    def __init__(self, base_betas, keep):
        acp_base = np.cumprod(1.0 - np.asarray(base_betas, dtype=np.float64))
        last, betas, self.timestep_map = 1.0, [], []
        keep = set(keep)
        for i, a in enumerate(acp_base):
            if i in keep:
                betas.append(1 - a / last)
                last = a
                self.timestep_map.append(i)
        b = self.betas = np.array(betas, dtype=np.float64)
        acp = self.acp = np.cumprod(1.0 - b)
        prev = self.acp_prev = np.append(1.0, acp[:-1])
        self.sqrt_recip = np.sqrt(1.0 / acp)
        self.sqrt_recipm1 = np.sqrt(1.0 / acp - 1)
        self.post_var = b * (1.0 - prev) / (1.0 - acp)
        self.post_logvar = np.log(np.append(self.post_var[1], self.post_var[1:]))
        self.coef1 = b * np.sqrt(prev) / (1.0 - acp)
        self.coef2 = (1.0 - prev) * np.sqrt(1.0 - b) / (1.0 - acp)
        self.fixed_large_var = np.append(self.post_var[1], b[1:])
        self.T = len(b)


def _f(arr, t):
    return torch.from_numpy(arr)[t].float()[:, None, None, None]


def predict_x0(s, x, t, eps, clip):
    x0 = _f(s.sqrt_recip, t) * x - _f(s.sqrt_recipm1, t) * eps
    return x0.clamp(-1, 1) if clip else x0


def p_sample_step(s, x, t, eps, noise, clip=True):
    x0 = predict_x0(s, x, t, eps, clip)
    mean = _f(s.coef1, t) * x0 + _f(s.coef2, t) * x
    logvar = _f(np.log(s.fixed_large_var), t)
    mask = (t != 0).float()[:, None, None, None]
    return mean + mask * torch.exp(0.5 * logvar) * noise, x0


def ddim_step(s, x, t, eps, noise, clip=True, eta=0.0):
    x0 = predict_x0(s, x, t, eps, clip)
    eps2 = (_f(s.sqrt_recip, t) * x - x0) / _f(s.sqrt_recipm1, t)
    ab, abp = _f(s.acp, t), _f(s.acp_prev, t)
    sigma = eta * torch.sqrt((1 - abp) / (1 - ab)) * torch.sqrt(1 - ab / abp)
    mean = x0 * torch.sqrt(abp) + torch.sqrt(1 - abp - sigma ** 2) * eps2
    mask = (t != 0).float()[:, None, None, None]
    return mean + mask * sigma * noise, x0


def sample_loop(s, model_fn, x_T, noise_fn, ddim, clip=True, eta=0.0):
    """model_fn(x, t_original_index (B,) int64) -> eps.  noise_fn(shape) is called once per step."""
    x = x_T
    B = x.shape[0]
    tmap = torch.tensor(s.timestep_map, dtype=torch.int64)
    for i in reversed(range(s.T)):
        t = torch.full((B,), i, dtype=torch.int64)
        eps = model_fn(x, tmap[t])
        n = noise_fn(x.shape)
        x = (ddim_step(s, x, t, eps, n, clip, eta) if ddim else p_sample_step(s, x, t, eps, n, clip))[0]
    return x
