"""Small helpers with the reference's names (human_diffusion/improved_diffusion/nn.py).

The modules below are PARAMETER HOLDERS: they give the UNet the reference's state_dict layout
(nn.py:17-39, 93-100).  Their math runs in the HIP kernels, never through torch.nn.functional.
"""
import torch as th
import torch.nn as nn


class SiLU(nn.Module):
    """Placeholder keeping the Sequential indices of the reference (nn.py:12-14)."""


class GroupNorm32(nn.GroupNorm):
    """GroupNorm(32, C), fp32 statistics (nn.py:17-19); evaluated by k_gn_partial/k_gn_coef."""


def normalization(channels):
    return GroupNorm32(32, channels)


def conv_nd(dims, *args, **kwargs):
    if dims == 1:
        return nn.Conv1d(*args, **kwargs)
    if dims == 2:
        return nn.Conv2d(*args, **kwargs)
    raise ValueError(f"unsupported dimensions: {dims} (the MI355X build covers the 2-D UNet)")


def linear(*args, **kwargs):
    return nn.Linear(*args, **kwargs)


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


def mean_flat(tensor):
    return tensor.mean(dim=list(range(1, tensor.dim())))


def update_ema(target_params, source_params, rate=0.99):
    for targ, src in zip(target_params, source_params):
        targ.detach().mul_(rate).add_(src, alpha=1 - rate)


def timestep_embedding(timesteps, dim, max_period=10000):
    """Sinusoidal embedding (nn.py:103-121), computed by the HIP kernel the UNet itself uses."""
    from .. import _lib
    assert dim % 2 == 0 and max_period == 10000, "only even widths / max_period=10000 are built"
    tf = ti = None
    if timesteps.is_floating_point():
        tf = timesteps.to(th.float32).contiguous()
    else:
        ti = timesteps.to(th.int64).contiguous()
    out = th.empty((timesteps.shape[0], dim), dtype=th.float32, device=timesteps.device)
    with _lib.on(timesteps.device):
        _lib.check(_lib.lib().hl_timestep_embedding(_lib.ptr(ti), _lib.ptr(tf), timesteps.shape[0], dim, _lib.ptr(out),
                                                    _lib.stream_ptr()), "hl_timestep_embedding")
    return out
