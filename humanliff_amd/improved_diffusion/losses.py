"""Likelihood helpers of the variational-bound training terms (the reference's improved_diffusion/losses.py: normal_kl :12-41,
approx_standard_normal_cdf :44-49, discretized_gaussian_log_likelihood :52-77).  Plain differentiable tensor algebra on the tensors'
device - the loss side of GaussianDiffusion.training_losses, not a kernel path."""
import math

import torch as th


def normal_kl(mean1, logvar1, mean2, logvar2):
    """KL( N(mean1, exp(logvar1)) || N(mean2, exp(logvar2)) ), elementwise; scalars broadcast against the tensor arguments."""
    ref = next((v for v in (mean1, logvar1, mean2, logvar2) if isinstance(v, th.Tensor)), None)
    assert ref is not None, "at least one argument must be a Tensor"
    lv1 = logvar1 if isinstance(logvar1, th.Tensor) else th.tensor(logvar1).to(ref)
    lv2 = logvar2 if isinstance(logvar2, th.Tensor) else th.tensor(logvar2).to(ref)
    return 0.5 * (-1.0 + lv2 - lv1 + th.exp(lv1 - lv2) + ((mean1 - mean2) ** 2) * th.exp(-lv2))


def approx_standard_normal_cdf(x):
    """tanh approximation of the standard normal CDF (the reference's, :44-49)."""
    return 0.5 * (1.0 + th.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * th.pow(x, 3))))


def discretized_gaussian_log_likelihood(x, *, means, log_scales):
    """log-probability (nats) of x under a Gaussian discretised to the 256 bins of an image rescaled to [-1, 1]: the CDF mass of the bin
    x falls in, open-ended at both borders (x < -0.999: everything below the upper edge; x > 0.999: everything above the lower edge)."""
    assert x.shape == means.shape == log_scales.shape
    d = x - means
    inv_std = th.exp(-log_scales)
    cdf_hi = approx_standard_normal_cdf(inv_std * (d + 1.0 / 255.0))
    cdf_lo = approx_standard_normal_cdf(inv_std * (d - 1.0 / 255.0))
    log_below = th.log(cdf_hi.clamp(min=1e-12))
    log_above = th.log((1.0 - cdf_lo).clamp(min=1e-12))
    log_bin = th.log((cdf_hi - cdf_lo).clamp(min=1e-12))
    out = th.where(x < -0.999, log_below, th.where(x > 0.999, log_above, log_bin))
    assert out.shape == x.shape
    return out
