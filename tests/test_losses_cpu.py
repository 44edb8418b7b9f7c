"""GaussianDiffusion.training_losses for every LossType of the reference (gaussian_diffusion.py:688-772): MSE / RESCALED_MSE with the hybrid
variational-bound term of learned variances, KL / RESCALED_KL, the three mean types - loss terms and the gradient through them against vectors
generated FROM THE REFERENCE (tests/golden/gen_golden_variants.py -> diffusion_variants.npz).  The loss side is differentiable tensor algebra
around the model call, so it is checked here on the CPU with the generator's stub model."""
import os

import numpy as np
import pytest
import torch

from tests.golden_util import GOLDEN


def _stub(x, t, x_cond, two):
    tt = t.float().view(-1, 1, 1, 1) * 0.001
    e = (0.6 * x + 0.25 * x_cond - tt).clamp(-1.5, 1.5) * 1.3
    if not two:
        return e
    return torch.cat([e, (0.4 * x - 0.3 * x_cond + tt).clamp(-1, 1)], dim=1)


@pytest.mark.parametrize("tag,loss_t,mean_t,var_t,two", [("kl", "KL", "EPSILON", "LEARNED_RANGE", True), ("rkl", "RESCALED_KL", "START_X", "FIXED_LARGE", False),
                                                         ("hyb", "MSE", "EPSILON", "LEARNED_RANGE", True), ("rhyb", "RESCALED_MSE", "EPSILON", "LEARNED", True),
                                                         ("msexp", "MSE", "PREVIOUS_X", "FIXED_SMALL", False), ("klxp", "KL", "PREVIOUS_X", "LEARNED", True)])
def test_training_losses_match_reference(tag, loss_t, mean_t, var_t, two):
    from humanliff_amd.improved_diffusion import gaussian_diffusion as gd
    from humanliff_amd.improved_diffusion.respace import SpacedDiffusion, space_timesteps
    g = np.load(os.path.join(GOLDEN, "diffusion_variants.npz"))
    gen = torch.Generator().manual_seed(7)
    _ = torch.randn((3, 27, 8, 8), generator=gen)
    xc = torch.randn((3, 27, 8, 8), generator=gen) * 0.5
    noise = torch.randn((3, 27, 8, 8), generator=gen)
    x0, t = torch.from_numpy(g["loss_x0"]), torch.from_numpy(g["loss_t"])
    d = SpacedDiffusion(use_timesteps=space_timesteps(1000, "ddim50"), betas=gd.get_named_beta_schedule("linear", 1000),
                        model_mean_type=getattr(gd.ModelMeanType, mean_t), model_var_type=getattr(gd.ModelVarType, var_t),
                        loss_type=getattr(gd.LossType, loss_t), rescale_timesteps=False)
    p = torch.tensor(0.8, requires_grad=True)
    model = lambda a, b, c, **k: _stub(a, b, (xc if c is None else c) * p, two)  # noqa: E731   (the KL losses call the model without x_cond, like the reference)
    terms = d.training_losses(model, x0, xc, t, noise=noise)
    terms["loss"].sum().backward()
    want = {k[len(f"loss_{tag}_"):]: g[k] for k in g.files if k.startswith(f"loss_{tag}_")}
    assert set(terms) == set(want) - {"dp"}
    for k, v in terms.items():
        ref = want[k]
        assert np.abs(v.detach().numpy() - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()), (tag, k)
    assert abs(float(p.grad) - float(want["dp"])) <= 2e-6 * max(1.0, abs(float(want["dp"]))), (tag, float(p.grad), float(want["dp"]))


def test_likelihood_helpers():
    """normal_kl of identical Gaussians is 0 and grows with the mean gap; the discretised likelihood integrates to 1 over the 256 bins."""
    from humanliff_amd.improved_diffusion.losses import discretized_gaussian_log_likelihood, normal_kl
    m = torch.tensor([0.3, -0.2])
    assert float(normal_kl(m, 0.1, m, 0.1).abs().max()) == 0.0
    assert float(normal_kl(m, 0.0, m + 1.0, 0.0)[0]) == pytest.approx(0.5)
    bins = torch.linspace(-1, 1, 256)
    lp = discretized_gaussian_log_likelihood(bins, means=torch.full_like(bins, 0.1), log_scales=torch.full_like(bins, -2.0))
    assert float(lp.exp().sum()) == pytest.approx(1.0, abs=2e-3)      # (tanh approximation of the CDF)
