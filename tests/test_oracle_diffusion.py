"""UNet + sampler oracles against golden vectors produced by the reference.  CPU only."""
import ast
import os

import numpy as np
import pytest
import torch

from humanliff_amd import synthetic as syn
from oracle import diffusion_oracle as do
from oracle import unet_oracle as uo
from tests.golden_util import GOLDEN


def load_unet_case(name):
    g = np.load(os.path.join(GOLDEN, f"unet_{name}.npz"))
    ks = [(str(k), tuple(ast.literal_eval(str(s)))) for k, s in zip(g["keys"], g["shapes"])]
    sd = syn.state_from_shapes(ks, seed=1)
    B, size = int(g["B"]), int(g["arg_image_size"])
    gen = torch.Generator().manual_seed(7)
    x = torch.randn((B, 27, size, size), generator=gen)
    xc = torch.randn((B, 27, size, size), generator=gen).clamp(-1, 1) * 0.7
    assert np.allclose([float(x.double().sum()), float(x.double().abs().sum())], g["x_ck"], atol=1e-6, rtol=0)
    return g, ks, sd, x, xc, torch.from_numpy(g["t"]).long(), torch.from_numpy(g["y"]).long()


@pytest.mark.parametrize("name", ["tiny32", "mid64", "deep256"])
def test_unet_oracle_matches_reference(name):
    g, ks, sd, x, xc, t, y = load_unet_case(name)
    with torch.no_grad():
        out = uo.unet_forward(sd, x, t, xc, y, num_heads=int(g["arg_num_heads"]))
    s = int(g["stride"])
    assert (out[:, :, ::s, ::s] - torch.from_numpy(g["out"])).abs().max() < 2e-5   # outputs are O(0.5)
    assert abs(float(out.double().abs().sum()) - g["out_ck"][1]) / g["out_ck"][1] < 1e-6


def test_timestep_embedding():
    g = np.load(os.path.join(GOLDEN, "diffusion_steps.npz"))
    got = uo.timestep_embedding(torch.tensor([0, 1, 500, 999]), 192)
    assert (got - torch.from_numpy(g["temb192"])).abs().max() < 1e-6


@pytest.mark.parametrize("tag,spec", [("full", [1000]), ("r250", "250"), ("ddim50", "ddim50"), ("ddim10", "ddim10"),
                                      ("mix", "10,15,20")])
def test_schedules(tag, spec):
    g = np.load(os.path.join(GOLDEN, "diffusion_steps.npz"))
    s = do.Schedule(do.linear_betas(1000), do.kept_timesteps(1000, spec))
    assert s.timestep_map == list(g[f"{tag}_map"])
    for mine, key in [(s.betas, "betas"), (s.post_logvar, "post_logvar"), (s.coef1, "coef1"), (s.coef2, "coef2"),
                      (s.sqrt_recip, "sqrt_recip"), (s.sqrt_recipm1, "sqrt_recipm1")]:
        assert np.array_equal(mine, g[f"{tag}_{key}"]), key       # float64 tables: bit-exact


def test_cosine_and_errors():
    g = np.load(os.path.join(GOLDEN, "diffusion_steps.npz"))
    assert np.array_equal(do.cosine_betas(50), g["cosine50_betas"])
    with pytest.raises(ValueError):
        do.kept_timesteps(1000, "ddim999")
    with pytest.raises(ValueError):
        do.kept_timesteps(10, "20")


def _stub(x, t_orig, xc, y):
    tt = t_orig.float().view(-1, 1, 1, 1) * 0.001
    return (0.6 * x + 0.25 * xc - tt + y.float().view(-1, 1, 1, 1) * 0.05).clamp(-1.5, 1.5) * 1.3


@pytest.mark.parametrize("tag,spec", [("full", [1000]), ("ddim50", "ddim50"), ("r250", "250")])
@pytest.mark.parametrize("clip", [True, False])
def test_single_steps(tag, spec, clip):
    g = np.load(os.path.join(GOLDEN, "diffusion_steps.npz"))
    gen = torch.Generator().manual_seed(7)
    x = torch.randn((3, 27, 8, 8), generator=gen)
    xc = torch.randn((3, 27, 8, 8), generator=gen) * 0.5
    noise = torch.randn((3, 27, 8, 8), generator=gen)
    y = torch.tensor([0, 3, 1])
    s = do.Schedule(do.linear_betas(1000), do.kept_timesteps(1000, spec))
    c = int(clip)
    t = torch.from_numpy(g[f"step_{tag}_{c}_t"]).long()
    eps = _stub(x, torch.tensor(s.timestep_map)[t], xc, y)
    ps, x0 = do.p_sample_step(s, x, t, eps, noise, clip)
    # same fp32 op order as the reference; bit-equal on the build host, within 2 ulp on hosts whose
    # CPU kernels round the per-timestep scalars differently (observed between x86 ISAs)
    def near(a, b):
        return ((a - b).abs() <= 3e-7 * b.abs() + 2e-7).all()
    assert near(ps, torch.from_numpy(g[f"step_{tag}_{c}_p_sample"]))
    assert near(x0, torch.from_numpy(g[f"step_{tag}_{c}_p_x0"]))
    dd, _ = do.ddim_step(s, x, t, eps, noise, clip, 0.0)
    assert near(dd, torch.from_numpy(g[f"step_{tag}_{c}_ddim_sample"]))
    de, _ = do.ddim_step(s, x, t, eps, noise, clip, 0.7)
    assert near(de, torch.from_numpy(g[f"step_{tag}_{c}_ddim_eta_sample"]))


@pytest.mark.parametrize("tag,spec,ddim", [("ddim10", "ddim10", True), ("p8", "8", False)])
def test_full_loops(tag, spec, ddim):
    g = np.load(os.path.join(GOLDEN, "diffusion_loops.npz"))
    _, ks, sd, _, xc, _, _ = load_unet_case("tiny32")
    s = do.Schedule(do.linear_betas(1000), do.kept_timesteps(1000, spec))
    y = torch.tensor([1, 2])
    n = {"i": 0}

    def draw(shape):
        gg = torch.Generator().manual_seed(7000 + n["i"])
        n["i"] += 1
        return torch.randn(tuple(shape), generator=gg)

    x_T = draw((2, 27, 32, 32))
    with torch.no_grad():
        out = do.sample_loop(s, lambda x, t: uo.unet_forward(sd, x, t, xc, y), x_T, draw, ddim)
    assert n["i"] == int(g[f"{tag}_ndraws"])
    assert (out - torch.from_numpy(g[f"{tag}_sample"])).abs().max() < 5e-5


def test_ddim50_loop_drift_against_reference():
    """50 recurrent steps (BASELINE configs[3]'s schedule) on the tiny net: the oracle stays within fp32 noise of the reference's
    loop (tests/golden/gen_golden_drift.py) - the mid-loop state (not yet clamped by the last step) as well as the final sample."""
    g = np.load(os.path.join(GOLDEN, "diffusion_drift.npz"))
    _, ks, sd, _, xc, _, _ = load_unet_case("tiny32")
    s = do.Schedule(do.linear_betas(1000), do.kept_timesteps(1000, "ddim50"))
    y = torch.tensor([1, 2])
    n = {"i": 0}

    def draw(shape):
        gg = torch.Generator().manual_seed(7000 + n["i"])
        n["i"] += 1
        return torch.randn(tuple(shape), generator=gg)

    x = draw((2, 27, 32, 32))
    tmap = torch.tensor(s.timestep_map, dtype=torch.int64)
    half = None
    with torch.no_grad():
        for k, i in enumerate(reversed(range(s.T))):
            t = torch.full((2,), i, dtype=torch.int64)
            x = do.ddim_step(s, x, t, uo.unet_forward(sd, x, tmap[t], xc, y), draw(x.shape))[0]
            if k == s.T // 2 - 1:
                half = x.clone()
    assert n["i"] == int(g["tiny32_ddim50_ndraws"])
    assert (half - torch.from_numpy(g["tiny32_ddim50_half"])).abs().max() < 1e-4
    assert (x - torch.from_numpy(g["tiny32_ddim50_sample"])).abs().max() < 1e-4
