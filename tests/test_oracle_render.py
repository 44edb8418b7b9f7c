"""The render oracle (oracle/render_oracle.py) against golden vectors produced by the
reference itself (tests/golden/gen_golden_render.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import render_oracle as ro
from tests.golden_util import GOLDEN, load_render_case, psnr
from humanliff_amd import synthetic as syn


@pytest.fixture(scope="module")
def units():
    import os
    return np.load(os.path.join(GOLDEN, "render_units.npz"))


def test_plane_features_match_reference(units):
    planes = syn.triplane(seed=11, H=64, W=64)[0]
    pts = torch.from_numpy(units["pts"])
    got = ro.plane_features(planes, pts, torch.tensor(syn.WORLD_BOUNDS))
    ref = torch.from_numpy(units["feats"])
    assert (got - ref).abs().max() < 2e-6
    # points far outside the box sample only zero padding
    far_out = torch.tensor([[5.0, 5.0, 5.0], [-3.0, 0.0, 0.0]])
    assert ro.plane_features(planes, far_out, torch.tensor(syn.WORLD_BOUNDS))[0].abs().max() == 0


def test_view_encoding_matches_reference(units):
    got = ro.view_encoding(torch.from_numpy(units["dirs"]))
    assert (got - torch.from_numpy(units["enc"])).abs().max() < 1e-6


def test_mlp_matches_reference(units):
    p = syn.render_mlp_state(3)
    feats = torch.from_numpy(units["feats"])
    dirs = torch.from_numpy(units["dirs"])
    rgb, sig = ro.mlp(p, feats, dirs)
    assert (rgb - torch.from_numpy(units["rgb_raw"])).abs().max() < 2e-6
    assert (sig - torch.from_numpy(units["sigma"])).abs().max() < 2e-6
    assert (ro.mlp(p, feats) - torch.from_numpy(units["sigma_only"])).abs().max() < 2e-6


def test_importance_sampling_matches_sample_pdf(units):
    """importance_z's inverse-CDF step == reference sample_pdf incl. all-zero-weight rows."""
    bins = torch.from_numpy(units["pdf_bins"])
    w = torch.from_numpy(units["pdf_w"])
    u = torch.from_numpy(units["pdf_u"])
    w2 = w + 1e-5
    pdf = w2 / w2.sum(1, keepdim=True)
    cdf = torch.cat([torch.zeros(16, 1), torch.cumsum(pdf, 1)], 1)
    idx = torch.searchsorted(cdf, u.contiguous(), right=True)
    lo, hi = (idx - 1).clamp(min=0), idx.clamp(max=cdf.shape[1] - 1)
    den = cdf.gather(1, hi) - cdf.gather(1, lo)
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    s = bins.gather(1, lo) + (u - cdf.gather(1, lo)) / den * (bins.gather(1, hi) - bins.gather(1, lo))
    assert (s - torch.from_numpy(units["pdf_samples"])).abs().max() < 1e-6


@pytest.mark.parametrize("name", ["a", "b", "c", "d", "e", "f"])
def test_full_render_matches_reference(name):
    i, e = load_render_case(name)
    rgb, acc, depth, aux = ro.render_rays(i["mlp"], i["planes"][0], i["bounds"], i["rays_o"], i["rays_d"], i["near"],
                                          i["far"], i["n_samples"], i["n_importance"], u=i["u"],
                                          white_bkgd=i["white_bkgd"], return_aux=True)
    assert (aux["sigma_coarse"] - e["sigma_coarse"]).abs().max() < 5e-6
    assert (rgb - e["rgb"]).abs().max() < 5e-6
    assert (acc - e["acc"]).abs().max() < 5e-6
    assert (depth - e["depth"]).abs().max() < 5e-6
    assert psnr(rgb, e["rgb"]) > 100
