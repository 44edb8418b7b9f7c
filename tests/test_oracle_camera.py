"""The camera-ray oracle against the vectors generated from the reference (SynBodyView_datasets.py get_rays /
get_near_far); CPU only."""
import os

import numpy as np
import pytest

from oracle import camera_oracle as co

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "camera_rays.npz")


def cases():
    g = np.load(GOLDEN)
    return [str(n) for n in g["names"]]


@pytest.mark.parametrize("name", cases())
def test_camera_oracle_matches_reference(name):
    g = np.load(GOLDEN)
    H, W = [int(v) for v in g[f"{name}_HW"]]
    ro, rd, near, far, mask = co.camera_rays(H, W, g[f"{name}_K"], g[f"{name}_R"], g[f"{name}_T"], g[f"{name}_bounds"])
    # float64 arithmetic rounded to float32: the only freedom is the summation order inside numpy's 3-term dots,
    # so equality is expected up to one float32 ulp on a handful of elements
    assert np.array_equal(mask, g[f"{name}_mask"])
    for got, want in [(ro, g[f"{name}_rays_o"]), (rd, g[f"{name}_rays_d"]), (near, g[f"{name}_near"]), (far, g[f"{name}_far"])]:
        assert got.dtype == np.float32 and got.shape == want.shape
        ulp = np.spacing(np.abs(want).astype(np.float32))
        assert (np.abs(got.astype(np.float64) - want.astype(np.float64)) <= ulp).all()
        assert (got != want).mean() < 0.01
    assert (rd == np.float32(1e-8)).sum() == (g[f"{name}_rays_d"] == np.float32(1e-8)).sum()
