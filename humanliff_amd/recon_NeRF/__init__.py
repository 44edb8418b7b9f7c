"""Mirror of the reference's tri-plane fitting package recon_NeRF/ (the hot path only: lib/renderer.py and the `render` of
run_nerf_batch.py) on the HIP kernels."""
from .lib.renderer import Renderer  # noqa: F401
from .run_nerf_batch import render  # noqa: F401
