"""Drop-in for recon_NeRF/lib/renderer.py `Renderer` (the tri-plane fitting twin of human_diffusion/NeRF/renderer.py).

Differences of the twin that are mirrored here (recon_NeRF/lib/renderer.py:13-50, 244-295):
  * the tri-planes are a Parameter of the module, (num_instances, 4 cloth layers, 3, triplane_ch/3, triplane_dim, triplane_dim),
    initialised N(0, 0.1), and `render` takes no `tri_planes` argument: it picks
    self.tri_planes[tp_input['instance_idx'], tp_input['cloth_layer_index']];
  * constructor signature without smpl_type.
With test=False (training) the returned rgb_map / acc_map carry gradients for tri_planes and the MLP through the HIP backward
kernels (humanliff_amd/NeRF/train.py), so run_nerf_batch.py's loss.backward() / Adam step work unchanged.
  * depth_map is normalised but NOT clamped to [0,1] (:288; the human_diffusion twin clamps, NeRF/renderer.py:272-274).
Pinned by tests/golden/recon_twin.npz, generated from the reference twin itself (tests/golden/gen_golden_recon.py).
depth_map carries no gradient (the fitting losses of run_nerf_batch.py:250-262 never use it).
"""
import torch
import torch.nn as nn

from ... import _lib
from ...NeRF.renderer import Renderer as _Renderer


class Renderer(_Renderer):
    def __init__(self, use_canonical_space=False, num_instances=1, triplane_dim=256, triplane_ch=18, test=False):
        super().__init__(use_canonical_space=use_canonical_space, num_instances=num_instances, triplane_dim=triplane_dim,
                         triplane_ch=triplane_ch, smpl_type='smpl', test=test)
        self.tri_planes = nn.Parameter(torch.empty(num_instances, 4, 3, triplane_ch // 3, triplane_dim, triplane_dim))
        nn.init.normal_(self.tri_planes, mean=0, std=0.1)
        self._depth_flags = _lib.HL_RENDER_NORMALIZE_DEPTH

    def render(self, tp_input, world_pts, z_vals, rays_o, rays_d, near, far, n_importance=128, white_bkgd=False, **kw):
        tri_planes = self.tri_planes[tp_input['instance_idx'], tp_input['cloth_layer_index']]
        return super().render(tp_input, world_pts, z_vals, rays_o, rays_d, near, far, tri_planes, n_importance, white_bkgd, **kw)

    def _own_planes(self, tp_input):
        return self.tri_planes[tp_input['instance_idx'], tp_input['cloth_layer_index']].detach()

    def density_grid(self, tp_input, resolution=512, **kw):
        return super().density_grid(tp_input, self._own_planes(tp_input)[:1], resolution, **kw)

    def extract_geometry(self, tp_input, resolution, threshold=0.0):
        """recon_NeRF/lib/renderer.py:304 (no tri_planes argument: the module's own, first subject of the batch)."""
        return super().extract_geometry(tp_input, self._own_planes(tp_input)[:1], resolution, threshold)
