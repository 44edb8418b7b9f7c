"""`render` of recon_NeRF/run_nerf_batch.py:29-67 (same argument names and order; no tri_planes argument, `renderer` may be wrapped in
DDP / DataParallel) on top of humanliff_amd.NeRF.render."""
from ..NeRF.renderer import render as _render


def render(chunk=1024 * 32, rays_o=None, rays_d=None, near=0., far=1., tp_input=None, renderer=None, n_samples=128, perturb=0.,
           n_importance=0, white_bkgd=False):
    core = renderer.module if hasattr(renderer, "module") else renderer
    tri_planes = core.tri_planes[tp_input['instance_idx'], tp_input['cloth_layer_index']]
    return _render(chunk=chunk, rays_o=rays_o, rays_d=rays_d, near=near, far=far, tri_planes=tri_planes, tp_input=tp_input,
                   renderer=_Bare(core), n_samples=n_samples, perturb=perturb, n_importance=n_importance, white_bkgd=white_bkgd)


class _Bare:
    """Presents the twin's module to NeRF.render with the human_diffusion twin's render signature (tri_planes passed in)."""

    def __init__(self, core):
        self.core = core

    def __getattr__(self, name):            # (test, uniforms_on_device, cpu_uniforms_on_host ...: the module's own switches)
        core = self.__dict__.get("core")
        if core is None:                    # (copy / pickle look attributes up on an instance whose __init__ has not run: AttributeError, not KeyError)
            raise AttributeError(name)
        return getattr(core, name)

    def render(self, tp_input, world_pts, z_vals, rays_o, rays_d, near, far, tri_planes, n_importance=128, white_bkgd=False, **kw):
        from ..NeRF.renderer import Renderer as _Renderer
        return _Renderer.render(self.core, tp_input, world_pts, z_vals, rays_o, rays_d, near, far, tri_planes, n_importance, white_bkgd, **kw)
