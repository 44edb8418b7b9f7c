"""Generate golden GRADIENT vectors for the renderer's training mode FROM THE REFERENCE.

Runs only in the build container (needs /root/reference).

    python tests/golden/gen_golden_render_grad.py

Reference code exercised (unmodified, imported from /root/reference): NeRF.renderer.Renderer.render with test=False
(human_diffusion/NeRF/renderer.py:234-281 -> up_sample under no_grad :243-253, render_core with the randn_like density noise :212)
followed by torch autograd of   L = sum(rgb_map * G_rgb) + sum(acc_map * G_acc)   - the structure of the fitting loss of
recon_NeRF/run_nerf_batch.py:250-252 (MSE on rgb_map and acc_map) with fixed cotangents.  Stratified depths restate
recon_NeRF/run_nerf_batch.py:49-56 (hard-coded device='cuda' there).

Random draws inside the reference, in call order after torch.manual_seed(seed): sample_pdf's torch.rand((R, n_importance)), then
render_core's torch.randn_like(alpha) with alpha (R*(n_samples+n_importance), 1); both are re-drawn here and stored as inputs.
"""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, "/root/reference/human_diffusion")

for n in ["mcubes", "cv2", "pytorch3d", "pytorch3d.ops", "pytorch3d.ops.knn"]:
    sys.modules[n] = types.ModuleType(n)
sys.modules["pytorch3d.ops.knn"].knn_points = None

from NeRF import renderer as R  # noqa: E402

R.read_pickle = lambda p: {}
R.SMPL_to_tensor = lambda params, device: {"f": None}
torch.cuda.current_device = lambda: 0

from humanliff_amd import synthetic as syn  # noqa: E402

MLP_KEYS = [f"{m}.{k}" for m in ("pts_linears.0", "pts_linears.1", "pts_linears.2", "feature_linear", "alpha_linear", "views_linear",
                                 "rgb_linear") for k in ("weight", "bias")]


def case(name, plane_hw, n_rays, n_samples, white_bkgd, seed):
    g = torch.Generator().manual_seed(seed)
    planes = syn.triplane(seed=11, H=plane_hw, W=plane_hw).clone().requires_grad_(True)      # (1,3,9,H,W)
    bounds = torch.tensor(syn.WORLD_BOUNDS)
    ro, rd, nr, fr = syn.orbit_rays(3, 8, 24, 24)
    hit = torch.nonzero(fr != 1).flatten()
    pick = hit[torch.randperm(hit.numel(), generator=g)[:n_rays]]
    ro, rd, nr, fr = ro[pick], rd[pick], nr[pick], fr[pick]
    # stratified depths (perturb = 1)
    t = torch.linspace(0., 1., steps=n_samples)
    z = nr[:, None] * (1. - t) + fr[:, None] * t
    mids = .5 * (z[:, 1:] + z[:, :-1])
    upper = torch.cat([mids, z[:, -1:]], -1)
    lower = torch.cat([z[:, :1], mids], -1)
    z = lower + (upper - lower) * torch.rand(z.shape, generator=g)
    G_rgb = torch.randn((n_rays, 3), generator=g)
    G_acc = torch.randn((n_rays,), generator=g)

    r = R.Renderer(use_canonical_space=False, triplane_dim=plane_hw, triplane_ch=27, smpl_type="smpl", test=False)
    r.load_state_dict(syn.render_mlp_state(3), strict=False)
    S = 2 * n_samples
    torch.manual_seed(seed)
    u = torch.rand((n_rays, n_samples))
    noise = torch.randn((n_rays * S, 1))
    torch.manual_seed(seed)
    pts = (ro[:, None, :] + rd[:, None, :] * z[:, :, None]).reshape(1, -1, 3)
    out = r.render({"world_bounds": bounds[None]}, pts, z[None], ro[None], rd[None], nr[None, :, None], fr[None, :, None], planes,
                   n_samples, white_bkgd)
    rgb, acc = out["rgb_map"][0], out["acc_map"][0]
    if white_bkgd:   # reference quirk (see gen_golden_render.py): (R, 3R) broadcast; the per-ray value is the diagonal
        idx = torch.arange(n_rays)
        rgb = rgb.reshape(n_rays, n_rays, 3)[idx, idx]
    loss = (rgb * G_rgb).sum() + (acc * G_acc).sum()
    loss.backward()
    sd = dict(r.named_parameters())
    np.savez_compressed(
        os.path.join(HERE, f"render_grad_{name}.npz"),
        plane_hw=plane_hw, n_samples=n_samples, white_bkgd=int(white_bkgd),
        rays_o=ro.numpy(), rays_d=rd.numpy(), near=nr.numpy(), far=fr.numpy(), z=z.numpy(), u=u.numpy(),
        noise=noise.reshape(n_rays, S).numpy(), G_rgb=G_rgb.numpy(), G_acc=G_acc.numpy(),
        rgb=rgb.detach().numpy(), acc=acc.detach().numpy(), d_planes=planes.grad[0].numpy(),
        **{"d_" + k: sd[k].grad.numpy() for k in MLP_KEYS})
    print(name, "loss", float(loss), "|d_planes|", float(planes.grad.abs().sum()), "acc mean", float(acc.mean()),
          {k: float(sd[k].grad.abs().max()) for k in MLP_KEYS[:4]})


if __name__ == "__main__":
    case("a", 32, 80, 16, False, 7)        # 80 rays: two full tiles and a ragged one
    case("white", 32, 40, 12, True, 9)
