"""Generate golden vectors for the tri-plane renderer FROM THE REFERENCE.

Runs only in the build container (needs /root/reference, which never travels to
the GPU box).  Inputs are rebuilt from seeds by humanliff_amd.synthetic, so the
fixtures hold only the reference's OUTPUTS plus input checksums.

    python tests/golden/gen_golden_render.py

Reference entry points exercised (unmodified, imported from /root/reference):
    NeRF.renderer.Renderer.render            human_diffusion/NeRF/renderer.py:234
    NeRF.renderer.sample_from_planes         :502
    NeRF.renderer.sample_pdf                 :533
    NeRF.renderer.Renderer.NeRF_network      :134
    NeRF.fields.PositionalEncoding           human_diffusion/NeRF/fields.py:40
The ten lines of scripts/triplane_sample_layered.py:262-279 that build z_vals and
pts cannot be called (hard-coded device='cuda'), so they are restated below.
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
from NeRF.fields import PositionalEncoding  # noqa: E402

torch.autograd.set_detect_anomaly(False)
R.read_pickle = lambda p: {}
R.SMPL_to_tensor = lambda params, device: {"f": None}
torch.cuda.current_device = lambda: 0

from humanliff_amd import synthetic as syn  # noqa: E402


def checksum(t):
    return float(t.double().sum()), float(t.double().abs().sum())


def build_renderer(seed):
    r = R.Renderer(use_canonical_space=False, triplane_dim=256, triplane_ch=27, smpl_type="smpl", test=True)
    sd = syn.render_mlp_state(seed)
    missing = r.load_state_dict(sd, strict=False)
    assert set(missing.missing_keys) <= {"view_enc._freqs", "view_enc._phases"}, missing
    return r, sd


def ref_render(r, planes, bounds, rays_o, rays_d, near, far, n_samples, n_importance, u_seed, white_bkgd):
    """Restates triplane_sample_layered.py:262-279 (perturb=0) then calls Renderer.render."""
    ro = rays_o[None]
    rd = rays_d[None]
    nr = near[None, :, None]
    fr = far[None, :, None]
    t_vals = torch.linspace(0.0, 1.0, steps=n_samples)
    z = nr * (1.0 - t_vals) + fr * t_vals
    pts = ro[..., None, :] + rd[..., None, :] * z[..., :, None]
    pts = pts.reshape(1, -1, 3)
    # sample_pdf draws torch.rand((R, n_importance)) from the global CPU generator;
    # syn.importance_u(R, n_importance, seed) reproduces exactly that draw.
    torch.manual_seed(u_seed)
    caught = {}
    orig_up = r.up_sample

    def spy(densities, z_vals, rays_d_, n_imp):
        caught["sigma_coarse"] = densities.detach().clone()
        out = orig_up(densities, z_vals, rays_d_, n_imp)
        caught["new_z"] = out.detach().clone()
        return out

    r.up_sample = spy
    out = r.render({"world_bounds": bounds[None]}, pts, z, ro, rd, nr, fr, planes, n_importance, white_bkgd)
    r.up_sample = orig_up
    return out, caught


def case(name, plane_hw, img_hw, view, n_views, n_samples, n_importance, white_bkgd, mlp_gain=1.0, ray_slice=None, layer_exp=None):
    planes = syn.triplane(seed=11, H=plane_hw, W=plane_hw)
    bounds = torch.tensor(syn.WORLD_BOUNDS)
    rays_o, rays_d, near, far = syn.orbit_rays(view, n_views, img_hw, img_hw)
    if ray_slice is not None:
        sl = slice(*ray_slice)
        rays_o, rays_d, near, far = rays_o[sl], rays_d[sl], near[sl], far[sl]
    Rn = rays_o.shape[0]
    r, sd = build_renderer(3)
    if mlp_gain != 1.0 or layer_exp:
        sd = syn.render_mlp_state(3, gain=mlp_gain, layer_exp=syn.LAYER_EXP[layer_exp] if layer_exp else None)
        r.load_state_dict(sd, strict=False)
    u = syn.importance_u(Rn, n_importance, seed=5)
    torch.manual_seed(5)
    assert torch.equal(torch.rand((Rn, n_importance)), u)
    out, caught = ref_render(r, planes, bounds, rays_o, rays_d, near, far, n_samples, n_importance, 5, white_bkgd)
    assert torch.equal(out["rgb_map"], out["normal_map"])
    rgb = out["rgb_map"][0]
    if white_bkgd:
        # Reference quirk (renderer.py:222-225): acc_map keeps its last dim, so
        # `rgb_map + (1 - acc_map[..., None])` broadcasts (R,3)+(R,1,1) -> (R,R,3) and the
        # returned map is (R, 3R): entry [i, j] = rgb[j] + (1 - acc[i]).  The per-ray value a
        # caller means is the diagonal i == j; that is what the fixture pins.
        assert rgb.shape == (Rn, 3 * Rn)
        idx = torch.arange(Rn)
        rgb = rgb.reshape(Rn, Rn, 3)[idx, idx]
    np.savez_compressed(
        os.path.join(HERE, f"render_{name}.npz"),
        plane_hw=plane_hw, img_hw=img_hw, view=view, n_views=n_views, n_samples=n_samples,
        n_importance=n_importance, white_bkgd=int(white_bkgd), mlp_gain=mlp_gain, layer_exp=layer_exp or "",
        ray_slice=np.array(ray_slice if ray_slice is not None else [0, Rn]),
        planes_ck=np.array(checksum(planes)), rays_ck=np.array(checksum(rays_d)), u_ck=np.array(checksum(u)),
        w_ck=np.array(checksum(torch.cat([v.flatten() for v in sd.values()]))),
        nearfar_ck=np.array(checksum(torch.stack([near, far]))),
        rgb=rgb.numpy(), acc=out["acc_map"][0].numpy(), depth=out["depth_map"][0].numpy(),
        sigma_coarse=caught["sigma_coarse"][0].numpy(),
        new_z=caught["new_z"][0].numpy(),
    )
    print(name, "rays", Rn, "hit frac", float((far != 1).float().mean()),
          "rgb mean", float(out["rgb_map"].mean()), "acc mean", float(out["acc_map"].mean()))


def units():
    """Per-function vectors: plane sampling, MLP, encoding, sample_pdf."""
    g = torch.Generator().manual_seed(21)
    planes = syn.triplane(seed=11, H=64, W=64)
    bounds = torch.tensor(syn.WORLD_BOUNDS)
    pts = (torch.rand((512, 3), generator=g) * 2 - 1) * torch.tensor([1.15, 1.25, 1.15])  # some outside the box
    # exact texel centres / borders as edge cases
    pts[:8] = torch.tensor([[-1.0, -1.1, -1.0], [1.0, 1.1, 1.0], [0.0, 0.0, 0.0], [1.0, -1.1, 0.5],
                            [-0.984375, 0.0, 0.0], [0.999, 1.099, 0.999], [-1.001, 0.0, 0.0], [0.0, 1.2, 0.0]])
    axes = R.generate_planes()
    feats = R.sample_from_planes(axes, planes, pts[None], padding_mode="zeros", box_warp=bounds[None])
    feats = feats.permute(0, 2, 1, 3).reshape(512, 27)
    r, sd = build_renderer(3)
    dirs = torch.randn((512, 3), generator=g)
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    with torch.no_grad():
        rgb_raw, sigma = r.NeRF_network(feats, canonical_viewdir=dirs[None])
        sigma_only = r.NeRF_network(feats)
        enc = PositionalEncoding(num_freqs=4)(dirs)
    # sample_pdf
    bins = torch.sort(torch.rand((16, 31), generator=g), dim=1)[0]
    w = torch.rand((16, 30), generator=g) ** 4
    w[0] = 0.0  # all-zero weights row
    w[1, 5:] = 0.0
    torch.manual_seed(9)
    samples = R.sample_pdf(bins, w, 24)
    torch.manual_seed(9)
    u = torch.rand((16, 24))
    np.savez_compressed(
        os.path.join(HERE, "render_units.npz"),
        pts=pts.numpy(), feats=feats.numpy(), dirs=dirs.numpy(), rgb_raw=rgb_raw.numpy(),
        sigma=sigma[:, 0].numpy(), sigma_only=sigma_only[:, 0].numpy(), enc=enc.numpy(),
        pdf_bins=bins.numpy(), pdf_w=w.numpy(), pdf_u=u.numpy(), pdf_samples=samples.numpy(),
        planes_ck=np.array(checksum(planes)),
    )
    print("units ok", feats.abs().mean().item())


if __name__ == "__main__":
    units()
    # a: 16x16 view (hits + misses), 32+32 samples (BASELINE config 1 sampling density)
    case("a", plane_hw=64, img_hw=16, view=3, n_views=36, n_samples=32, n_importance=32, white_bkgd=False)
    # b: production sampling density 128+128 on a row band of a 32x32 view, white background
    case("b", plane_hw=64, img_hw=32, view=10, n_views=36, n_samples=128, n_importance=128, white_bkgd=True,
         ray_slice=(448, 544))
    # c: sharper densities (MLP weights x3) so transmittance actually saturates; ragged ray count (R=77)
    case("c", plane_hw=32, img_hw=16, view=20, n_views=36, n_samples=32, n_importance=32, white_bkgd=False,
         mlp_gain=3.0, ray_slice=(90, 167))
    # d, e, f (round 6): the range of the MLP's weights - layers 2^-8 below and 2^4 above nn.Linear's initialisation in three patterns (syn.LAYER_EXP), biases as they are:
    # what pins the scaled fp16 weight planes of the default product mode (k_mlp_scales_h2) to the reference
    case("d", plane_hw=64, img_hw=16, view=5, n_views=36, n_samples=32, n_importance=32, white_bkgd=False, layer_exp="d")
    case("e", plane_hw=64, img_hw=16, view=7, n_views=36, n_samples=32, n_importance=32, white_bkgd=False, layer_exp="e")
    case("f", plane_hw=64, img_hw=16, view=9, n_views=36, n_samples=32, n_importance=32, white_bkgd=False, layer_exp="f")
