"""Golden vectors FROM THE REFERENCE's recon_NeRF twin - the tri-plane fitting renderer BASELINE.json's north_star names
(recon_NeRF/run_nerf_batch.py `render` -> recon_NeRF/lib/renderer.py `Renderer.render`).

Runs only in the build container (imports /root/reference/recon_NeRF/lib/renderer.py unmodified).

    python tests/golden/gen_golden_recon.py

What differs from the human_diffusion twin and is pinned here (recon_NeRF/lib/renderer.py):
  :26-27   tri_planes is a Parameter of the module, (num_instances, 4 cloth layers, 3, C/3, H, W)
  :246-251 render() has no tri_planes argument: tri_planes = self.tri_planes[tp_input['instance_idx'], tp_input['cloth_layer_index']]
           (tensor indices -> a freshly gathered (bs,3,C/3,H,W) tensor per call; its gradient scatters back into the Parameter)
  :288     depth_map = (depth - near) / (far - near + 1e-5) is NOT clamped to [0,1] (the human_diffusion twin clamps, :272-274)
Cases:
  test     test=True, batch of 2 subjects (instance, layer) = (1,2), (0,3); 96 rays each, 32+32 samples; the `near` / `far` ARGUMENTS
           of the second subject are deliberately offset from the depth range of its z_vals, so the normalised depth leaves [0,1]
           and the missing clamp is visible.
  train    test=False, same gather, stratified depths, randn_like density noise, L = sum(rgb*G_rgb) + sum(acc*G_acc) -> gradient
           of the tri_planes Parameter (non-zero only at the two gathered slots) and of the MLP.
`render` of run_nerf_batch.py:29-67 itself cannot be imported (argparse at import, hard-coded device='cuda'); its z_vals / pts lines
(:45-57) are restated like in gen_golden_render.py.
"""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, "/root/reference/recon_NeRF")

for n in ["mcubes", "cv2", "pytorch3d", "pytorch3d.ops", "pytorch3d.ops.knn"]:
    sys.modules[n] = types.ModuleType(n)
sys.modules["pytorch3d.ops.knn"].knn_points = None

from lib import renderer as R  # noqa: E402

torch.autograd.set_detect_anomaly(False)
R.read_pickle = lambda p: {}
R.SMPL_to_tensor = lambda params, device: {"f": None}
torch.cuda.current_device = lambda: 0

from humanliff_amd import synthetic as syn  # noqa: E402

HW, NI, NS = 32, 2, 32
MLP_KEYS = [f"{m}.{k}" for m in ("pts_linears.0", "pts_linears.1", "pts_linears.2", "feature_linear", "alpha_linear", "views_linear",
                                 "rgb_linear") for k in ("weight", "bias")]


def module_planes(seed=17):
    """The module's Parameter, seeded: clamp(0.3 * randn, -1, 1) like synthetic.triplane (the reference initialises N(0, 0.1) and
    clamps to [-1,1] after every step, run_nerf_batch.py:271-272)."""
    g = torch.Generator().manual_seed(seed)
    return (torch.randn((NI, 4, 3, 9, HW, HW), generator=g) * 0.3).clamp_(-1, 1)


def build(test):
    r = R.Renderer(use_canonical_space=False, num_instances=NI, triplane_dim=HW, triplane_ch=27, test=test)
    r.load_state_dict(syn.render_mlp_state(3), strict=False)
    with torch.no_grad():
        r.tri_planes.copy_(module_planes())
    return r


def rays(n_rays, seed):
    g = torch.Generator().manual_seed(seed)
    out = []
    for view in (3, 5):
        ro, rd, nr, fr = syn.orbit_rays(view, 8, 24, 24)
        hit = torch.nonzero(fr != 1).flatten()
        miss = torch.nonzero(fr == 1).flatten()
        pick = torch.cat([hit[torch.randperm(hit.numel(), generator=g)[:n_rays - 8]], miss[torch.randperm(miss.numel(), generator=g)[:8]]])
        out.append([t[pick] for t in (ro, rd, nr, fr)])
    return [torch.stack([a[i] for a in out]) for i in range(4)]      # (2, R, .)


def case_test():
    r = build(True)
    ro, rd, nr, fr = rays(96, 3)
    t = torch.linspace(0., 1., steps=NS)
    z = nr[..., None] * (1. - t) + fr[..., None] * t                  # run_nerf_batch.py:45-46
    pts = (ro[..., None, :] + rd[..., None, :] * z[..., :, None]).reshape(2, -1, 3)
    # the near / far ARGUMENTS of subject 1: shifted by 30 % of the span -> normalised depths < 0 and > 1
    span = fr[1] - nr[1]
    near_arg, far_arg = nr.clone(), fr.clone()
    near_arg[1] = nr[1] + 0.3 * span
    far_arg[1] = fr[1] - 0.3 * span
    tp = {"world_bounds": torch.tensor(syn.WORLD_BOUNDS)[None].expand(2, 2, 3), "instance_idx": torch.tensor([1, 0]),
          "cloth_layer_index": torch.tensor([2, 3])}
    torch.manual_seed(5)
    u = torch.rand((2 * 96, NS))
    torch.manual_seed(5)
    with torch.no_grad():
        out = r.render(tp, pts, z, ro, rd, near_arg[..., None], far_arg[..., None], NS, False)
    d = out["depth_map"]
    print("test: depth range", float(d.min()), float(d.max()), "outside [0,1]:", int(((d < 0) | (d > 1)).sum()), "rgb mean", float(out["rgb_map"].mean()))
    assert ((d < 0) | (d > 1)).any()
    return dict(t_rays_o=ro.numpy(), t_rays_d=rd.numpy(), t_near=nr.numpy(), t_far=fr.numpy(), t_near_arg=near_arg.numpy(),
                t_far_arg=far_arg.numpy(), t_u=u.numpy(), t_rgb=out["rgb_map"].numpy(), t_acc=out["acc_map"].numpy(),
                t_depth=d.numpy())


def case_train():
    r = build(False)
    g = torch.Generator().manual_seed(11)
    ro, rd, nr, fr = rays(40, 4)
    n_rays = 40
    t = torch.linspace(0., 1., steps=16)
    z = nr[..., None] * (1. - t) + fr[..., None] * t
    mids = .5 * (z[..., 1:] + z[..., :-1])                             # run_nerf_batch.py:47-55 (perturb = 1)
    upper = torch.cat([mids, z[..., -1:]], -1)
    lower = torch.cat([z[..., :1], mids], -1)
    z = lower + (upper - lower) * torch.rand(z.shape, generator=g)
    pts = (ro[..., None, :] + rd[..., None, :] * z[..., :, None]).reshape(2, -1, 3)
    G_rgb = torch.randn((2, n_rays, 3), generator=g)
    G_acc = torch.randn((2, n_rays), generator=g)
    tp = {"world_bounds": torch.tensor(syn.WORLD_BOUNDS)[None].expand(2, 2, 3), "instance_idx": torch.tensor([1, 0]),
          "cloth_layer_index": torch.tensor([2, 3])}
    S = 32
    torch.manual_seed(9)
    u = torch.rand((2 * n_rays, 16))
    noise = torch.randn((2 * n_rays * S, 1))
    torch.manual_seed(9)
    out = r.render(tp, pts, z, ro, rd, nr[..., None], fr[..., None], 16, False)
    loss = (out["rgb_map"] * G_rgb).sum() + (out["acc_map"] * G_acc).sum()
    loss.backward()
    gp = r.tri_planes.grad
    touched = torch.nonzero(gp.abs().sum(dim=(2, 3, 4, 5)))
    print("train: loss", float(loss), "grad slots", touched.tolist(), "|grad|", float(gp.abs().sum()))
    assert touched.tolist() == [[0, 3], [1, 2]]
    sd = dict(r.named_parameters())
    return dict(g_rays_o=ro.numpy(), g_rays_d=rd.numpy(), g_near=nr.numpy(), g_far=fr.numpy(), g_z=z.numpy(), g_u=u.numpy(),
                g_noise=noise.reshape(2, n_rays, S).numpy(), g_G_rgb=G_rgb.numpy(), g_G_acc=G_acc.numpy(),
                g_rgb=out["rgb_map"].detach().numpy(), g_acc=out["acc_map"].detach().numpy(), g_depth=out["depth_map"].detach().numpy(),
                g_d_planes_1_2=gp[1, 2].numpy(), g_d_planes_0_3=gp[0, 3].numpy(),
                **{"g_d_" + k: sd[k].grad.numpy() for k in MLP_KEYS})


if __name__ == "__main__":
    res = {"hw": HW, "num_instances": NI, "n_samples": NS, "planes_ck": np.array([float(module_planes().double().sum()),
                                                                                 float(module_planes().double().abs().sum())])}
    res.update(case_test())
    res.update(case_train())
    np.savez_compressed(os.path.join(HERE, "recon_twin.npz"), **res)
    print("ok", os.path.getsize(os.path.join(HERE, "recon_twin.npz")) // 1024, "KiB")
