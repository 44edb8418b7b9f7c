"""Golden vectors for the END-TO-END flow FROM THE REFERENCE: layer-conditioned DDIM sampling on the production network, then
tri-plane decode - BASELINE configs[0] / [3] / [4] at one-GPU test scale.

Runs only in the build container (imports /root/reference/human_diffusion unmodified; ~3 minutes of CPU).

    python tests/golden/gen_golden_chain.py

What the reference's scripts/triplane_sample_layered.py does per subject, restated around the reference's own functions (the script
itself needs MPI, blobfile, datasets and a checkpoint):
  :112-121  y = layer_index                                      -> model_kwargs
  :124-134  x_cond = zeros for layer 0, else the PREVIOUS layer's saved sample
  :139-154  sample = diffusion.ddim_sample_loop(model, (B, 27, 256, 256), x_cond=x_cond, clip_denoised=True, model_kwargs)
  :158      tri_planes = sample[id:id+1].reshape(1, 3, -1, 256, 256)
  :177      render(chunk, rays_o, rays_d, near, far, tri_planes, tp_input, human_nerf, n_samples, perturb=0, n_importance)
            = z_vals linspace + Renderer.render (:262-279)
Configuration: the shipped F4 network (497 M parameters, seeded synthetic weights - humanliff_amd.synthetic.state_from_shapes seed 1),
timestep_respacing="ddim10", B = 1, two cloth layers, one 128x128 orbit view at 32 + 32 samples per ray of each layer's tri-plane.
Noise: x_T and the per-step randn_like draws come from torch.Generator().manual_seed(9000 + draw index); sample_pdf's uniforms from
torch.manual_seed(5) (= synthetic.importance_u(R, 32, 5)).
Stored (the full samples would be 14 MB): every 8th pixel of both samples, their checksums, per-channel means, and the complete rendered
rgb / acc / depth maps.
"""
import os
import sys
import time
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/human_diffusion")

for n in ["mcubes", "cv2", "pytorch3d", "pytorch3d.ops", "pytorch3d.ops.knn"]:
    sys.modules[n] = types.ModuleType(n)
sys.modules["pytorch3d.ops.knn"].knn_points = None

from improved_diffusion.script_util import create_model_and_diffusion  # noqa: E402
from NeRF import renderer as R  # noqa: E402

torch.autograd.set_detect_anomaly(False)
R.read_pickle = lambda p: {}
R.SMPL_to_tensor = lambda params, device: {"f": None}
torch.cuda.current_device = lambda: 0

from humanliff_amd import synthetic as syn  # noqa: E402

F4 = dict(image_size=256, in_channels=27, out_channels=27, num_channels=192, num_res_blocks=3, num_heads=4,
          num_heads_upsample=-1, attention_resolutions="32,16,8", dropout=0.0, learn_sigma=False, sigma_small=False,
          class_cond=True, diffusion_steps=1000, noise_schedule="linear", timestep_respacing="ddim10", use_kl=False,
          predict_xstart=False, rescale_timesteps=False, rescale_learned_sigmas=True, use_checkpoint=False,
          use_scale_shift_norm=True, cond_type="controlnet", use_3d_aware=False)
N_LAYERS, IMG, NS, VIEW, N_VIEWS, STRIDE = 2, 128, 32, 3, 36, 8


def checksum(t):
    return np.array([float(t.double().sum()), float(t.double().abs().sum())])


def main():
    t0 = time.time()
    model, diffusion = create_model_and_diffusion(**F4)
    model.eval()
    ks = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    model.load_state_dict(syn.state_from_shapes(ks, 1), strict=True)
    rend = R.Renderer(use_canonical_space=False, triplane_dim=256, triplane_ch=27, smpl_type="smpl", test=True)
    rend.load_state_dict(syn.render_mlp_state(3), strict=False)
    bounds = torch.tensor(syn.WORLD_BOUNDS)
    rays_o, rays_d, near, far = syn.orbit_rays(VIEW, N_VIEWS, IMG, IMG)
    draws = {"n": 0}

    def draw(shape):
        g = torch.Generator().manual_seed(9000 + draws["n"])
        draws["n"] += 1
        return torch.randn(tuple(shape), generator=g)

    res = {"n_layers": N_LAYERS, "img": IMG, "n_samples": NS, "view": VIEW, "n_views": N_VIEWS, "stride": STRIDE}
    shape = (1, 27, 256, 256)
    x_cond = torch.zeros(shape)
    orig = torch.randn_like
    torch.randn_like = lambda ref: draw(ref.shape)
    try:
        for layer in range(N_LAYERS):
            y = torch.full((1,), layer, dtype=torch.int64)
            x_T = draw(shape)
            with torch.no_grad():
                sample = diffusion.ddim_sample_loop(model, shape, x_cond=x_cond, noise=x_T, clip_denoised=True, model_kwargs={"y": y},
                                                    device=torch.device("cpu"))
            res[f"sample{layer}_sub"] = sample[:, :, ::STRIDE, ::STRIDE].numpy()
            res[f"sample{layer}_ck"] = checksum(sample)
            res[f"sample{layer}_chmean"] = sample.double().mean(dim=(0, 2, 3)).numpy()
            res[f"sample{layer}_row100"] = sample[0, :, 100, :].numpy()
            tri_planes = sample[0:1].reshape(1, 3, -1, 256, 256)
            # render(): triplane_sample_layered.py:262-279 with perturb = 0
            ro, rd, nr, fr = rays_o[None], rays_d[None], near[None, :, None], far[None, :, None]
            t_vals = torch.linspace(0.0, 1.0, steps=NS)
            z = nr * (1.0 - t_vals) + fr * t_vals
            pts = (ro[..., None, :] + rd[..., None, :] * z[..., :, None]).reshape(1, -1, 3)
            torch.manual_seed(5)
            with torch.no_grad():
                out = rend.render({"world_bounds": bounds[None]}, pts, z, ro, rd, nr, fr, tri_planes, NS, False)
            res[f"rgb{layer}"] = out["rgb_map"][0].numpy()
            res[f"acc{layer}"] = out["acc_map"][0].numpy()
            res[f"depth{layer}"] = out["depth_map"][0].numpy()
            print(f"layer {layer}: sample abs mean {float(sample.abs().mean()):.4f} max {float(sample.abs().max()):.4f} "
                  f"rgb mean {float(out['rgb_map'].mean()):.4f} acc mean {float(out['acc_map'].mean()):.4f}  [{time.time() - t0:.0f} s]", flush=True)
            x_cond = sample
    finally:
        torch.randn_like = orig
    res["ndraws"] = draws["n"]
    res["rays_ck"] = checksum(rays_d)
    np.savez_compressed(os.path.join(HERE, "chain_f4_ddim10.npz"), **res)
    print("draws", draws["n"], "file", os.path.getsize(os.path.join(HERE, "chain_f4_ddim10.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
