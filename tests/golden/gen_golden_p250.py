"""Golden vectors FROM THE REFERENCE for the SHIPPED sampler configuration on the production network: `--batch_size 1
--timestep_respacing 250`, `p_sample_loop` (human_diffusion/triplane_scripts/SynBody_triplane_sample_layered_*.sh:24-26).

Runs only in the build container (imports /root/reference/human_diffusion unmodified; ~25 minutes of CPU).

    python tests/golden/gen_golden_p250.py

This file follows gen_golden_ddim50.py; what differs: timestep_respacing="250" (250 of the 1000 steps, DDPM ancestral sampling: every
step's `randn_like` draw is USED), `p_sample_loop_progressive`, states kept after steps 1, 50, 125, 200, 250.  The remainder of this
docstring is the DDIM-50 generator's description of the common set-up:

`scripts/triplane_sample_layered.py:139-154` calls `diffusion.ddim_sample_loop(model, (B, 27, 256, 256), x_cond=..., clip_denoised=True,
model_kwargs={"y": layer})` with `timestep_respacing="250"` (BASELINE configs[3] / [4]). Here: the shipped F4 network (497 M parameters,
seeded synthetic weights - humanliff_amd.synthetic.state_from_shapes seed 1), B = 1, cloth layer 1 conditioned on a seeded tri-plane-like
x_cond (clamp(0.3 randn, -1, 1), seed 77), x_T from seed 9100, the per-step `randn_like` draws (drawn although eta = 0,
gaussian_diffusion.py:521) from seeds 9101.. . The loop is walked through the reference's `ddim_sample_loop_progressive` so that
intermediate states can be kept: every 8th pixel of the sample after steps 1, 10, 25, 40 and 50, whole-tensor checksums of all of them,
one full row of the final sample and its per-channel means.
"""
import os
import sys
import time

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/human_diffusion")

from improved_diffusion.script_util import create_model_and_diffusion  # noqa: E402

from humanliff_amd import synthetic as syn  # noqa: E402

F4 = dict(image_size=256, in_channels=27, out_channels=27, num_channels=192, num_res_blocks=3, num_heads=4,
          num_heads_upsample=-1, attention_resolutions="32,16,8", dropout=0.0, learn_sigma=False, sigma_small=False,
          class_cond=True, diffusion_steps=1000, noise_schedule="linear", timestep_respacing="250", use_kl=False,
          predict_xstart=False, rescale_timesteps=False, rescale_learned_sigmas=True, use_checkpoint=False,
          use_scale_shift_norm=True, cond_type="controlnet", use_3d_aware=False)
STRIDE, KEEP, LAYER = 8, (1, 50, 125, 200, 250), 1


def checksum(t):
    return np.array([float(t.double().sum()), float(t.double().abs().sum())])


def main():
    t0 = time.time()
    torch.set_num_threads(8)
    model, diffusion = create_model_and_diffusion(**F4)
    model.eval()
    ks = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    model.load_state_dict(syn.state_from_shapes(ks, 1), strict=True)
    draws = {"n": 0}

    def draw(shape):
        g = torch.Generator().manual_seed(9100 + draws["n"])
        draws["n"] += 1
        return torch.randn(tuple(shape), generator=g)

    shape = (1, 27, 256, 256)
    x_cond = (torch.randn(shape, generator=torch.Generator().manual_seed(77)) * 0.3).clamp_(-1, 1)
    y = torch.full((1,), LAYER, dtype=torch.int64)
    res = {"stride": STRIDE, "keep": np.array(KEEP), "layer": LAYER, "x_cond_ck": checksum(x_cond)}
    orig = torch.randn_like
    torch.randn_like = lambda ref: draw(ref.shape)
    try:
        x_T = draw(shape)
        with torch.no_grad():
            for n, out in enumerate(diffusion.p_sample_loop_progressive(model, shape, x_cond=x_cond, noise=x_T, clip_denoised=True,
                                                                        model_kwargs={"y": y}, device=torch.device("cpu")), 1):
                s = out["sample"]
                if n in KEEP:
                    res[f"step{n}_sub"] = s[:, :, ::STRIDE, ::STRIDE].numpy().copy()
                    res[f"step{n}_ck"] = checksum(s)
                    print(f"step {n}: abs mean {float(s.abs().mean()):.4f} max {float(s.abs().max()):.4f} [{time.time() - t0:.0f} s]", flush=True)
    finally:
        torch.randn_like = orig
    res["final_row100"] = s[0, :, 100, :].numpy()
    res["final_chmean"] = s.double().mean(dim=(0, 2, 3)).numpy()
    res["ndraws"] = draws["n"]
    path = os.path.join(HERE, "f4_p250.npz")
    np.savez_compressed(path, **res)
    print("draws", draws["n"], "file", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
