"""Golden vectors for LONG sampling loops FROM THE REFERENCE: recurrent-error evidence over 50 / 250 / 1000 steps.

Runs only in the build container (imports /root/reference/human_diffusion/improved_diffusion unmodified).

    python tests/golden/gen_golden_drift.py

The one-step and 8/10-step fixtures (gen_golden_diffusion.py) cannot show how the fp32 differences between two
implementations of the same network grow when the output of step i is the input of step i+1.  Here the reference runs
  ddim50   ddim_sample_loop, timestep_respacing="ddim50", eta 0     (BASELINE configs[3]'s schedule)       tiny32, mid64
  r250     p_sample_loop,    timestep_respacing="250"                (README.md:154's sampling schedule)   tiny32
  full     p_sample_loop,    timestep_respacing="" (1000 steps)      (BASELINE configs[1]'s schedule)      tiny32
on the small controlnet + class-cond nets of gen_golden_diffusion.py with injected noise (x_T and every per-step randn_like draw come
from torch.Generator().manual_seed(7000 + draw index), rebuilt identically by the tests).  Stored: the final sample, the sample after
half of the steps, the number of draws.
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, "/root/reference/human_diffusion")

from improved_diffusion.script_util import create_model_and_diffusion  # noqa: E402

from gen_golden_diffusion import UNET_CASES, load_seeded, unet_args, unet_inputs  # noqa: E402

CASES = [
    # tag, net, respacing, ddim, batch, labels
    ("tiny32_ddim50", "tiny32", "ddim50", True, 2, [1, 2]),
    ("tiny32_r250", "tiny32", "250", False, 2, [1, 2]),
    ("tiny32_full", "tiny32", "", False, 2, [3, 0]),
    ("mid64_ddim50", "mid64", "ddim50", True, 1, [2]),
]


def main():
    res = {}
    for tag, net, respacing, use_ddim, B, ys in CASES:
        a = unet_args(UNET_CASES[net][0])
        a["timestep_respacing"] = respacing
        model, diffusion = create_model_and_diffusion(**a)
        model.eval()
        load_seeded(model, seed=1)
        size = a["image_size"]
        draws = {"n": 0}

        def draw(shape):
            g = torch.Generator().manual_seed(7000 + draws["n"])
            draws["n"] += 1
            return torch.randn(tuple(shape), generator=g)

        x_T = draw((B, 27, size, size))
        _, xc = unet_inputs(B, size, seed=7)
        y = torch.tensor(ys)
        orig = torch.randn_like
        torch.randn_like = lambda ref: draw(ref.shape)
        try:
            fn = diffusion.ddim_sample_loop_progressive if use_ddim else diffusion.p_sample_loop_progressive
            T = diffusion.num_timesteps
            half = None
            with torch.no_grad():
                for i, out in enumerate(fn(model, (B, 27, size, size), x_cond=xc, noise=x_T, clip_denoised=True, model_kwargs={"y": y},
                                           device=torch.device("cpu"))):
                    if i == T // 2 - 1:
                        half = out["sample"].clone()
            sample = out["sample"]
        finally:
            torch.randn_like = orig
        res[f"{tag}_sample"] = sample.numpy()
        res[f"{tag}_half"] = half.numpy()
        res[f"{tag}_ndraws"] = draws["n"]
        res[f"{tag}_steps"] = T
        print(tag, "steps", T, "draws", draws["n"], "sample abs mean", float(sample.abs().mean()), "max", float(sample.abs().max()), flush=True)
    np.savez_compressed(os.path.join(HERE, "diffusion_drift.npz"), **res)


if __name__ == "__main__":
    main()
