"""Generate golden vectors for GaussianDiffusion.training_losses + autograd FROM THE REFERENCE (training_losses must
keep working through autograd).  Runs only in the build container (imports /root/reference/human_diffusion/improved_diffusion).

    python tests/golden/gen_golden_train_loss.py

Exercised: create_model_and_diffusion (script_util.py:42), GaussianDiffusion.training_losses with LossType.MSE / epsilon prediction
(gaussian_diffusion.py:688-772: q_sample, UNetModel.forward with autograd, mean_flat MSE) on the "tiny32" controlnet + class-cond UNet
of gen_golden_diffusion.py, then loss.mean().backward().  The fixture holds the losses and the gradients of eight parameters spread
over the network plus a checksum of all of them (weights / inputs are rebuilt from seeds by humanliff_amd.synthetic).
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

from tests.golden.gen_golden_diffusion import load_seeded, unet_args, unet_inputs  # noqa: E402

PICK = ["time_embed.0.weight", "input_blocks.0.0.weight", "input_blocks.1.0.emb_layers.1.weight", "input_blocks.2.1.qkv.weight",
        "middle_block.0.in_layers.2.weight", "input_blocks_cond.1.0.out_layers.3.weight", "input_blocks_proj_cond.2.weight",
        "output_blocks.2.0.skip_connection.weight", "out.2.weight", "label_emb.weight"]

if __name__ == "__main__":
    args = unet_args(dict(image_size=32, num_channels=32, num_res_blocks=1, attention_resolutions="16,8"))
    model, diffusion = create_model_and_diffusion(**args)
    load_seeded(model, seed=1)
    x0, xc = unet_inputs(2, 32, seed=7)
    g = torch.Generator().manual_seed(11)
    noise = torch.randn(x0.shape, generator=g)
    t = torch.tensor([999, 17])
    y = torch.tensor([3, 0])
    losses = diffusion.training_losses(model, x0.clamp(-1, 1), xc, t, model_kwargs={"y": y}, noise=noise)
    losses["loss"].mean().backward()
    sd = dict(model.named_parameters())
    pick = [k for k in PICK if k in sd]
    assert len(pick) >= 8, pick
    tot = sum(float(p.grad.double().abs().sum()) for p in sd.values())
    np.savez_compressed(os.path.join(HERE, "train_loss_tiny32.npz"), loss=losses["loss"].detach().numpy(), mse=losses["mse"].detach().numpy(),
                        noise=noise.numpy(), grad_abs_sum=tot, keys=np.array(pick), **{"g_" + k: sd[k].grad.numpy() for k in pick})
    print("loss", losses["loss"].tolist(), "grad abs sum", tot, pick)
