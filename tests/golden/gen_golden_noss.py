"""Golden vectors FROM THE REFERENCE for use_scale_shift_norm=False (unet.py:186-191, 216-218: emb_layers emits C values that are added to
the first convolution's output before out_layers' GroupNorm): the tiny controlnet + class-cond net's forward, and
GaussianDiffusion.training_losses + backward through it (loss, ten parameter gradients, the sum of |grad| over all of them).

    python tests/golden/gen_golden_noss.py
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
from tests.golden.gen_golden_train_loss import PICK  # noqa: E402

if __name__ == "__main__":
    args = unet_args(dict(image_size=32, num_channels=32, num_res_blocks=1, attention_resolutions="16,8", use_scale_shift_norm=False))
    model, diffusion = create_model_and_diffusion(**args)
    load_seeded(model, seed=1)
    assert model.input_blocks[1][0].emb_layers[1].weight.shape[0] == 32          # C, not 2C
    x0, xc = unet_inputs(2, 32, seed=7)
    t, y = torch.tensor([999, 17]), torch.tensor([3, 0])
    model.eval()
    with torch.no_grad():
        fwd = model(x0, t, xc, y=y)
    model.train()
    noise = torch.randn(x0.shape, generator=torch.Generator().manual_seed(11))
    losses = diffusion.training_losses(model, x0.clamp(-1, 1), xc, t, model_kwargs={"y": y}, noise=noise)
    losses["loss"].mean().backward()
    sd = dict(model.named_parameters())
    pick = [k for k in PICK if k in sd]
    tot = sum(float(p.grad.double().abs().sum()) for p in sd.values())
    np.savez_compressed(os.path.join(HERE, "unet_noss.npz"), out=fwd.numpy(), nkeys=len(model.state_dict()), loss=losses["loss"].detach().numpy(),
                        noise=noise.numpy(), grad_abs_sum=tot, keys=np.array(pick), **{"g_" + k: sd[k].grad.numpy() for k in pick})
    print("forward abs mean", float(fwd.abs().mean()), "loss", losses["loss"].tolist(), "grad abs sum", tot)
