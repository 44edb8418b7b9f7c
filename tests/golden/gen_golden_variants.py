"""Golden vectors FROM THE REFERENCE for the sampler branches outside the shipped configuration (files SURVEY 8(a) marks on the path):
learned variances (learn_sigma=True: LEARNED_RANGE, LEARNED; gaussian_diffusion.py:262-276), START_X prediction (predict_xstart=True,
:302-303) and denoised_fn (:293-296), single steps of p_sample / ddim_sample / p_mean_variance with a stub model and injected noise.

    python tests/golden/gen_golden_variants.py
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, "/root/reference/human_diffusion")

from improved_diffusion import gaussian_diffusion as gd  # noqa: E402
from improved_diffusion.respace import SpacedDiffusion, space_timesteps  # noqa: E402


def stub(x, t, x_cond, y=None, two=False):
    """+,-,*,clamp only (bit-identical on every host); `two`: 2C output channels, the second half in [-1,1] (variance fraction)."""
    tt = t.float().view(-1, 1, 1, 1) * 0.001
    e = (0.6 * x + 0.25 * x_cond - tt).clamp(-1.5, 1.5) * 1.3
    if not two:
        return e
    v = (0.4 * x - 0.3 * x_cond + tt).clamp(-1, 1)
    return torch.cat([e, v], dim=1)


def main():
    betas = gd.get_named_beta_schedule("linear", 1000)
    g = torch.Generator().manual_seed(7)
    x = torch.randn((3, 27, 8, 8), generator=g)
    xc = torch.randn((3, 27, 8, 8), generator=g) * 0.5
    noise = torch.randn((3, 27, 8, 8), generator=g)
    out = {}
    orig = torch.randn_like
    torch.randn_like = lambda ref: noise.clone()
    try:
        for tag, mean_t, var_t, two in [("range", gd.ModelMeanType.EPSILON, gd.ModelVarType.LEARNED_RANGE, True),
                                        ("learned", gd.ModelMeanType.EPSILON, gd.ModelVarType.LEARNED, True),
                                        ("x0", gd.ModelMeanType.START_X, gd.ModelVarType.FIXED_LARGE, False),
                                        ("x0range", gd.ModelMeanType.START_X, gd.ModelVarType.LEARNED_RANGE, True),
                                        ("xprev", gd.ModelMeanType.PREVIOUS_X, gd.ModelVarType.FIXED_SMALL, False),
                                        ("xprevrange", gd.ModelMeanType.PREVIOUS_X, gd.ModelVarType.LEARNED_RANGE, True)]:
            d = SpacedDiffusion(use_timesteps=space_timesteps(1000, "ddim50"), betas=betas, model_mean_type=mean_t, model_var_type=var_t,
                                loss_type=gd.LossType.MSE, rescale_timesteps=False)
            t = torch.tensor([49, 0, 17])
            model = lambda a, b, c, **k: stub(a, b, c, two=two)  # noqa: E731
            for clip in (True, False):
                c = int(clip)
                ps = d.p_sample(model, x, xc, t, clip_denoised=clip)
                dd = d.ddim_sample(model, x, t, x_cond=xc, clip_denoised=clip, eta=0.3)
                pm = d.p_mean_variance(model, x, t, x_cond=xc, clip_denoised=clip)
                fn = d.p_sample(model, x, xc, t, clip_denoised=clip, denoised_fn=lambda z: 0.5 * z + 0.1)
                out[f"{tag}_{c}_p_sample"] = ps["sample"].numpy()
                out[f"{tag}_{c}_p_x0"] = ps["pred_xstart"].numpy()
                out[f"{tag}_{c}_ddim"] = dd["sample"].numpy()
                out[f"{tag}_{c}_mean"] = pm["mean"].numpy()
                out[f"{tag}_{c}_logvar"] = pm["log_variance"].numpy()
                out[f"{tag}_{c}_fn_sample"] = fn["sample"].numpy()
            out[f"{tag}_t"] = t.numpy()
    finally:
        torch.randn_like = orig
    # ---- training_losses of every LossType / mean type / variance type (:688-772), with the gradient with respect to a scalar the stub
    # model multiplies its conditioning by (the variational-bound terms must be differentiable through mean and log-variance)
    x0 = (torch.randn((3, 27, 8, 8), generator=g) * 0.6).clamp(-1, 1)
    x0[0, :, :2] = -1.0                                      # (the decoder likelihood's open-ended border bins)
    x0[1, :, :2] = 1.0
    tl = torch.tensor([0, 49, 17])
    for tag, loss_t, mean_t, var_t, two in [("kl", gd.LossType.KL, gd.ModelMeanType.EPSILON, gd.ModelVarType.LEARNED_RANGE, True),
                                            ("rkl", gd.LossType.RESCALED_KL, gd.ModelMeanType.START_X, gd.ModelVarType.FIXED_LARGE, False),
                                            ("hyb", gd.LossType.MSE, gd.ModelMeanType.EPSILON, gd.ModelVarType.LEARNED_RANGE, True),
                                            ("rhyb", gd.LossType.RESCALED_MSE, gd.ModelMeanType.EPSILON, gd.ModelVarType.LEARNED, True),
                                            ("msexp", gd.LossType.MSE, gd.ModelMeanType.PREVIOUS_X, gd.ModelVarType.FIXED_SMALL, False),
                                            ("klxp", gd.LossType.KL, gd.ModelMeanType.PREVIOUS_X, gd.ModelVarType.LEARNED, True)]:
        d = SpacedDiffusion(use_timesteps=space_timesteps(1000, "ddim50"), betas=betas, model_mean_type=mean_t, model_var_type=var_t,
                            loss_type=loss_t, rescale_timesteps=False)
        p = torch.tensor(0.8, requires_grad=True)
        model = lambda a, b, c, **k: stub(a, b, (xc if c is None else c) * p, two=two)  # noqa: E731   (the KL losses call the model without x_cond)
        terms = d.training_losses(model, x0, xc, tl, noise=noise)
        terms["loss"].sum().backward()
        for k, v in terms.items():
            out[f"loss_{tag}_{k}"] = v.detach().numpy()
        out[f"loss_{tag}_dp"] = p.grad.numpy()
    out["loss_x0"] = x0.numpy()
    out["loss_t"] = tl.numpy()
    np.savez_compressed(os.path.join(HERE, "diffusion_variants.npz"), **out)
    print("ok", len(out), "arrays", os.path.getsize(os.path.join(HERE, "diffusion_variants.npz")) // 1024, "KiB")


def unet_concat():
    """cond_type='concat' (unet.py:572-573: x = cat([x, x_cond], 1), in_channels counts both) and cond_type='' on the tiny net."""
    from improved_diffusion.script_util import create_model_and_diffusion, model_and_diffusion_defaults
    from humanliff_amd import synthetic as syn
    out = {}
    for tag, cond, cin in (("concat", "concat", 54), ("plain", "", 27)):
        a = model_and_diffusion_defaults()
        a.update(dict(in_channels=cin, out_channels=27, class_cond=True, learn_sigma=False, num_heads=4, use_scale_shift_norm=True,
                      cond_type=cond, rescale_timesteps=False, dropout=0.0, image_size=32, num_channels=32, num_res_blocks=1,
                      attention_resolutions="16,8"))
        model, _ = create_model_and_diffusion(**a)
        model.eval()
        ks = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
        model.load_state_dict(syn.state_from_shapes(ks, 1), strict=True)
        g = torch.Generator().manual_seed(7)
        x = torch.randn((2, 27, 32, 32), generator=g)
        xc = torch.randn((2, 27, 32, 32), generator=g).clamp(-1, 1) * 0.7
        with torch.no_grad():
            y = model(x, torch.tensor([999, 17]), xc if cond else None, y=torch.tensor([3, 0]))
        out[f"{tag}_out"] = y.numpy()
        out[f"{tag}_nkeys"] = len(ks)
        print(tag, "keys", len(ks), "out abs mean", float(y.abs().mean()))
    # cond_type='AdaGN' (unet.py:519-525, 574-578): Linear(64*64, E) fixes the input at 256x256; a narrow 6-level net, batch 1.  The
    # output is 27x256x256: every 8th pixel is stored plus the sum and the sum of absolute values of all of them (float64)
    a = model_and_diffusion_defaults()
    a.update(dict(in_channels=27, out_channels=27, class_cond=True, learn_sigma=False, num_heads=2, use_scale_shift_norm=True,
                  cond_type="AdaGN", rescale_timesteps=False, dropout=0.0, image_size=256, num_channels=32, num_res_blocks=1,
                  attention_resolutions="32,16,8"))
    model, _ = create_model_and_diffusion(**a)
    model.eval()
    ks = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    model.load_state_dict(syn.state_from_shapes(ks, 1), strict=True)
    g = torch.Generator().manual_seed(11)
    x = torch.randn((1, 27, 256, 256), generator=g)
    xc = torch.randn((1, 27, 256, 256), generator=g).clamp(-1, 1) * 0.7
    with torch.no_grad():
        y = model(x, torch.tensor([412]), xc, y=torch.tensor([731]))
        y0 = model(x, torch.tensor([412]), torch.zeros_like(xc), y=torch.tensor([731]))
    out["adagn_out_s8"] = y[:, :, ::8, ::8].numpy()
    out["adagn_sums"] = np.array([float(y.double().sum()), float(y.double().abs().sum())])
    out["adagn_nkeys"] = len(ks)
    out["adagn_cond_effect"] = float((y - y0).abs().max())       # the condition matters (the projection is not a no-op)
    print("AdaGN keys", len(ks), "out abs mean", float(y.abs().mean()), "effect of x_cond", out["adagn_cond_effect"])
    # cond_type='cross_attention' (unet.py:404-405, 427, 463, 579-582; spatial_transformer.py): SpatialTransformer blocks attending to one
    # context token = the same projection of x_cond; the narrow 256x256 net again
    a = model_and_diffusion_defaults()
    a.update(dict(in_channels=27, out_channels=27, class_cond=True, learn_sigma=False, num_heads=2, use_scale_shift_norm=True,
                  cond_type="cross_attention", rescale_timesteps=False, dropout=0.0, image_size=256, num_channels=32, num_res_blocks=1,
                  attention_resolutions="32,16,8"))
    model, _ = create_model_and_diffusion(**a)
    model.eval()
    ks = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    model.load_state_dict(syn.state_from_shapes(ks, 1), strict=True)
    g = torch.Generator().manual_seed(17)
    x = torch.randn((1, 27, 256, 256), generator=g)
    xc = torch.randn((1, 27, 256, 256), generator=g).clamp(-1, 1) * 0.7
    with torch.no_grad():
        y = model(x, torch.tensor([412]), xc, y=torch.tensor([2]))
        y0 = model(x, torch.tensor([412]), torch.zeros_like(xc), y=torch.tensor([2]))
    out["xattn_out_s8"] = y[:, :, ::8, ::8].numpy()
    out["xattn_sums"] = np.array([float(y.double().sum()), float(y.double().abs().sum())])
    out["xattn_nkeys"] = len(ks)
    out["xattn_cond_effect"] = float((y - y0).abs().max())
    print("cross_attention keys", len(ks), "out abs mean", float(y.abs().mean()), "effect of x_cond", out["xattn_cond_effect"])
    # use_3d_aware=True (unet.py:158-166, 208-214, 566-570, 613-614): 27-channel tri-planes, a 9-channel network on the planes side by
    # side; with the control tower (whose ResBlocks stay plain, :477-518) and without conditioning
    # ... and with cond_type='concat' (:566-573: x and x_cond are rolled out plane by plane FIRST and joined after, so plane p of the
    # 18-channel network input is [x_p | cond_p])
    for tag, cond in (("aware3d_controlnet", "controlnet"), ("aware3d_plain", ""), ("aware3d_concat", "concat")):
        a = model_and_diffusion_defaults()
        a.update(dict(in_channels=18 if cond == "concat" else 9, out_channels=9, class_cond=True, learn_sigma=False, num_heads=4, use_scale_shift_norm=True,
                      cond_type=cond, use_3d_aware=True, rescale_timesteps=False, dropout=0.0, image_size=32, num_channels=32,
                      num_res_blocks=1, attention_resolutions="16,8"))
        model, _ = create_model_and_diffusion(**a)
        model.eval()
        ks = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
        model.load_state_dict(syn.state_from_shapes(ks, 1), strict=True)
        g = torch.Generator().manual_seed(13)
        x = torch.randn((2, 27, 32, 32), generator=g)
        xc = torch.randn((2, 27, 32, 32), generator=g).clamp(-1, 1) * 0.7
        with torch.no_grad():
            y = model(x, torch.tensor([999, 17]), xc if cond else None, y=torch.tensor([3, 0]))
        out[f"{tag}_out"] = y.numpy()
        out[f"{tag}_nkeys"] = len(ks)
        print(tag, "keys", len(ks), "out", tuple(y.shape), "abs mean", float(y.abs().mean()))
    np.savez_compressed(os.path.join(HERE, "unet_cond_types.npz"), **out)


if __name__ == "__main__":
    main()
    unet_concat()
