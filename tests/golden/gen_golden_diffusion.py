"""Generate golden vectors for the diffusion sampler + UNet FROM THE REFERENCE.

Runs only in the build container (imports /root/reference/human_diffusion/improved_diffusion
unmodified).  Weights and inputs are rebuilt from seeds by humanliff_amd.synthetic; the fixtures
hold the reference's outputs, the state_dict key/shape list and input checksums.

    python tests/golden/gen_golden_diffusion.py

Exercised: script_util.create_model_and_diffusion (script_util.py:42), UNetModel.forward
(unet.py:550), SpacedDiffusion / space_timesteps (respace.py:7,63), GaussianDiffusion.p_sample /
ddim_sample / p_sample_loop / ddim_sample_loop / p_mean_variance (gaussian_diffusion.py:232-651),
timestep_embedding (nn.py:103).
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
from improved_diffusion import nn as rnn  # noqa: E402
from improved_diffusion.respace import SpacedDiffusion, space_timesteps  # noqa: E402
from improved_diffusion.script_util import create_model_and_diffusion, model_and_diffusion_defaults  # noqa: E402

from humanliff_amd import synthetic as syn  # noqa: E402

UNET_CASES = {
    # name: (overrides, batch, timesteps, labels)
    "tiny32": (dict(image_size=32, num_channels=32, num_res_blocks=1, attention_resolutions="16,8"), 2, [999, 17], [3, 0]),
    "mid64": (dict(image_size=64, num_channels=64, num_res_blocks=2, attention_resolutions="16,8"), 1, [431], [2]),
    "deep256": (dict(image_size=256, num_channels=32, num_res_blocks=1, attention_resolutions="32,16,8"), 1, [250], [1]),
}


def unet_args(over):
    a = model_and_diffusion_defaults()
    a.update(dict(in_channels=27, out_channels=27, class_cond=True, learn_sigma=False, num_heads=4,
                  use_scale_shift_norm=True, cond_type="controlnet", rescale_timesteps=False, dropout=0.0,
                  diffusion_steps=1000, noise_schedule="linear", timestep_respacing=""))
    a.update(over)
    return a


def load_seeded(model, seed):
    ks = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    sd = syn.state_from_shapes(ks, seed)
    model.load_state_dict(sd, strict=True)
    return ks


def unet_inputs(B, size, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((B, 27, size, size), generator=g)
    xc = torch.randn((B, 27, size, size), generator=g).clamp(-1, 1) * 0.7
    return x, xc


def checksum(t):
    return np.array([float(t.double().sum()), float(t.double().abs().sum())])


def gen_unet():
    for name, (over, B, ts, ys) in UNET_CASES.items():
        args = unet_args(over)
        model, _ = create_model_and_diffusion(**args)
        model.eval()
        ks = load_seeded(model, seed=1)
        x, xc = unet_inputs(B, args["image_size"], seed=7)
        t = torch.tensor(ts, dtype=torch.int64)
        y = torch.tensor(ys, dtype=torch.int64)
        with torch.no_grad():
            out = model(x, t, xc, y=y)
            emb = rnn.timestep_embedding(t, args["num_channels"])
        full = out.numpy()
        stride = 1 if args["image_size"] <= 64 else 8
        np.savez_compressed(
            os.path.join(HERE, f"unet_{name}.npz"),
            keys=np.array([k for k, _ in ks]), shapes=np.array([str(list(s)) for _, s in ks]),
            n_params=sum(int(np.prod(s)) for _, s in ks), stride=stride, B=B, t=np.array(ts), y=np.array(ys),
            out=full[:, :, ::stride, ::stride], out_ck=checksum(out), x_ck=checksum(x), xc_ck=checksum(xc),
            temb=emb.numpy(),
            **{f"arg_{k}": np.array(v) for k, v in args.items()},
        )
        print(name, "params", sum(int(np.prod(s)) for _, s in ks), "out abs mean", float(out.abs().mean()),
              "max", float(out.abs().max()))


def stub_model(x, t, x_cond, y=None):
    """Cheap deterministic eps-predictor used to pin the sampler arithmetic without a UNet.
    Only +,-,*,clamp: bit-identical on every CPU ISA (tanh/exp would differ by an ulp between hosts)."""
    tt = t.float().view(-1, 1, 1, 1) * 0.001
    yy = 0.0 if y is None else y.float().view(-1, 1, 1, 1) * 0.05
    return (0.6 * x + 0.25 * x_cond - tt + yy).clamp(-1.5, 1.5) * 1.3


def gen_schedules_and_steps():
    out = {}
    betas = gd.get_named_beta_schedule("linear", 1000)
    for tag, spec in [("full", [1000]), ("r250", "250"), ("ddim50", "ddim50"), ("ddim10", "ddim10"), ("mix", "10,15,20")]:
        d = SpacedDiffusion(use_timesteps=space_timesteps(1000, spec), betas=betas,
                            model_mean_type=gd.ModelMeanType.EPSILON, model_var_type=gd.ModelVarType.FIXED_LARGE,
                            loss_type=gd.LossType.MSE, rescale_timesteps=False)
        out[f"{tag}_map"] = np.array(d.timestep_map)
        out[f"{tag}_betas"] = d.betas
        out[f"{tag}_post_logvar"] = d.posterior_log_variance_clipped
        out[f"{tag}_coef1"] = d.posterior_mean_coef1
        out[f"{tag}_coef2"] = d.posterior_mean_coef2
        out[f"{tag}_sqrt_recip"] = d.sqrt_recip_alphas_cumprod
        out[f"{tag}_sqrt_recipm1"] = d.sqrt_recipm1_alphas_cumprod
    cos = gd.get_named_beta_schedule("cosine", 50)
    out["cosine50_betas"] = cos
    # single steps with the stub model and injected noise
    g = torch.Generator().manual_seed(7)
    x = torch.randn((3, 27, 8, 8), generator=g)
    xc = torch.randn((3, 27, 8, 8), generator=g) * 0.5
    noise = torch.randn((3, 27, 8, 8), generator=g)
    y = torch.tensor([0, 3, 1])
    orig = torch.randn_like
    torch.randn_like = lambda ref: noise.clone()
    try:
        for tag, spec in [("full", [1000]), ("ddim50", "ddim50"), ("r250", "250")]:
            d = SpacedDiffusion(use_timesteps=space_timesteps(1000, spec), betas=betas,
                                model_mean_type=gd.ModelMeanType.EPSILON, model_var_type=gd.ModelVarType.FIXED_LARGE,
                                loss_type=gd.LossType.MSE, rescale_timesteps=False)
            T = d.num_timesteps
            t = torch.tensor([T - 1, 0, T // 3])
            for clip in (True, False):
                ps = d.p_sample(stub_model, x, xc, t, clip_denoised=clip, model_kwargs={"y": y})
                dd = d.ddim_sample(stub_model, x, t, x_cond=xc, clip_denoised=clip, model_kwargs={"y": y}, eta=0.0)
                de = d.ddim_sample(stub_model, x, t, x_cond=xc, clip_denoised=clip, model_kwargs={"y": y}, eta=0.7)
                pm = d.p_mean_variance(stub_model, x, t, x_cond=xc, clip_denoised=clip, model_kwargs={"y": y})
                c = int(clip)
                out[f"step_{tag}_{c}_t"] = t.numpy()
                out[f"step_{tag}_{c}_p_sample"] = ps["sample"].numpy()
                out[f"step_{tag}_{c}_p_x0"] = ps["pred_xstart"].numpy()
                out[f"step_{tag}_{c}_ddim_sample"] = dd["sample"].numpy()
                out[f"step_{tag}_{c}_ddim_eta_sample"] = de["sample"].numpy()
                out[f"step_{tag}_{c}_mean"] = pm["mean"].numpy()
                out[f"step_{tag}_{c}_logvar"] = pm["log_variance"][:, 0, 0, 0].numpy()
                out[f"step_{tag}_{c}_var"] = pm["variance"][:, 0, 0, 0].numpy()
    finally:
        torch.randn_like = orig
    out["step_x_ck"] = checksum(x)
    te = rnn.timestep_embedding(torch.tensor([0, 1, 500, 999]), 192)
    out["temb192"] = te.numpy()
    np.savez_compressed(os.path.join(HERE, "diffusion_steps.npz"), **out)
    print("schedules+steps ok")


def gen_loops():
    """Whole sampling loops on the tiny UNet with injected noise (seed 7 + draw index)."""
    args = unet_args(UNET_CASES["tiny32"][0])
    res = {}
    for tag, respacing, use_ddim in [("ddim10", "ddim10", True), ("p8", "8", False)]:
        a = dict(args)
        a["timestep_respacing"] = respacing
        model, diffusion = create_model_and_diffusion(**a)
        model.eval()
        load_seeded(model, seed=1)
        B = 2
        draws = {"n": 0}

        def draw(shape):
            g = torch.Generator().manual_seed(7000 + draws["n"])
            draws["n"] += 1
            return torch.randn(tuple(shape), generator=g)

        x_T = draw((B, 27, 32, 32))
        _, xc = unet_inputs(B, 32, seed=7)
        y = torch.tensor([1, 2])
        orig = torch.randn_like
        torch.randn_like = lambda ref: draw(ref.shape)
        try:
            fn = diffusion.ddim_sample_loop if use_ddim else diffusion.p_sample_loop
            sample = fn(model, (B, 27, 32, 32), x_cond=xc, noise=x_T, clip_denoised=True, model_kwargs={"y": y},
                        device=torch.device("cpu"))
        finally:
            torch.randn_like = orig
        res[f"{tag}_sample"] = sample.numpy()
        res[f"{tag}_ndraws"] = draws["n"]
        print(tag, "draws", draws["n"], "sample abs mean", float(sample.abs().mean()))
    np.savez_compressed(os.path.join(HERE, "diffusion_loops.npz"), **res)


if __name__ == "__main__":
    gen_schedules_and_steps()
    gen_unet()
    gen_loops()
