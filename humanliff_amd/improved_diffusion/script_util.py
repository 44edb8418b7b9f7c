"""Factory functions with the reference's names and kwargs (human_diffusion/improved_diffusion/script_util.py):
model_and_diffusion_defaults (:11-39), create_model_and_diffusion (:42-95), create_model (:98-150),
create_gaussian_diffusion (:260-298) and the argparse helpers (:301-326)."""
import argparse

from . import gaussian_diffusion as gd
from .respace import SpacedDiffusion, space_timesteps
from .unet import UNetModel

NUM_CLASSES = 4

_CHANNEL_MULT = {256: (1, 1, 2, 2, 4, 4), 224: (1, 1, 2, 2, 4, 4), 192: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 2, 4, 4),
                 64: (1, 2, 3, 4), 32: (1, 2, 2, 2)}


def model_and_diffusion_defaults():
    return dict(image_size=64, in_channels=3, num_channels=128, out_channels=3, num_res_blocks=2, num_heads=4,
                num_heads_upsample=-1, attention_resolutions="16,8", dropout=0.0, learn_sigma=False, sigma_small=False,
                class_cond=False, diffusion_steps=1000, noise_schedule="linear", timestep_respacing="", use_kl=False,
                predict_xstart=False, rescale_timesteps=True, rescale_learned_sigmas=True, use_checkpoint=False,
                use_scale_shift_norm=True, cond_type="controlnet", use_3d_aware=False)


def create_model_and_diffusion(image_size, class_cond, learn_sigma, sigma_small, in_channels, num_channels, out_channels,
                               num_res_blocks, num_heads, num_heads_upsample, attention_resolutions, dropout,
                               diffusion_steps, noise_schedule, timestep_respacing, use_kl, predict_xstart,
                               rescale_timesteps, rescale_learned_sigmas, use_checkpoint, use_scale_shift_norm, cond_type,
                               use_3d_aware):
    model = create_model(image_size, in_channels, num_channels, out_channels, num_res_blocks, learn_sigma=learn_sigma,
                         class_cond=class_cond, use_checkpoint=use_checkpoint, attention_resolutions=attention_resolutions,
                         num_heads=num_heads, num_heads_upsample=num_heads_upsample,
                         use_scale_shift_norm=use_scale_shift_norm, cond_type=cond_type, use_3d_aware=use_3d_aware,
                         dropout=dropout)
    diffusion = create_gaussian_diffusion(steps=diffusion_steps, learn_sigma=learn_sigma, sigma_small=sigma_small,
                                          noise_schedule=noise_schedule, use_kl=use_kl, predict_xstart=predict_xstart,
                                          rescale_timesteps=rescale_timesteps,
                                          rescale_learned_sigmas=rescale_learned_sigmas,
                                          timestep_respacing=timestep_respacing)
    return model, diffusion


def create_model(image_size, in_channels, num_channels, out_channels, num_res_blocks, learn_sigma, class_cond,
                 use_checkpoint, attention_resolutions, num_heads, num_heads_upsample, use_scale_shift_norm, cond_type,
                 use_3d_aware, dropout):
    if image_size not in _CHANNEL_MULT:
        raise ValueError(f"unsupported image size: {image_size}")
    attention_ds = tuple(image_size // int(res) for res in attention_resolutions.split(","))
    n_classes = 1000 if (cond_type == 'AdaGN' and not use_3d_aware) else NUM_CLASSES
    return UNetModel(in_channels=in_channels, model_channels=num_channels,
                     out_channels=(out_channels if not learn_sigma else out_channels * 2), num_res_blocks=num_res_blocks,
                     attention_resolutions=attention_ds, dropout=dropout, channel_mult=_CHANNEL_MULT[image_size],
                     num_classes=(n_classes if class_cond else None), use_checkpoint=use_checkpoint, num_heads=num_heads,
                     num_heads_upsample=num_heads_upsample, use_scale_shift_norm=use_scale_shift_norm, cond_type=cond_type,
                     use_3d_aware=use_3d_aware)


def create_gaussian_diffusion(*, steps=1000, learn_sigma=False, sigma_small=False, noise_schedule="linear", use_kl=False,
                              predict_xstart=False, rescale_timesteps=False, rescale_learned_sigmas=False,
                              timestep_respacing=""):
    betas = gd.get_named_beta_schedule(noise_schedule, steps)
    if use_kl:
        loss_type = gd.LossType.RESCALED_KL
    elif rescale_learned_sigmas:
        loss_type = gd.LossType.RESCALED_MSE
    else:
        loss_type = gd.LossType.MSE
    if learn_sigma:
        var_type = gd.ModelVarType.LEARNED_RANGE
    else:
        var_type = gd.ModelVarType.FIXED_SMALL if sigma_small else gd.ModelVarType.FIXED_LARGE
    return SpacedDiffusion(use_timesteps=space_timesteps(steps, timestep_respacing or [steps]), betas=betas,
                           model_mean_type=(gd.ModelMeanType.START_X if predict_xstart else gd.ModelMeanType.EPSILON),
                           model_var_type=var_type, loss_type=loss_type, rescale_timesteps=rescale_timesteps)


def add_dict_to_argparser(parser, default_dict):
    for k, v in default_dict.items():
        v_type = str if v is None else (str2bool if isinstance(v, bool) else type(v))
        parser.add_argument(f"--{k}", default=v, type=v_type)


def args_to_dict(args, keys):
    return {k: getattr(args, k) for k in keys}


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("boolean value expected")
