"""ctypes binding of libhumanliff_hip.so (the C ABI declared in include/humanliff_hip.h).

There is no fallback: if the HIP library is missing or a call fails, this raises.
"""
import contextlib
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HL_LIB_PATH") or os.path.join(_HERE, "libhumanliff_hip.so")   # (HL_LIB_PATH: developer A / B of build variants)

HL_RENDER_MLP_FP16 = 16
HL_RENDER_MLP_BF16X3 = 32
HL_RENDER_MLP_FP16X2 = 64
HL_RENDER_FOUR_LAUNCH = 128
HL_CONV_FP32 = 0
HL_CONV_BF16X3 = 1
HL_CONV_FP32_F23 = 3
HL_CONV_FP32_DIRECT = 2
HL_CONV_BF16 = 4
HL_CONV_FP16 = 5
HL_CONV_FP32_MFMA = 6
HL_RENDER_WHITE_BKGD = 1
HL_RENDER_NORMALIZE_DEPTH = 2
HL_RENDER_REEVALUATE = 4
HL_RENDER_CLAMP_DEPTH = 8


class HipLibraryMissing(RuntimeError):
    pass


class HipCallError(RuntimeError):
    pass


class RenderMlpParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "pts0_w", "pts0_b", "pts1_w", "pts1_b", "pts2_w", "pts2_b", "feat_w", "feat_b",
        "alpha_w", "alpha_b", "views_w", "views_b", "rgb_w", "rgb_b")]


class UNetCfg(C.Structure):
    _fields_ = [("in_channels", C.c_int), ("model_channels", C.c_int), ("out_channels", C.c_int),
                ("num_res_blocks", C.c_int), ("n_levels", C.c_int), ("channel_mult", C.c_int * 8),
                ("n_attention_ds", C.c_int), ("attention_ds", C.c_int * 8), ("num_heads", C.c_int),
                ("num_heads_upsample", C.c_int), ("num_classes", C.c_int), ("controlnet", C.c_int), ("adagn", C.c_int), ("cross_attn", C.c_int), ("aware3d", C.c_int),
                ("no_scale_shift", C.c_int)]


_lib = None

_p, _i, _i64, _u, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_uint, C.c_size_t

# name -> (restype, argtypes); must list every symbol include/humanliff_hip.h declares
SIGNATURES = {
    "hl_version": (_i, []),
    "hl_last_error": (C.c_char_p, []),
    "hl_render_mlp_packed_bytes": (_sz, []),
    "hl_render_mlp_pack": (_i, [C.POINTER(RenderMlpParams), _p, _p]),
    "hl_planes_packed_bytes": (_sz, [_i, _i]),
    "hl_planes_pack": (_i, [_p, _i, _i, _p, _p]),
    "hl_render_workspace_bytes": (_sz, [_i64, _i, _i]),
    "hl_render_rays": (_i, [_p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _i64, _i, _i, _u, _p, _p, _p, _p, _p]),
    "hl_render_rays_u_event": (_i, [_p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i, _i, _u, _p, _p, _p, _p, _p]),
    "hl_render_coarse": (_i, [_p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _i64, _i, _p, _p]),
    "hl_render_importance": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i, _i, _p, _p]),
    "hl_deform_points": (_i, [_p, _p, _p, _p, _p, _p, _i, _i64, _p, _p, _p, _p]),
    "hl_deform_rays": (_i, [_p, _p, _p, _p, _p, _i, _i64, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p]),
    "hl_render_eval_points": (_i, [_p, _p, _i, _i, _p, _p, _p, _i64, _i, _p, _p]),
    "hl_render_canonical_workspace_bytes": (_sz, [_i64, _i, _i]),
    "hl_render_rays_canonical": (_i, [_p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _i64, _i, _i, _u, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p]),
    "hl_camera_rays": (_i, [_p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p]),
    "hl_render_eval": (_i, [_p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _i, _i64, _i, _p, _p]),
    "hl_render_eval_products": (_i, [_p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _i, _i64, _i, _u, _p, _p]),
    "hl_render_importance_new": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i, _i, _p, _p]),
    "hl_render_composite": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i, _i, _u, _p, _p, _p, _p]),
    "hl_render_composite_noise": (_i, [_p, _p, _p, _p, _p, _p, _p, _i64, _i, _i, _u, _p, _p, _p, _p]),
    "hl_render_mlp_bwd_packed_bytes": (_sz, []),
    "hl_render_mlp_pack_bwd": (_i, [C.POINTER(RenderMlpParams), _p, _p]),
    "hl_render_train_rows": (None, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "hl_render_eval_acts": (_i, [_p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _i, _i64, _i, _p, _p, _i64, _i64, _p]),
    "hl_render_composite_backward_scratch_bytes": (_sz, [_i64, _i, _i]),
    "hl_render_composite_backward": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i, _i, _u, _p, _p, _p, _i64, _p, _p]),
    "hl_render_eval_points_acts": (_i, [_p, _p, _i, _i, _p, _p, _p, _i64, _i, _p, _p, _i64, _i64, _p]),
    "hl_render_plane_grads_points_scratch_bytes": (_sz, [_i64, _i, _i]),
    "hl_render_plane_grads_points": (_i, [_i, _i, _p, _p, _p, _i64, _i, _i, _p, _i64, _p, _p, _p]),
    "hl_render_weight_grads": (_i, [_p, _i64, _p, _i64, _i64, C.POINTER(RenderMlpParams), _p, _p]),
    "hl_render_weight_grads_scratch_bytes": (_sz, [_i64]),
    "hl_render_plane_grads_scratch_bytes": (_sz, [_i64]),
    "hl_render_mlp_backward": (_i, [_p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _i, _i64, _i, _p, _p, _i64, _i64, _p, _i64, _i64, _p]),
    "hl_render_plane_grads": (_i, [_i, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i64, _i, _i, _p, _i64, _p, _p, _p]),
    "hl_render_fine": (_i, [_p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _i, _i64, _i, _u, _p, _p, _p, _p]),
    "hl_unet_packed_bytes": (_sz, [C.POINTER(UNetCfg)]),
    "hl_unet_create": (_i, [C.POINTER(UNetCfg), _i, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                            _p, _p, C.POINTER(C.c_void_p)]),
    "hl_unet_destroy": (None, [_p]),
    "hl_unet_workspace_bytes": (_sz, [_p, _i, _i, _i]),
    "hl_unet_forward": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p]),
    "hl_unet_set_overlap": (_i, [_p, _i]),
    "hl_unet_set_conv_mode": (_i, [_p, _i]),
    "hl_unet_profile": (_i, [_p, _i]),
    "hl_unet_profile_read": (_i, [_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "hl_unet_profile_read_ex": (_i, [_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "hl_unet_dispatch_census": (_i, [_p, C.POINTER(C.c_int64)]),
    "hl_unet_dispatch_census_ex": (_i, [_p, _p, _i]),
    "hl_unet_profile_dominant": (_i, [_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "hl_diffusion_step": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _i64, _i, _i, _i, _p, _p]),
    "hl_conv2d_nhwc": (_i, [_p, _i, _i, _i, _i, _p, _p, _i, _i, _i, _i, _p, _p, _i, _p, _p, _p, _sz, _p]),
    "hl_conv2d_nhwc_mode": (_i, [_i, _p, _i, _i, _i, _i, _p, _p, _i, _i, _i, _i, _p, _p, _i, _p, _p, _p, _sz, _p]),
    "hl_conv2d_nhwc_gn": (_i, [_i, _p, _i, _i, _i, _i, _p, _p, _i, _i, _i, _i, _p, _p, _i, _p, _p, _p, _p, _p, _p, C.POINTER(C.c_int), _p, _sz, _p]),
    "hl_conv2d_nhwc_bwd_data": (_i, [_i, _p, _i, _i, _i, _i, _p, _i, _i, _i, _i, _i, _p, _i, _p, _sz, _p]),
    "hl_conv2d_wgrad_scratch_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "hl_conv2d_wgrad_nhwc_ws": (_i, [_p, _i, _i, _i, _i, _p, _i, _i, _i, _i, _p, _i, _i, _p, _p, _sz, _p]),
    "hl_conv2d_wgrad_nhwc_ws_mode": (_i, [_i, _p, _i, _i, _i, _i, _p, _i, _i, _i, _i, _p, _i, _i, _p, _p, _sz, _p]),
    "hl_gn_apply_nhwc": (_i, [_p, _i64, _i, _i, _i, _p, _p, _i, _p, _p]),
    "hl_gn_backward_scratch_bytes": (_sz, [_i, _i, _i]),
    "hl_gn_backward_reduce": (_i, [_p, _i64, _p, _i, _i, _i, _p, _p, _i, _p, _p, _sz, _p]),
    "hl_gn_backward_apply": (_i, [_p, _i64, _p, _i, _i, _i, _p, _p, _i, _p, _p, _p, _p, _p, _p]),
    "hl_groupnorm_train_forward": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _i, _p, _p, _p, _p, _p, _sz, _p]),
    "hl_groupnorm_train_backward": (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "hl_upsample2_backward_nhwc": (_i, [_p, _i, _i, _i, _i, _p, _p]),
    "hl_zero_stuff2_nhwc": (_i, [_p, _i, _i, _i, _i, _p, _p]),
    "hl_groupnorm_coef": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "hl_attention_nhwc": (_i, [_p, _i, _i, _i, _i, _p, _p]),
    "hl_attention_nhwc_mode": (_i, [_i, _p, _i, _i, _i, _i, _p, _p]),
    "hl_attention_backward_scratch_bytes": (_sz, [_i, _i, _i, _i]),
    "hl_attention_nhwc_backward": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _p, _sz, _p]),
    "hl_timestep_embedding": (_i, [_p, _p, _i, _i, _p, _p]),
    "hl_mt19937_uniform": (_i, [_p, _i, _p, C.c_int64, _p, _p]),
    "hl_debug_set_h16_min_blocks": (_i, [C.c_long]),
    "hl_debug_set_single_op_scale_source": (_i, [_i]),
    "hl_debug_set_mt19937_piece": (_i, [C.c_int64]),
}


def lib():
    """Load (once) and return the ctypes handle. Raises HipLibraryMissing if it was never built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -m humanliff_amd.build` "
                "(there is no CPU or PyTorch fallback for the hot paths)")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().hl_last_error()
        raise HipCallError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


_PTR_DTYPES = (torch.float32, torch.int64)


def ptr(t, dtype=None):
    """Device pointer of a contiguous CUDA(HIP) tensor, or None.  The pointers of the C ABI are `float*` or `int64_t*`
    (include/humanliff_hip.h) unless a call site names another `dtype` (the uint8 box mask of hl_camera_rays, the int32 vertex
    ids of hl_deform_points); anything else would be reinterpreted bytewise by the kernels, so it is refused here."""
    if t is None:
        return None
    assert t.is_cuda, "humanliff_amd kernels need device tensors (no CPU path)"
    assert t.is_contiguous()
    if (t.dtype != dtype) if dtype is not None else (t.dtype not in _PTR_DTYPES):
        raise TypeError(f"humanliff_amd kernel argument must be {dtype or 'float32 / int64'}, got {t.dtype}")
    return C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    """hipStream_t of torch's current stream on `device` (default: the current device).  Through the raw-stream query (what torch's own
    generated launch code uses: ~0.3 us) rather than torch.cuda.current_stream(), whose Stream object costs ~9 us a call - 5 ms of the
    ~70 ms of Python a UNet training step spends enqueueing its ~1 300 launches."""
    if device is None:
        idx = torch.cuda.current_device()
    else:
        idx = torch.device(device).index if not isinstance(device, int) else device
        if idx is None:
            idx = torch.cuda.current_device()
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(idx))


_SAME_DEVICE = contextlib.nullcontext()


def on(device):
    """Context that makes `device` the current HIP device for the launches inside it: the library launches on the caller's stream,
    and a stream can only be used from its own device.  Nothing to do - and no torch.cuda.device object to build, ~10 us a use - when it
    is the current device already, the case of every launch of a one-process-per-GPU job."""
    idx = device if isinstance(device, int) else torch.device(device).index
    if idx is None or idx == torch.cuda.current_device():
        return _SAME_DEVICE
    return torch.cuda.device(idx)
