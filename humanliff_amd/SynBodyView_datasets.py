"""Per-view ray generation on the GPU - the ray part of the reference's dataset module
(/root/reference/human_diffusion/SynBodyView_datasets.py), SURVEY.md 8(f) rank 2.

The reference builds every view's rays with numpy on the host (get_rays :316-329, get_near_far :370-403, called from
sample_ray_batch :405-436) and uploads 6.3 MB per 512x512 view; here one kernel writes the same float32 arrays directly
in HBM.  Image / mask handling of sample_ray_batch (cv2.fillPoly bound mask, rgb) is dataset code and stays out.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _f64(a, shape):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(shape))
    return a, a.ctypes.data_as(C.c_void_p)


def camera_rays(H, W, K, R, T, bounds, device=None, return_mask=True):
    """Rays of one pinhole view, as sample_ray_batch returns them (:422-433):
    rays_o, rays_d (H*W,3) float32, near, far (H*W) float32, mask_at_box (H*W) bool - device tensors.

    K (3,3) intrinsics, R (3,3) / T (3,1) world->camera extrinsics, bounds (2,3) world_bounds; any array-likes.
    Like the reference, exact zeros of rays_d come back as 1e-8 (get_near_far writes them in place, :373) and rays
    that do not cross the 0.01-padded box exactly twice get near=0, far=1.
    """
    assert int(H) > 0 and int(W) > 0
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    Ki, pKi = _f64(np.linalg.inv(np.asarray(K, dtype=np.float64)), (3, 3))   # :324
    Rm, pR = _f64(R, (3, 3))
    Tm, pT = _f64(T, (3,))
    Bm, pB = _f64(bounds, (2, 3))
    n = int(H) * int(W)
    rays_o = torch.empty((n, 3), device=device, dtype=torch.float32)
    rays_d = torch.empty((n, 3), device=device, dtype=torch.float32)
    near = torch.empty((n,), device=device, dtype=torch.float32)
    far = torch.empty((n,), device=device, dtype=torch.float32)
    mask = torch.empty((n,), device=device, dtype=torch.uint8) if return_mask else None
    with torch.cuda.device(device):
        _lib.check(_lib.lib().hl_camera_rays(pKi, pR, pT, pB, int(H), int(W), _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(near),
                                             _lib.ptr(far), _lib.ptr(mask, torch.uint8) if return_mask else None, _lib.stream_ptr()))
    return rays_o, rays_d, near, far, (mask.bool() if return_mask else None)


def get_rays(H, W, K, R, T, device=None):
    """get_rays (:316-329) -> rays_o, rays_d (H,W,3).  Device float32 (the reference returns host float64 and casts to
    float32 before use, :422-423); bounds are irrelevant here, so a unit box is passed."""
    ro, rd, _, _, _ = camera_rays(H, W, K, R, T, [[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]], device, return_mask=False)
    return ro.view(H, W, 3), rd.view(H, W, 3)
