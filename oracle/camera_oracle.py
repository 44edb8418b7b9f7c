"""ORACLE (test infrastructure, not product code) - CPU restatement of the per-view ray generation.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(humanliff_amd) never does.

Pinned against golden vectors generated from the reference itself (tests/golden/gen_golden_camera.py ->
tests/golden/camera_rays.npz, checked in tests/test_oracle_camera.py).

Written as explicit per-ray scalar formulas (no matrix products) so that the HIP kernel can follow it term by
term; citations are into /root/reference/human_diffusion/SynBodyView_datasets.py:

  camera_rays   get_rays :316-329 (float64 arithmetic: K, R, T come from json as float64), the float32 casts of
                sample_ray_batch :422-423, get_near_far :370-403 (float64 on the float32 rays; exact zeros of ray_d
                are replaced by float32(1e-8) IN PLACE and so also reach the caller; |ray_d| is a float32 norm), and
                the near=0 / far=1 fill for rays that do not cross the padded box exactly twice :428-433.
"""
import numpy as np


def camera_rays(H, W, K, R, T, bounds):
    """-> rays_o (H*W,3) f32, rays_d (H*W,3) f32, near (H*W) f32, far (H*W) f32, mask_at_box (H*W) bool."""
    K = np.asarray(K, dtype=np.float64)
    R = np.asarray(R, dtype=np.float64)
    T = np.asarray(T, dtype=np.float64).reshape(3)
    Ki = np.linalg.inv(K)
    o64 = -np.array([R[0, c] * T[0] + R[1, c] * T[1] + R[2, c] * T[2] for c in range(3)])       # -(R^T T)
    x = np.tile(np.arange(W, dtype=np.float64), H)
    y = np.repeat(np.arange(H, dtype=np.float64), W)
    pc = [x * Ki[c, 0] + y * Ki[c, 1] + Ki[c, 2] for c in range(3)]                                # xy1 . inv(K)^T
    q = [pc[c] - T[c] for c in range(3)]
    pw = [q[0] * R[0, c] + q[1] * R[1, c] + q[2] * R[2, c] for c in range(3)]                      # (pc - T) . R
    rays_d = np.stack([pw[c] - o64[c] for c in range(3)], axis=1).astype(np.float32)
    rays_o = np.broadcast_to(o64, rays_d.shape).astype(np.float32)

    b = np.asarray(bounds).astype(np.float64) + np.array([-0.01, 0.01])[:, None]                   # :372
    rays_d[rays_d == 0.0] = np.float32(1e-8)                                                       # :373
    o = rays_o.astype(np.float64)
    d = rays_d.astype(np.float64)
    n = rays_d.shape[0]
    cnt = np.zeros(n, dtype=np.int64)
    dist = np.zeros((n, 2))
    eps = 1e-6
    norm32 = np.sqrt((rays_d[:, 0] * rays_d[:, 0] + rays_d[:, 1] * rays_d[:, 1]) + rays_d[:, 2] * rays_d[:, 2])  # f32 :395
    for k in range(6):                                   # planes in the order min_x, min_y, min_z, max_x, max_y, max_z
        side, ax = divmod(k, 3)
        t = (b[side, ax] - o[:, ax]) / d[:, ax]                                                    # :376
        p = t[:, None] * d + o                                                                     # :378
        inside = np.ones(n, dtype=bool)
        for c in range(3):
            inside &= (p[:, c] >= b[0, c] - eps) & (p[:, c] <= b[1, c] + eps)                      # :382-387
        e = p - o
        r = np.sqrt((e[:, 0] * e[:, 0] + e[:, 1] * e[:, 1]) + e[:, 2] * e[:, 2]) / norm32          # :396-397
        first = inside & (cnt == 0)
        second = inside & (cnt == 1)
        dist[first, 0] = r[first]
        dist[second, 1] = r[second]
        cnt += inside
    mask = cnt == 2                                                                                # :389
    near = np.where(mask, np.minimum(dist[:, 0], dist[:, 1]), 0.0).astype(np.float32)              # :398, 428-433
    far = np.where(mask, np.maximum(dist[:, 0], dist[:, 1]), 1.0).astype(np.float32)
    return rays_o, rays_d, near, far, mask
