"""Generate golden vectors for per-view ray generation FROM THE REFERENCE.

Runs only in the build container (needs /root/reference).

    python tests/golden/gen_golden_camera.py

Reference entry points exercised (unmodified, imported from /root/reference):
    get_rays        human_diffusion/SynBodyView_datasets.py:316-329
    get_near_far    human_diffusion/SynBodyView_datasets.py:370-403
sample_ray_batch (:405-436) cannot be called (needs cv2.fillPoly and an image); its ray part - the float32 casts
(:422-426) and the near=0 / far=1 fill for rays that miss the box (:428-433) - is restated below.
"""
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, "/root/reference/human_diffusion")
for n in ["cv2", "imageio", "smpl", "smpl.smpl_numpy", "smplx", "smplx.body_models"]:
    sys.modules[n] = types.ModuleType(n)
sys.modules["smpl.smpl_numpy"].SMPL = None
sys.modules["smplx.body_models"].SMPLX = None

import SynBodyView_datasets as D  # noqa: E402

from humanliff_amd import synthetic as syn  # noqa: E402


def ref_rays(H, W, K, R, T, bounds):
    ray_o, ray_d = D.get_rays(H, W, K, R, T)
    ray_o = ray_o.reshape(-1, 3).astype(np.float32)          # :422
    ray_d = ray_d.reshape(-1, 3).astype(np.float32)          # :423
    near, far, mask = D.get_near_far(bounds, ray_o, ray_d)   # :424 (replaces exact zeros of ray_d by 1e-8 in place)
    near_all = np.zeros_like(ray_o[:, 0])                    # :428-433
    far_all = np.ones_like(ray_o[:, 0])
    near_all[mask] = near.astype(np.float32)
    far_all[mask] = far.astype(np.float32)
    return ray_o, ray_d, near_all, far_all, mask


def cameras():
    out = []
    # orbit cameras of the bench / tests (world->camera R, T from the camera-to-world of synthetic.orbit_camera)
    for view, H, W in [(0, 24, 32), (5, 20, 28), (17, 16, 16)]:
        K, c2w, cam = syn.orbit_camera(view, 36, H, W)
        R = c2w.T.copy()
        T = (-R @ cam).reshape(3, 1)
        out.append((f"orbit{view}", H, W, K, R, T, np.asarray(syn.WORLD_BOUNDS, dtype=np.float32)))
    # a close, tilted camera with an off-centre principal point: many rays miss, some graze edges
    rng = np.random.RandomState(3)
    A = rng.randn(3, 3)
    Q, _ = np.linalg.qr(A)
    if np.linalg.det(Q) < 0:
        Q[:, 0] = -Q[:, 0]
    K = np.array([[35.0, 0.0, 11.5], [0.0, 33.0, 9.25], [0.0, 0.0, 1.0]])
    T = np.array([[0.1], [-0.2], [2.5]])
    out.append(("tilted", 18, 26, K, Q, T, np.array([[-0.4, -0.9, -0.3], [0.5, 0.8, 0.35]], dtype=np.float32)))
    # axis-aligned camera: the central column / row of pixels has exactly zero direction components
    K = np.array([[16.0, 0.0, 8.0], [0.0, 16.0, 6.0], [0.0, 0.0, 1.0]])
    out.append(("axis", 12, 16, K, np.eye(3), np.array([[0.0], [0.0], [3.0]]), np.array([[-1, -1, -1], [1, 1, 1]], dtype=np.float32)))
    return out


def main():
    data = {}
    names = []
    for name, H, W, K, R, T, bounds in cameras():
        ro, rd, near, far, mask = ref_rays(H, W, K.copy(), R.copy(), T.copy(), bounds.copy())
        names.append(name)
        data[f"{name}_HW"] = np.array([H, W])
        data[f"{name}_K"], data[f"{name}_R"], data[f"{name}_T"], data[f"{name}_bounds"] = K, R, T, bounds
        data[f"{name}_rays_o"], data[f"{name}_rays_d"] = ro, rd
        data[f"{name}_near"], data[f"{name}_far"], data[f"{name}_mask"] = near, far, mask
        print(name, H, W, "hit", int(mask.sum()), "of", mask.size, "zero-dir fixes", int((rd == np.float32(1e-8)).sum()))
    data["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "camera_rays.npz"), **data)


if __name__ == "__main__":
    main()
