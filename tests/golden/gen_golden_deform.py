"""Generate golden vectors for the canonical-space deformation FROM THE REFERENCE.

Runs only in the build container (needs /root/reference).

    python tests/golden/gen_golden_deform.py

Reference entry point exercised (unmodified, imported from /root/reference):
    NeRF.renderer.Renderer.deform_target2c        human_diffusion/NeRF/renderer.py:114-132
      -> deform_target2c_op :52-112, get_transform_params_torch :354-384, get_rigid_transformation_torch :386-417,
         batch_rodrigues_torch :419-438, batch_rodrigues :440-470
What is NOT the reference here, and why:
  * the SMPL body model: assets/SMPL_NEUTRAL.pkl is licensed data that is not in the repository, so a synthetic model of the
    same structure (v_template, shapedirs, posedirs, J_regressor, kintree_table, weights) is drawn from a seed
    (humanliff_amd.synthetic.smpl_like_model); the arithmetic applied to it is the reference's;
  * pytorch3d.ops.knn.knn_points (external dependency, pytorch3d, README.md:41): stubbed with its published semantics for K=1 -
    squared Euclidean distances, index of the nearest point, first index on ties;
  * `.cuda()` calls inside the reference functions are made no-ops (this container has no GPU).
"""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, "/root/reference/human_diffusion")

for n in ["mcubes", "cv2", "pytorch3d", "pytorch3d.ops", "pytorch3d.ops.knn"]:
    sys.modules[n] = types.ModuleType(n)


def knn_points(p1, p2, K=1):
    assert K == 1
    d = ((p1[:, :, None, :].double() - p2[:, None, :, :].double()) ** 2).sum(-1)   # (bs, P1, P2)
    dist, idx = d.min(dim=-1, keepdim=True)
    return dist.float(), idx, None


sys.modules["pytorch3d.ops.knn"].knn_points = knn_points

from NeRF import renderer as R  # noqa: E402

torch.autograd.set_detect_anomaly(False)
R.read_pickle = lambda p: {}
R.SMPL_to_tensor = lambda params, device: {"f": None}
torch.cuda.current_device = lambda: 0
torch.Tensor.cuda = lambda self, *a, **k: self

from humanliff_amd import synthetic as syn  # noqa: E402


def main():
    data = {}
    names = []
    for name, V, P, seed in [("small", 300, 700, 1), ("mid", 1200, 2048, 2)]:
        model = syn.smpl_like_model(V, seed)
        pose = syn.smpl_like_pose(V, model, seed + 10, n_points=P)
        r = R.Renderer(use_canonical_space=False, triplane_dim=256, triplane_ch=27, smpl_type="smpl", test=True)
        r.use_canonical_space = True
        r.SMPL_NEUTRAL = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in model.items()}
        tp = {"params": {k: v.clone() for k, v in pose["params"].items()},
              "t_params": {k: v.clone() for k, v in pose["t_params"].items()},
              "vertices": pose["vertices"].clone(), "t_world_bounds": pose["t_world_bounds"].clone()}
        with torch.no_grad():
            cp, cd, box = r.deform_target2c(tp, pose["pts"].clone(), pose["viewdirs"].clone())
            cp2, none, _ = r.deform_target2c(tp, pose["pts"].clone())
        assert none is None and torch.equal(cp, cp2)
        names.append(name)
        data[f"{name}_VP"] = np.array([V, P, seed])
        data[f"{name}_can_pts"] = cp.numpy()
        data[f"{name}_can_dirs"] = cd.numpy()
        data[f"{name}_box"] = box.numpy()
        print(name, "V", V, "P", P, "can_pts mean-abs", float(cp.abs().mean()), "dirs mean-abs", float(cd.abs().mean()))
    data["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "deform.npz"), **data)


if __name__ == "__main__":
    main()
