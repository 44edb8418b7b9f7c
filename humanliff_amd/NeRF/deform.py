"""Canonical-space deformation: host mirror of Renderer.deform_target2c
(/root/reference/human_diffusion/NeRF/renderer.py:52-132) on top of hl_deform_points.

The reference blends joint transforms, inverts 3x3 matrices and gathers blend-shape offsets PER QUERY POINT (tens of millions
per view) with stock PyTorch ops, after an external pytorch3d 1-NN search.  All of that depends on the query only through the
nearest body vertex, so here it is folded once per posed subject into a table with one row per vertex (6 890 rows for SMPL) by a
few small tensor ops - plumbing - and the per-point work (1-NN search + applying the row) is one HIP kernel.
"""
import numpy as np
import torch

from .. import _lib


def batch_rodrigues(rot_vecs):
    """(N,3) axis-angle -> (N,3,3); angle = |v + 1e-8| like renderer.py:419-470."""
    angle = torch.linalg.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    k = rot_vecs / angle
    z = torch.zeros_like(k[:, 0])
    K = torch.stack([z, -k[:, 2], k[:, 1], k[:, 2], z, -k[:, 0], -k[:, 1], k[:, 0], z], dim=1).reshape(-1, 3, 3)
    eye = torch.eye(3, dtype=rot_vecs.dtype, device=rot_vecs.device)[None]
    return eye + torch.sin(angle)[:, :, None] * K + (1 - torch.cos(angle))[:, :, None] * torch.matmul(K, K)


def joint_transforms(model, poses, shapes):
    """get_transform_params_torch + get_rigid_transformation_torch (renderer.py:354-417) for one subject: (J,4,4)."""
    sd = model["shapedirs"][..., :shapes.shape[-1]]
    v_shaped = model["v_template"] + (sd * shapes[None, None, :]).sum(-1)
    joints = model["J_regressor"] @ v_shaped
    rot = batch_rodrigues(poses.reshape(-1, 3))
    parents, levels = _tree(model["kintree_table"])
    J = joints.shape[0]
    rel = joints.clone()
    rel[1:] = joints[1:] - joints[parents[1:]]
    local = torch.zeros((J, 4, 4), dtype=joints.dtype, device=joints.device)
    local[:, :3, :3] = rot
    local[:, :3, 3] = rel
    local[:, 3, 3] = 1.0
    # the kinematic chain level by level (all joints of one depth in one batched product): 9 launches for SMPL instead of 23
    G = local.clone()
    for idx, par in levels:
        G[idx] = torch.matmul(G[par], local[idx])
    chain = None
    jh = torch.cat([joints, torch.zeros((J, 1), dtype=joints.dtype, device=joints.device)], dim=1)
    G[:, :, 3] = G[:, :, 3] - (G * jh[:, None, :]).sum(-1)
    return G


_TREE_CACHE = {}
_TABLE_CACHE = {}


def _tree(kintree_table):
    """parents (python list) and, per depth level >= 1, (joint indices, their parents) as index tensors on the table's device."""
    key = (kintree_table.data_ptr(), kintree_table._version, str(kintree_table.device))
    hit = _TREE_CACHE.get(key)
    if hit is None:
        parents = [int(v) for v in kintree_table[0].tolist()]
        depth = [0] * len(parents)
        for i in range(1, len(parents)):
            depth[i] = depth[parents[i]] + 1
        levels = []
        for d in range(1, max(depth) + 1):
            idx = [i for i in range(len(parents)) if depth[i] == d]
            levels.append((torch.tensor(idx, device=kintree_table.device), torch.tensor([parents[i] for i in idx], device=kintree_table.device)))
        if len(_TREE_CACHE) > 16:
            _TREE_CACHE.clear()
        _TREE_CACHE[key] = hit = (parents, levels)
    return hit


def _pose_offsets(model, poses):
    V = model["v_template"].shape[0]
    rot = batch_rodrigues(poses.reshape(-1, 3))
    eye = torch.eye(3, dtype=rot.dtype, device=rot.device)[None]
    feat = (rot[1:] - eye).reshape(1, -1)
    return (feat @ model["posedirs"].reshape(V * 3, -1).t()).reshape(V, 3)


def deform_tables(model, params, t_params, vertices):
    """Per-subject tables of hl_deform_points for batch element 0: verts4 (V,4) = xyz of (vertices - Th) R, table (V,36),
    and the host float32 arrays R (3,3), Th (3,).  `model` holds the SMPL_to_tensor keys (renderer.py:343-352) on the device."""
    dev = vertices.device
    # the same posed subject is rendered view after view: keep the last few tables, keyed by CONTENT (the bytes of the small pose
    # tensors and two checksums of the vertices - storage addresses get reused by new tensors)
    small = torch.cat([t.detach().reshape(-1).to(device=dev, dtype=torch.float32) for t in
                       (params["R"][0], params["Th"][0], params["poses"][0], params["shapes"][0], t_params["poses"][0])])
    vflat = vertices[0].detach().reshape(-1).to(torch.float64)
    ramp = torch.arange(vflat.numel(), device=dev, dtype=torch.float64)
    fp = torch.cat([small.to(torch.float64), vflat.sum()[None], (vflat * ramp).sum()[None]]).cpu().numpy().tobytes()
    key = (fp, model["v_template"].data_ptr(), str(dev))
    hit = _TABLE_CACHE.get(key)
    if hit is not None:
        return hit
    f = lambda t: t.to(device=dev, dtype=torch.float32)  # noqa: E731
    m = {k: (f(v) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in model.items()}
    Rw, Th = f(params["R"][0]), f(params["Th"][0]).reshape(3)
    poses, shapes, tposes = f(params["poses"][0]), f(params["shapes"][0]), f(t_params["poses"][0])
    V = m["v_template"].shape[0]
    A_t = joint_transforms(m, poses, shapes).reshape(-1, 16)
    A_b = joint_transforms(m, tposes, torch.zeros_like(shapes)).reshape(-1, 16)      # mean shape: renderer.py:96
    w = m["weights"]
    At, Ab = (w @ A_t).reshape(V, 4, 4), (w @ A_b).reshape(V, 4, 4)
    sd = m["shapedirs"][..., :shapes.shape[-1]]
    table = torch.zeros((V, 36), dtype=torch.float32, device=dev)
    table[:, 0:3] = At[:, :3, 3]
    table[:, 3:12] = torch.inverse(At[:, :3, :3]).reshape(V, 9)
    table[:, 12:15] = _pose_offsets(m, poses)
    table[:, 15:18] = sd @ shapes
    table[:, 18:21] = _pose_offsets(m, tposes)
    table[:, 21:30] = Ab[:, :3, :3].reshape(V, 9)
    table[:, 30:33] = Ab[:, :3, 3]
    verts4 = torch.zeros((V, 4), dtype=torch.float32, device=dev)
    verts4[:, :3] = (f(vertices[0]) - Th) @ Rw
    out = (verts4.contiguous(), table.contiguous(), Rw.detach().cpu().numpy().astype(np.float32).copy(),
           Th.detach().cpu().numpy().astype(np.float32).copy())
    if len(_TABLE_CACHE) > 8:
        _TABLE_CACHE.clear()
    _TABLE_CACHE[key] = out
    return out


def deform_target2c(model, tp_input, pts, viewdir=None, return_ids=False):
    """Renderer.deform_target2c for use_canonical_space=True, batch 1: pts (1,P,3) [, viewdir (1,P,3)] on the device ->
    (canonical_pts (1,P,3), canonical_viewdir (1,P,3) or None, box_warp = tp_input['t_world_bounds'])."""
    assert pts.dim() == 3 and pts.shape[0] == 1, "batch size 1 (one posed subject per call)"
    if not pts.is_cuda:
        raise RuntimeError("deform_target2c needs CUDA(HIP) tensors; there is no CPU path")
    verts4, table, Rh, Th = deform_tables(model, tp_input["params"], tp_input["t_params"], tp_input["vertices"].to(pts.device))
    P = pts.shape[1]
    p = pts[0].to(torch.float32).contiguous()
    d = viewdir[0].to(torch.float32).contiguous() if viewdir is not None else None
    can = torch.empty((P, 3), dtype=torch.float32, device=pts.device)
    cd = torch.empty((P, 3), dtype=torch.float32, device=pts.device) if d is not None else None
    ids = torch.empty((P,), dtype=torch.int32, device=pts.device) if return_ids else None
    with torch.cuda.device(pts.device):
        _lib.check(_lib.lib().hl_deform_points(_lib.ptr(p), _lib.ptr(d), Rh.ctypes.data, Th.ctypes.data, _lib.ptr(verts4),
                                               _lib.ptr(table), int(verts4.shape[0]), P, _lib.ptr(can), _lib.ptr(cd),
                                               _lib.ptr(ids, torch.int32), _lib.stream_ptr()), "hl_deform_points")
    out = (can[None], cd[None] if cd is not None else None, tp_input["t_world_bounds"])
    return out + (ids,) if return_ids else out
