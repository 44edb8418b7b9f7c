"""Seeded synthetic inputs for tests, bench and the golden-vector generator.

Everything here is generated on the CPU from `torch.Generator().manual_seed(...)`
so that the container (where the reference can be imported to produce golden
outputs) and the GPU box (where only this repo exists) build bit-identical
inputs.  Conventions follow SURVEY.md §8(d):

* tri-planes  clamp(0.3*randn, -1, 1), shape (1, 3, 9, H, W)
  (reference init/clamp: recon_NeRF/lib/renderer.py:27, run_nerf_batch.py:271-272)
* render MLP  nn.Linear-style U(-1/sqrt(fan_in), 1/sqrt(fan_in)) per tensor
  (state_dict names of human_diffusion/NeRF/renderer.py:29-39)
* cameras     pinhole orbit, un-normalised ray directions
  (human_diffusion/SynBodyView_datasets.py:316-329), near/far from a slab test
  against the bounds padded by 0.01, rays that miss get near=0, far=1
  (SynBodyView_datasets.py:370-403, 428-433)
"""
import math

import numpy as np
import torch

RENDER_MLP_SHAPES = [
    ("pts_linears.0.weight", (128, 27)),
    ("pts_linears.0.bias", (128,)),
    ("pts_linears.1.weight", (128, 128)),
    ("pts_linears.1.bias", (128,)),
    ("pts_linears.2.weight", (128, 155)),
    ("pts_linears.2.bias", (128,)),
    ("feature_linear.weight", (128, 128)),
    ("feature_linear.bias", (128,)),
    ("alpha_linear.weight", (1, 128)),
    ("alpha_linear.bias", (1,)),
    ("views_linear.weight", (64, 155)),
    ("views_linear.bias", (64,)),
    ("rgb_linear.weight", (3, 64)),
    ("rgb_linear.bias", (3,)),
]

WORLD_BOUNDS = [[-1.0, -1.1, -1.0], [1.0, 1.1, 1.0]]


def _gen(seed):
    g = torch.Generator()
    g.manual_seed(int(seed))
    return g


# per-layer powers of two for the WEIGHTS (biases untouched) of the range fixtures render_d / render_e: small and large layers alternate, so the network stays a
# non-trivial function while its weights sit 2^-8 below / 2^4 above nn.Linear's initialisation
LAYER_EXP = {
    "d": {"pts_linears.0": -8, "pts_linears.1": 4, "pts_linears.2": 4, "feature_linear": -8, "views_linear": 4},    # (pre-activations of pts_linears.2 reach ~100: F.softplus's x > 20 branch)
    "e": {"pts_linears.0": 4, "pts_linears.1": -8, "pts_linears.2": 4, "feature_linear": -8, "views_linear": 4},
    "f": {"feature_linear": 8, "views_linear": -8},                                                              # (views_linear small behind a large linear layer)
}


def render_mlp_state(seed=3, gain=1.0, layer_exp=None):
    """Deterministic render-MLP parameters keyed like the reference state_dict.  layer_exp: {layer: k} multiplies that layer's weight by 2^k."""
    out = {}
    fan_in = {}
    for name, shape in RENDER_MLP_SHAPES:
        stem = name.rsplit(".", 1)[0]
        if name.endswith("weight"):
            fan_in[stem] = shape[1]
    for i, (name, shape) in enumerate(RENDER_MLP_SHAPES):
        stem = name.rsplit(".", 1)[0]
        bound = gain / math.sqrt(fan_in[stem])
        t = (torch.rand(shape, generator=_gen(seed * 1000 + i)) * 2 - 1) * bound
        if layer_exp and name.endswith("weight") and stem in layer_exp:
            t = t * (2.0 ** layer_exp[stem])
        out[name] = t.float()
    return out


def triplane(seed=11, H=256, W=256, sigma=0.3, batch=1):
    x = torch.randn((batch, 3, 9, H, W), generator=_gen(seed)) * sigma
    return x.clamp_(-1, 1).float()


def importance_u(n_rays, n_importance, seed=5):
    return torch.rand((n_rays, n_importance), generator=_gen(seed)).float()


def state_from_shapes(keys_shapes, seed, std=0.05, norm_jitter=0.1):
    """Deterministic tensors for an arbitrary (name, shape) list.

    Used for UNet parity: the weights never travel as fixtures, only the key
    list does; both sides rebuild them from the seed.  Biases and weights are
    N(0, std^2) scaled by 1/sqrt(fan_in) for matrices/convs so activations stay
    O(1); GroupNorm affine gets 1 + jitter / jitter.
    """
    out = {}
    for i, (name, shape) in enumerate(keys_shapes):
        g = _gen(seed * 100003 + i)
        shape = tuple(shape)
        t = torch.randn(shape, generator=g)
        if len(shape) >= 2:
            fan = 1
            for d in shape[1:]:
                fan *= d
            t = t * (1.0 / math.sqrt(fan))
        elif "norm" in name or name.endswith("in_layers.0.weight") or name.endswith("out_layers.0.weight") \
                or name.endswith("in_layers.0.bias") or name.endswith("out_layers.0.bias") \
                or name.startswith("out.0."):
            t = t * norm_jitter + (1.0 if name.endswith("weight") else 0.0)
        else:
            t = t * std
        out[name] = t.float()
    return out


def orbit_camera(view, n_views, H, W, radius=3.0, focal_mult=1.1):
    """Pinhole camera looking at the origin from azimuth 2*pi*view/n_views."""
    az = 2.0 * math.pi * view / n_views
    cam = np.array([radius * math.sin(az), 0.0, radius * math.cos(az)], dtype=np.float64)
    fwd = -cam / np.linalg.norm(cam)
    up = np.array([0.0, -1.0, 0.0])  # image y grows downwards
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    # camera-to-world rotation with columns (x=right, y=down, z=fwd)
    c2w = np.stack([right, down, fwd], axis=1)
    f = focal_mult * W
    K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1.0]])
    return K, c2w, cam


def orbit_rays(view, n_views, H, W, bounds=None):
    """rays_o, rays_d (H*W,3), near, far (H*W,) as float32 torch tensors."""
    bounds = np.asarray(WORLD_BOUNDS if bounds is None else bounds, dtype=np.float64)
    K, c2w, cam = orbit_camera(view, n_views, H, W)
    i, j = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64), indexing="xy")
    pix = np.stack([i, j, np.ones_like(i)], axis=-1).reshape(-1, 3)
    d_cam = pix @ np.linalg.inv(K).T
    rays_d = d_cam @ c2w.T  # not normalised, like the reference
    rays_o = np.broadcast_to(cam, rays_d.shape).copy()
    near, far = near_far_from_bounds(bounds, rays_o, rays_d)
    return (torch.from_numpy(rays_o).float(), torch.from_numpy(rays_d).float(),
            torch.from_numpy(near).float(), torch.from_numpy(far).float())


def near_far_from_bounds(bounds, rays_o, rays_d, pad=0.01):
    lo = bounds[0] - pad
    hi = bounds[1] + pad
    d = np.where(rays_d == 0.0, 1e-8, rays_d)
    t0 = (lo[None] - rays_o) / d
    t1 = (hi[None] - rays_o) / d
    tmin = np.minimum(t0, t1).max(axis=1)
    tmax = np.maximum(t0, t1).min(axis=1)
    hit = (tmax > tmin) & (tmax > 0)
    near = np.where(hit, np.maximum(tmin, 0.0), 0.0)
    far = np.where(hit, tmax, 1.0)
    return near, far


# ---- synthetic body model for the canonical-space deformation --------------------------------------
SMPL_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]


def smpl_like_model(n_vertices=6890, seed=21):
    """A body model with the STRUCTURE of SMPL_NEUTRAL (the keys SMPL_to_tensor keeps, NeRF/renderer.py:343-352) and seeded
    random content: the licensed asset is not available here.  24 joints on the SMPL kinematic tree."""
    g = _gen(seed)
    V, J = n_vertices, 24
    box = torch.tensor([0.9, 1.7, 0.4])
    v_template = (torch.rand((V, 3), generator=g) - 0.5) * box
    logits = torch.randn((J, V), generator=g) * 3.0
    J_regressor = torch.softmax(logits, dim=1)
    w = torch.randn((V, J), generator=g) * 2.5
    top = torch.topk(w, 4, dim=1)
    weights = torch.zeros((V, J)).scatter_(1, top.indices, torch.softmax(top.values, dim=1))
    return {
        "v_template": v_template,
        "shapedirs": torch.randn((V, 3, 10), generator=g) * 0.01,
        "posedirs": torch.randn((V, 3, 207), generator=g) * 0.004,
        "J_regressor": J_regressor,
        "kintree_table": torch.tensor([SMPL_PARENTS, list(range(J))], dtype=torch.long),
        "weights": weights,
        "f": torch.zeros((1, 3), dtype=torch.long),
    }


def _rodrigues1(v):
    a = torch.linalg.norm(v) + 1e-12
    k = v / a
    K = torch.tensor([[0.0, -k[2], k[1]], [k[2], 0.0, -k[0]], [-k[1], k[0], 0.0]])
    return torch.eye(3) + torch.sin(a) * K + (1 - torch.cos(a)) * (K @ K)


def smpl_like_pose(n_vertices, model, seed=31, n_points=1024):
    """tp_input pieces for one posed subject (what SynBodyView_datasets.py puts in 'params', 't_params', 'vertices',
    't_world_bounds', :300-306) plus query points near the body and unit view directions, all float32, batch 1."""
    g = _gen(seed)
    V = n_vertices
    poses = torch.randn((1, 72), generator=g) * 0.25
    shapes = torch.randn((1, 10), generator=g) * 0.5
    Rw = _rodrigues1(torch.randn(3, generator=g) * 0.4)[None]
    Th = torch.randn((1, 1, 3), generator=g) * 0.2
    verts_smpl = model["v_template"][None] + torch.randn((1, V, 3), generator=g) * 0.01
    vertices = torch.matmul(verts_smpl, Rw.transpose(1, 2)) + Th          # so that (vertices - Th) @ R is verts_smpl
    big = torch.zeros((1, 72))
    big[0, 5] = 0.5
    big[0, 8] = -0.5
    t_params = {"poses": big, "shapes": torch.zeros((1, 10)), "R": torch.eye(3)[None], "Th": torch.zeros((1, 1, 3))}
    idx = torch.randint(0, V, (n_points,), generator=g)
    pts = vertices[:, idx] + torch.randn((1, n_points, 3), generator=g) * 0.03
    vd = torch.randn((1, n_points, 3), generator=g)
    vd = vd / torch.linalg.norm(vd, dim=-1, keepdim=True)
    return {"params": {"poses": poses, "shapes": shapes, "R": Rw, "Th": Th}, "t_params": t_params, "vertices": vertices,
            "t_world_bounds": torch.tensor([[[-1.0, -1.2, -0.6], [1.0, 1.2, 0.6]]]), "pts": pts, "viewdirs": vd}
