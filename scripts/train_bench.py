"""Tri-plane fitting iteration at the reference's training configuration (recon_NeRF/configs/SynBody.txt: batch_size 2 subjects x
n_rand 2048 rays x 128+128 samples, perturb 1, 256x256x27 tri-planes, Adam on MLP + tri-planes, MSE rgb + 0.1 MSE acc).
Prints ms per iteration and the split forward / backward / optimizer.   python scripts/train_bench.py [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humanliff_amd import synthetic as syn
from humanliff_amd.NeRF import Renderer

dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
canonical = len(sys.argv) > 2 and sys.argv[2] == "canonical"     # TightCap fitting: sample points through the body deformation
torch.manual_seed(0)
r = Renderer(use_canonical_space=canonical, triplane_dim=256, triplane_ch=27, test=False)
r.load_state_dict(syn.render_mlp_state(3), strict=False)
r = r.to(dev)
r.uniforms_on_device = os.environ.get('HL_U_DEVICE', '1') == '1'
tri = torch.nn.Parameter((0.1 * torch.randn((2, 4, 3, 9, 256, 256))).to(dev))
opt = torch.optim.Adam([{'params': list(r.parameters()), 'lr': 5e-4}, {'params': [tri], 'lr': 1e-2}], betas=(0.9, 0.999),
                           fused=os.environ.get('HL_ADAMW_FUSED', '1') == '1')
bs, R, N = 2, 2048, 128
ro, rd, nr, fr = syn.orbit_rays(2, 8, 128, 128)
pick = torch.nonzero(fr != 1).flatten()
pick = pick[torch.randperm(pick.numel())[:R]]
ro, rd, nr, fr = (t[pick].to(dev) for t in (ro, rd, nr, fr))
target = torch.rand((bs, R, 3), device=dev)
tp = {"world_bounds": torch.tensor(syn.WORLD_BOUNDS)[None].expand(bs, 2, 3).to(dev)}
if canonical:
    model = syn.smpl_like_model(6890, 7)
    pose = syn.smpl_like_pose(6890, model, 17, n_points=8)
    r.SMPL_NEUTRAL = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in model.items()}
    rep = lambda v: v.expand(bs, *v.shape[1:]).contiguous()  # noqa: E731
    tp = {"params": {k: rep(v) for k, v in pose["params"].items()}, "t_params": {k: rep(v) for k, v in pose["t_params"].items()},
          "vertices": rep(pose["vertices"]), "t_world_bounds": rep(pose["t_world_bounds"])}
    centre = pose["vertices"][0].mean(0)
    lo, hi = pose["vertices"][0].min(0).values - 0.1, pose["vertices"][0].max(0).values + 0.1
    o2, d2, _, _ = syn.orbit_rays(4, 36, 128, 128)
    o2 = o2 + centre
    n2, f2 = syn.near_far_from_bounds(torch.stack([lo, hi]).double().numpy(), o2.double().numpy(), d2.double().numpy())
    n2, f2 = torch.from_numpy(n2).float(), torch.from_numpy(f2).float()
    pick = torch.nonzero(f2 != 1).flatten()
    pick = pick[torch.randperm(pick.numel())[:R]]
    ro, rd, nr, fr = (t[pick].to(dev) for t in (o2, d2, n2, f2))
ids, layer = torch.tensor([0, 1], device=dev), torch.tensor([1, 3], device=dev)   # device index tensors, like the reference's to_cuda batch
t = torch.linspace(0., 1., steps=N, device=dev)
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
tf = tb = to = 0.0
for it in range(iters + 3):
    e0, e1, e2, e3 = ev(), ev(), ev(), ev()
    e0.record()
    z = (nr[:, None] * (1. - t) + fr[:, None] * t)[None].expand(bs, R, N)
    mids = .5 * (z[..., 1:] + z[..., :-1])
    upper, lower = torch.cat([mids, z[..., -1:]], -1), torch.cat([z[..., :1], mids], -1)
    z = lower + (upper - lower) * torch.rand(z.shape, device=dev)
    out = r.render(tp, None, z, ro[None].expand(bs, R, 3), rd[None].expand(bs, R, 3), nr[None, :, None].expand(bs, R, 1),
                   fr[None, :, None].expand(bs, R, 1), tri[ids, layer], N, False)
    loss = ((out["rgb_map"] - target) ** 2).mean() + 0.1 * ((out["acc_map"] - 1.0) ** 2).mean()
    e1.record()
    loss.backward()
    e2.record()
    opt.step()
    opt.zero_grad()
    e3.record()
    torch.cuda.synchronize()
    if it >= 3:
        tf += e0.elapsed_time(e1); tb += e1.elapsed_time(e2); to += e2.elapsed_time(e3)
tot = (tf + tb + to) / iters
pts = bs * R * 2 * N
print(f"fitting iteration: {tot:.2f} ms ({1000 / tot:.1f} it/s; {pts / tot / 1e3:.1f} M sample points/s)  forward {tf / iters:.2f}  backward {tb / iters:.2f}  "
      f"optimizer {to / iters:.2f} ms; loss {float(loss):.4f}")
