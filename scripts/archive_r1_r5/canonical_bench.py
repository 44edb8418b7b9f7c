"""Developer probe: one 512x512 view at 128+128 in canonical space (synthetic 6890-vertex body) - total and per-stage times."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humanliff_amd import synthetic as syn
from humanliff_amd.NeRF import Renderer
dev = torch.device("cuda:0")
V = 6890
cpu_model = syn.smpl_like_model(V, 7)
pose = syn.smpl_like_pose(V, cpu_model, 17, n_points=8)
r = Renderer(use_canonical_space=True, triplane_ch=27, test=True); r.load_state_dict(syn.render_mlp_state(3), strict=False); r = r.to(dev)
r.SMPL_NEUTRAL = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in cpu_model.items()}
planes = syn.triplane(seed=11).to(dev)
centre = pose["vertices"][0].mean(0)
lo, hi = pose["vertices"][0].min(0).values - 0.1, pose["vertices"][0].max(0).values + 0.1
ro, rd, _, _ = syn.orbit_rays(4, 36, 512, 512)
ro = ro + centre
nr, fr = syn.near_far_from_bounds(torch.stack([lo, hi]).double().numpy(), ro.double().numpy(), rd.double().numpy())
ro, rd, nr, fr = ro.to(dev), rd.to(dev), torch.from_numpy(nr).float().to(dev), torch.from_numpy(fr).float().to(dev)
u = torch.rand((512 * 512, 128), device=dev)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = r.render(pose, None, None, ro[None], rd[None], nr[None], fr[None], planes, 128, False, n_samples=128, u=u)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"canonical 512x512 view: {dt * 1e3:.1f} ms -> {512 * 512 / dt / 1e6:.3f} Mrays/s; acc mean {float(out['acc_map'].mean()):.4f}")
