"""Where the fitting iteration spends host time: enqueue time against wall time, and the CPU uniforms draw."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humanliff_amd import synthetic as syn
from humanliff_amd.NeRF import Renderer

dev = torch.device("cuda:0")
torch.manual_seed(0)
r = Renderer(use_canonical_space=False, triplane_dim=256, triplane_ch=27, test=False)
r.load_state_dict(syn.render_mlp_state(3), strict=False)
r = r.to(dev)
tri = torch.nn.Parameter((0.1 * torch.randn((2, 4, 3, 9, 256, 256))).to(dev))
opt = torch.optim.Adam([{'params': list(r.parameters()), 'lr': 5e-4}, {'params': [tri], 'lr': 1e-2}], betas=(0.9, 0.999), fused=True)
bs, R, N = 2, 2048, 128
ro, rd, nr, fr = syn.orbit_rays(2, 8, 128, 128)
pick = torch.nonzero(fr != 1).flatten()
pick = pick[torch.randperm(pick.numel())[:R]]
ro, rd, nr, fr = (t[pick].to(dev) for t in (ro, rd, nr, fr))
target = torch.rand((bs, R, 3), device=dev)
tp = {"world_bounds": torch.tensor(syn.WORLD_BOUNDS)[None].expand(bs, 2, 3).to(dev)}
ids, layer = torch.tensor([0, 1], device=dev), torch.tensor([1, 3], device=dev)   # the reference's loop indexes with the batch's device tensors (to_cuda, run_nerf_batch.py:233)
t = torch.linspace(0., 1., steps=N, device=dev)
T = {}


def one():
    z = (nr[:, None] * (1. - t) + fr[:, None] * t)[None].expand(bs, R, N)
    mids = .5 * (z[..., 1:] + z[..., :-1])
    upper, lower = torch.cat([mids, z[..., -1:]], -1), torch.cat([z[..., :1], mids], -1)
    z = lower + (upper - lower) * torch.rand(z.shape, device=dev)
    t0 = time.perf_counter()
    out = r.render(tp, None, z, ro[None].expand(bs, R, 3), rd[None].expand(bs, R, 3), nr[None, :, None].expand(bs, R, 1),
                   fr[None, :, None].expand(bs, R, 1), tri[ids, layer], N, False)
    t1 = time.perf_counter()
    loss = ((out["rgb_map"] - target) ** 2).mean() + 0.1 * ((out["acc_map"] - 1.0) ** 2).mean()
    loss.backward()
    t2 = time.perf_counter()
    opt.step()
    opt.zero_grad()
    t3 = time.perf_counter()
    T["fwd"] = T.get("fwd", 0) + t1 - t0
    T["bwd"] = T.get("bwd", 0) + t2 - t1
    T["opt"] = T.get("opt", 0) + t3 - t2
    T["pre"] = T.get("pre", 0) + t0
    return loss


for dev_u in (False, True):
    r.uniforms_on_device = dev_u
    for _ in range(5):
        one()
    torch.cuda.synchronize()
    T.clear()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        one()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("HOST uniforms_on_device=%d: enqueue %.3f ms/it, wall %.3f ms/it; host fwd %.3f bwd %.3f opt %.3f" %
          (dev_u, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, T["fwd"] / n * 1e3, T["bwd"] / n * 1e3, T["opt"] / n * 1e3))
t0 = time.perf_counter()
for _ in range(20):
    u = torch.rand([bs * R, N])
print("HOST torch.rand CPU draw: %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
t0 = time.perf_counter()
for _ in range(20):
    u = torch.rand([bs * R, N], pin_memory=True)
print("HOST torch.rand CPU draw pinned: %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))

# ---- host time per C-ABI call inside one iteration (no device synchronisation anywhere) ----
from humanliff_amd import _lib
real = _lib.lib()
acc = {}


class Proxy:
    def __getattr__(self, name):
        f = getattr(real, name)

        def g(*a):
            t0 = time.perf_counter()
            rc = f(*a)
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
            return rc
        return g


_lib.lib = lambda: Proxy()
r.uniforms_on_device = True
for _ in range(3):
    one()
torch.cuda.synchronize()
acc.clear()
for _ in range(10):
    one()
torch.cuda.synchronize()
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("HOST call %-44s %.3f ms/it" % (k, v / 10 * 1e3))
