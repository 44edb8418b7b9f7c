import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_oracle_diffusion import _stub
from humanliff_amd.improved_diffusion.script_util import create_gaussian_diffusion
from oracle import diffusion_oracle as do
dev = torch.device("cuda:0")
g = np.load("tests/golden/diffusion_steps.npz")
gen = torch.Generator().manual_seed(7)
x = torch.randn((3, 27, 8, 8), generator=gen); xc = torch.randn((3, 27, 8, 8), generator=gen) * 0.5
noise = torch.randn((3, 27, 8, 8), generator=gen); y = torch.tensor([0, 3, 1])
d = create_gaussian_diffusion(steps=1000, timestep_respacing="")
t = torch.from_numpy(g["step_full_1_t"]).long()
model = lambda xx, tt, xcond, y=None: _stub(xx.cpu(), tt.cpu(), xcond.cpu(), y.cpu()).to(dev)
torch.randn_like = lambda ref: noise.to(ref.device)
ps = d.p_sample(model, x.to(dev), xc.to(dev), t.to(dev), clip_denoised=True, model_kwargs={"y": y.to(dev)})
want = torch.from_numpy(g["step_full_1_p_sample"])
diff = (ps["sample"].cpu() - want).abs()
print("p_sample maxdiff", diff.max().item(), "per batch", diff.amax(dim=(1,2,3)))
print("x0 maxdiff", (ps["pred_xstart"].cpu() - torch.from_numpy(g["step_full_1_p_x0"])).abs().amax(dim=(1,2,3)))
tab = d._table("ddpm", dev).cpu()
s = do.Schedule(do.linear_betas(1000), list(range(1000)))
print("tab rows", tab[t], )
print("oracle", torch.from_numpy(s.sqrt_recip)[t].float(), torch.from_numpy(s.sqrt_recipm1)[t].float(), torch.from_numpy(s.coef1)[t].float(), torch.from_numpy(s.coef2)[t].float(), torch.exp(0.5*torch.from_numpy(np.log(s.fixed_large_var))[t].float()))
