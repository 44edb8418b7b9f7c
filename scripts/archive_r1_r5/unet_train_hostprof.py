"""cProfile of the host side of one UNet training step (where the ~80 ms of enqueue time go)."""
import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from humanliff_amd.improved_diffusion import unet_train as ut

dev = torch.device("cuda:0")
model, diffusion, _ = bench.build_unet(dev)
model.train()
ut.set_train_arithmetic(os.environ.get('HL_TRAIN_ARITH', 'bf16'))
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.0, fused=True)
g = torch.Generator(device=dev).manual_seed(0)
B = 2
x0 = torch.randn((B, 27, 256, 256), device=dev, generator=g).clamp(-1, 1)
xc = torch.zeros_like(x0)
y = torch.zeros((B,), dtype=torch.int64, device=dev)


def step():
    t = torch.randint(0, 1000, (B,), device=dev, generator=g)
    loss = diffusion.training_losses(model, x0, xc, t, model_kwargs={"y": y})["loss"].mean()
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


for _ in range(2):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
st.sort_stats("cumulative").print_stats(45)
