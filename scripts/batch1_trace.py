import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda:0")
model, diffusion, _ = bench.build_unet(dev)
B = int(os.environ.get("HL_B", "1"))
g = torch.Generator().manual_seed(0)
x_T = torch.randn((B, 27, 256, 256), generator=g).to(dev)
xc = torch.zeros_like(x_T); y = torch.zeros((B,), dtype=torch.int64, device=dev)
it = diffusion.p_sample_loop_progressive(model, (B, 27, 256, 256), x_cond=xc, noise=x_T, clip_denoised=True, model_kwargs={"y": y}, device=dev)
for _ in range(6): next(it)
torch.cuda.synchronize()
