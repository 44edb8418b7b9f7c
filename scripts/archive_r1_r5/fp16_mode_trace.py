"""Developer script: kernel mix of the fp16-operand inference mode (UNetModel.set_conv_mode('fp16')), batch 4, a few DDPM steps.
   rocprofv3 --kernel-trace --stats -d <dir> -- python scripts/fp16_mode_trace.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
model, diffusion, _ = bench.build_unet(dev)
model.set_conv_mode(os.environ.get("HL_MODE", "fp16"))
B = 4
g = torch.Generator().manual_seed(0)
x_T = torch.randn((B, 27, 256, 256), generator=g).to(dev)
xc = torch.zeros_like(x_T); y = torch.zeros((B,), dtype=torch.int64, device=dev)
it = diffusion.p_sample_loop_progressive(model, (B, 27, 256, 256), x_cond=xc, noise=x_T, clip_denoised=True, model_kwargs={"y": y}, device=dev)
for _ in range(6):
    next(it)
torch.cuda.synchronize()
