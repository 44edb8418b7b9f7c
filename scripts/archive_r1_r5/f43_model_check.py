"""Developer probe: the production UNet (batch 4) with the F(4x4,3x3) convolutions against the other arithmetic modes: forward output
differences (the direct mode is the closest to float64: every product formed once) and denoise-steps/s of the sampling loop per mode."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
B = 4
model, diffusion, sd = bench.build_unet(dev)
g = torch.Generator().manual_seed(99)
xx = torch.randn((B, 27, 256, 256), generator=g).to(dev)
xc = (torch.randn((B, 27, 256, 256), generator=g) * 0.5).to(dev)
tt = torch.tensor([999, 500, 120, 3], dtype=torch.int64, device=dev)
y = torch.zeros((B,), dtype=torch.int64, device=dev)
outs = {}
with torch.no_grad():
    for mode in ("fp32_direct", "fp32_f23", "fp32"):
        model.set_conv_mode(mode)
        outs[mode] = model(xx, tt, xc, y=y).double()
ref = outs["fp32_direct"]
print("output mean |x| %.4f max %.3f" % (ref.abs().mean().item(), ref.abs().max().item()))
for mode in ("fp32_f23", "fp32"):
    d = outs[mode] - ref
    print("%-9s vs direct: max-abs %.3e  rms %.3e" % (mode, d.abs().max().item(), d.pow(2).mean().sqrt().item()))
x_T = torch.randn((B, 27, 256, 256), generator=g).to(dev)
x_cond = torch.zeros((B, 27, 256, 256), device=dev)
for mode in (sys.argv[1:] or ["fp32_f23", "fp32", "fp32_f23", "fp32"]):
    model.set_conv_mode(mode)
    it = diffusion.p_sample_loop_progressive(model, (B, 27, 256, 256), x_cond=x_cond, noise=x_T, clip_denoised=True, model_kwargs={"y": y}, device=dev)
    for _ in range(3):
        next(it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(12):
        out = next(it)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-9s %.2f denoise-steps/s  (%.2f ms/step)" % (mode, B * 12 / dt, dt / 12 * 1e3))
    del it
