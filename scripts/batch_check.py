"""Developer probe: UNet forward at batch 8 and 16 (the 2 GiB buffer-offset fallback), one sample cross-checked at batch 1."""
import sys; sys.path.insert(0, '/root/repo')
import torch, bench
dev = torch.device("cuda:0")
model, diffusion, sd = bench.build_unet(dev)
g = torch.Generator().manual_seed(3)
for B in (8, 16):
    x = torch.randn((B, 27, 256, 256), generator=g).to(dev); xc = torch.randn((B, 27, 256, 256), generator=g).to(dev) * 0.3
    t = torch.randint(0, 1000, (B,), generator=g).to(dev); y = torch.randint(0, 5, (B,), generator=g).to(dev)
    with torch.no_grad():
        out = model(x, t, xc, y=y)
        one = model(x[3:4], t[3:4], xc[3:4], y=y[3:4])
    torch.cuda.synchronize()
    print(B, bool(torch.isfinite(out).all()), float((out[3] - one[0]).abs().max()), float(out.abs().mean()), torch.cuda.max_memory_allocated() / 2**30)
