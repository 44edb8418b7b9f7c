"""Developer probe: wall time of the production UNet forward at a few batch sizes (HL_B list), 10 forwards each."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda:0")
model, diffusion, sd = bench.build_unet(dev)
if os.environ.get("HL_NO_OVERLAP"):   # one stream: per-kernel times of a trace add up to the forward
    from humanliff_amd import _lib
    x0 = torch.zeros((1, 27, 256, 256), device=dev)
    with torch.no_grad(): model(x0, torch.zeros((1,), dtype=torch.int64, device=dev), x0, y=torch.zeros((1,), dtype=torch.int64, device=dev))
    _lib.check(_lib.lib().hl_unet_set_overlap(model._hip[0], 0))
for B in [int(b) for b in os.environ.get("HL_B", "1,4").split(",")]:
    x = torch.randn((B, 27, 256, 256), device=dev); xc = torch.zeros_like(x)
    t = torch.full((B,), 500, dtype=torch.int64, device=dev); y = torch.zeros((B,), dtype=torch.int64, device=dev)
    with torch.no_grad():
        for _ in range(3): model(x, t, xc, y=y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): model(x, t, xc, y=y)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
    print(f"B={B}: {dt * 1e3:.3f} ms per forward = {B / dt:.2f} samples/s", flush=True)
