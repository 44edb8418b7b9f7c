"""Developer probe: host-side enqueue time of one production UNet forward (batch 4) vs its GPU time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda:0")
model, diffusion, sd = bench.build_unet(dev)
B = 4
x = torch.randn((B, 27, 256, 256), device=dev); xc = torch.zeros_like(x)
t = torch.full((B,), 500, dtype=torch.int64, device=dev); y = torch.zeros((B,), dtype=torch.int64, device=dev)
with torch.no_grad():
    for _ in range(3): model(x, t, xc, y=y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): model(x, t, xc, y=y)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / 10:.2f} ms per forward; GPU-complete {1e3 * (t2 - t0) / 10:.2f} ms per forward")
from humanliff_amd import _lib
_lib.check(_lib.lib().hl_unet_set_overlap(model._hip[0], 0))
with torch.no_grad():
    for _ in range(2): model(x, t, xc, y=y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): model(x, t, xc, y=y)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
print(f"single stream: enqueue {1e3 * (t1 - t0) / 10:.2f} ms per forward")
import ctypes as C
L = _lib.lib()
nb = L.hl_unet_workspace_bytes(model._hip[0], 4, 256, 256)
t0 = time.perf_counter()
for _ in range(10): L.hl_unet_workspace_bytes(model._hip[0], 4, 256, 256)
print(f"dry structure walk: {1e3 * (time.perf_counter() - t0) / 10:.3f} ms")
_lib.check(_lib.lib().hl_unet_set_overlap(model._hip[0], 1))
with torch.no_grad():
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); model(x, t, xc, y=y); ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
print("one forward into an empty queue: enqueue ms", [round(1e3 * v, 2) for v in ts])
