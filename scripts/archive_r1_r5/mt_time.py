"""Developer probe: duration of hl_mt19937_uniform for one 512x512 view's uniforms (33.5 M numbers), alone and beside a render."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humanliff_amd.NeRF.cpu_rng import rand_like_cpu
dev = torch.device("cuda:0")
torch.manual_seed(1)
for n in (1 << 20, 512 * 512 * 128):
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        u, p = rand_like_cpu([n], dev); p.finish(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"n={n}: {dt * 1e3:.2f} ms = {n / dt / 1e9:.2f} G numbers/s")
t0 = time.perf_counter(); torch.rand(512 * 512, 128); print(f"torch.rand on the host: {(time.perf_counter() - t0) * 1e3:.1f} ms")
