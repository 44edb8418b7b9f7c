"""Developer probe: k_conv time vs batch (rounds of workgroups) and vs Cin (K length) at 256x256, Cout=192."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
CONV_MODE = int(os.environ.get("HL_SWEEP_MODE", "0"))   # 0 exact fp32, 1 bf16x3 emulation
from humanliff_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
def run(N, C, H, W, Co, ks, mode=2, reps=4):
    x = torch.randn((N, H, W, C), device=dev); w = torch.randn((Co, C, ks, ks), device=dev) * 0.02; b = torch.randn(Co, device=dev)
    cA = torch.rand((N, C), device=dev) + 0.5; cB = torch.randn((N, C), device=dev) * 0.1
    out = torch.empty((N, H, W, Co), device=dev); scratch = torch.empty(((Co + 63) // 64 * 64) * C * ks * ks * 5 + 256 + N * H * W * C + (4 << 20), device=dev)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for r in range(reps + 1):
        if r == 1: evs[0].record()
        _lib.check(L.hl_conv2d_nhwc_mode(CONV_MODE, _lib.ptr(x), N, H, W, C, _lib.ptr(w), _lib.ptr(b), Co, ks, 1, 0, _lib.ptr(cA) if mode else None,
                                    _lib.ptr(cB) if mode else None, 1 if mode == 2 else 0, None, _lib.ptr(out), _lib.ptr(scratch), scratch.numel() * 4, _lib.stream_ptr()))
    evs[1].record(); torch.cuda.synchronize()
    ms = evs[0].elapsed_time(evs[1]) / reps
    fl = 2.0 * N * H * W * Co * C * ks * ks
    print(f"N{N} C{C} {H}x{W}->{Co} k{ks} mode{mode}: {ms*1e3:8.1f} us  {fl/ms/1e9:6.1f} TF/s  blocks {N*H*W//128*(Co//96)}", flush=True)
for N in (1, 2, 3, 4, 8):
    run(N, 192, 256, 256, 192, 3)
for C in (48, 96, 192, 384, 768):
    run(4, C, 256, 256, 192, 3)
run(4, 192, 256, 256, 192, 3, mode=0)
run(4, 192, 256, 256, 192, 1, mode=0)
run(4, 192, 256, 256, 96, 3)
run(4, 192, 256, 256, 384, 3)
