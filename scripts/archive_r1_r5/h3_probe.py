"""Developer probe: the 3x3 layers of the four levels at batch HL_B (default 4) through the single-op entry point in the default mode, 5 calls each -
run under rocprofv3 --kernel-trace --stats for the per-kernel durations (the fp16x2 kernel against the Winograd kernels: HL_H2_CONV3_MIN_BLOCKS)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humanliff_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
B = int(os.environ.get("HL_B", "4"))
shapes = [(B, 256, 256, 192, 192), (B, 256, 256, 384, 192), (B, 128, 128, 384, 384), (B, 128, 128, 768, 384), (B, 64, 64, 576, 576), (B, 64, 64, 1152, 576), (B, 32, 32, 768, 768), (B, 32, 32, 1536, 768)]
only = os.environ.get("HL_SHAPES")
if only: shapes = [shapes[int(i)] for i in only.split(",")]
for (N, H, W, C, Co) in shapes:
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn((N, H, W, C), device=dev, generator=g); w = torch.randn((Co, C, 3, 3), device=dev, generator=g) / (C * 9) ** 0.5; b = torch.randn(Co, device=dev, generator=g)
    scratch = torch.empty(Co * C * 9 * 6 + 256 + (64 << 20), device=dev)
    out = torch.zeros((N, H, W, Co), device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(6):
        if it == 1: e0.record()
        _lib.check(L.hl_conv2d_nhwc_mode(0, _lib.ptr(x), N, H, W, C, _lib.ptr(w), _lib.ptr(b), Co, 3, 1, 0, None, None, 0, None, _lib.ptr(out), _lib.ptr(scratch), scratch.numel() * 4, _lib.stream_ptr()))
    e1.record(); torch.cuda.synchronize()
    print(f"N{N} {H}x{W} {C}->{Co}: {e0.elapsed_time(e1) / 5 * 1e3:.0f} us per call incl. packing", flush=True)
