"""Developer probe: launch production conv shapes a few times (for rocprofv3 kernel-trace / PMC runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humanliff_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
shapes = [(4, 192, 256, 256, 192, 3), (4, 384, 256, 256, 192, 3), (4, 192, 128, 128, 192, 3), (4, 384, 64, 64, 384, 3),
          (4, 384, 32, 32, 384, 3), (4, 768, 16, 16, 768, 3), (4, 768, 8, 8, 768, 3)]
only = int(sys.argv[1]) if len(sys.argv) > 1 else None
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for i, (N, C, H, W, Co, ks) in enumerate(shapes):
    if only is not None and i != only: continue
    x = torch.randn((N, H, W, C), device=dev); w = torch.randn((Co, C, ks, ks), device=dev) * 0.02; b = torch.randn(Co, device=dev)
    cA = torch.rand((N, C), device=dev) + 0.5; cB = torch.randn((N, C), device=dev) * 0.1
    out = torch.empty((N, H, W, Co), device=dev); scratch = torch.empty(((Co + 63) // 64 * 64) * C * ks * ks + 64 + (16 << 20), device=dev)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for r in range(reps + 1):
        if r == 1: evs[0].record()
        _lib.check(L.hl_conv2d_nhwc(_lib.ptr(x), N, H, W, C, _lib.ptr(w), _lib.ptr(b), Co, ks, 1, 0, _lib.ptr(cA), _lib.ptr(cB), 1, None,
                                    _lib.ptr(out), _lib.ptr(scratch), scratch.numel() * 4, _lib.stream_ptr()))
    evs[1].record(); torch.cuda.synchronize()
    ms = evs[0].elapsed_time(evs[1]) / reps
    fl = 2.0 * N * H * W * Co * C * ks * ks
    print(f"shape {i}: N{N} C{C} {H}x{W} -> {Co} k{ks}: {ms:.3f} ms/call incl. weight pack, {fl/ms/1e9:.1f} TFLOP/s", flush=True)
