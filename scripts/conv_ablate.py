"""Developer probe: time the main conv shape with ablation builds of the library (scripts/exp_*.so)."""
import sys, os, glob, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humanliff_amd import _lib
dev = torch.device("cuda:0")
N, Cc, H, W, Co = 4, 192, 256, 256, 192
x = torch.randn((N, H, W, Cc), device=dev); w = torch.randn((Co, Cc, 3, 3), device=dev) * 0.02; b = torch.randn(Co, device=dev)
cA = torch.rand((N, Cc), device=dev) + 0.5; cB = torch.randn((N, Cc), device=dev) * 0.1
out = torch.empty((N, H, W, Co), device=dev); scratch = torch.empty(Co * Cc * 9 + 64, device=dev)
libs = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "exp_*.so")))
for rnd in range(2):
    for path in libs:
        L = C.CDLL(path)
        fn = L.hl_conv2d_nhwc; fn.restype = C.c_int; fn.argtypes = _lib.SIGNATURES["hl_conv2d_nhwc"][1]
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for r in range(6):
            if r == 1: evs[0].record()
            fn(_lib.ptr(x), N, H, W, Cc, _lib.ptr(w), _lib.ptr(b), Co, 3, 1, 0, _lib.ptr(cA), _lib.ptr(cB), 1, None, _lib.ptr(out), _lib.ptr(scratch), scratch.numel() * 4, _lib.stream_ptr())
        evs[1].record(); torch.cuda.synchronize()
        ms = evs[0].elapsed_time(evs[1]) / 5
        print(f"{os.path.basename(path):34s} {ms*1e3:8.1f} us {2.0*N*H*W*Co*Cc*9/ms/1e9:6.1f} TF/s", flush=True)
