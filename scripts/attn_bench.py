"""Developer probe: time hl_attention_nhwc at the UNet's three attention levels (batch 4)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humanliff_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
for (N, T, C, heads) in [(4, 1024, 384, 4), (4, 256, 768, 4), (4, 64, 768, 4)]:
    qkv = torch.randn((N, T, 3 * C), device=dev); out = torch.empty((N, T, C), device=dev)
    for _ in range(3):
        _lib.check(L.hl_attention_nhwc(_lib.ptr(qkv), N, T, C, heads, _lib.ptr(out), _lib.stream_ptr()))
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        _lib.check(L.hl_attention_nhwc(_lib.ptr(qkv), N, T, C, heads, _lib.ptr(out), _lib.stream_ptr()))
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    fl = 4.0 * N * T * T * C
    print(f"N{N} T{T} C{C} h{heads}: {us:8.1f} us  {fl / us / 1e6:6.1f} TF/s")
