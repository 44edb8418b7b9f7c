"""A/B of the fitting iteration (bench.py `fit` leg): subject streams on / off."""
import argparse, json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from humanliff_amd.NeRF import Renderer

a = argparse.Namespace()
dev = torch.device("cuda:0")
for on in (True, False, True):
    Renderer_init = Renderer.__init__

    def init(self, *x, _on=on, **k):
        Renderer_init(self, *x, **k)
        self.subject_streams = _on
    Renderer.__init__ = init
    try:
        r = bench.bench_fit(a, 0, 1, dev, iters=30)
    finally:
        Renderer.__init__ = Renderer_init
    print("FIT streams=%d: %.2f it/s (%.3f ms)  uniforms_on_device %.2f it/s" % (on, r["value"], r["ms_per_iteration"], r["uniforms_on_device"]["value"]))
