"""Developer check: per-tensor gradient error of the HIP training path against the reference goldens and oracle intermediates."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_oracle_render_grad import MLP_KEYS, load_grad_case
from tests.test_render_train_gpu import hip_grads
from oracle import render_oracle as ro

dev = torch.device("cuda:0")
for name in ["a", "white"]:
    i, g = load_grad_case(name)
    rgb, acc, d_planes, d_mlp = hip_grads(i, dev)
    ref = torch.from_numpy(g["d_planes"])
    print(name, "planes", float((d_planes - ref).abs().max()), float(ref.abs().max()))
    for q in range(9):
        a, b = d_planes.reshape(9, 3, *d_planes.shape[-2:])[q], ref.reshape(9, 3, *ref.shape[-2:])[q]
        print("   q", q, float((a - b).abs().max()), float(b.abs().max()))
    for k in MLP_KEYS:
        ref = torch.from_numpy(g["d_" + k])
        print("  ", k, float((d_mlp[k] - ref).abs().max()), float(ref.abs().max()))
