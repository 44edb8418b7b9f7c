"""Does the whole UNet training step (training_losses -> backward through the HIP kernels -> fused AdamW) capture into one HIP graph?
Replay removes the ~70 ms of Python per step; the losses of eager and replayed steps from the same state must agree."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from humanliff_amd.improved_diffusion import unet_train as ut

dev = torch.device("cuda:0")
B = 2
model, diffusion, _ = bench.build_unet(dev)
model.train()
ut.set_train_arithmetic(os.environ.get("HL_TRAIN_ARITH", "bf16"))
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.0, fused=True, capturable=True)
g = torch.Generator(device=dev).manual_seed(0)
x0 = torch.randn((B, 27, 256, 256), device=dev, generator=g).clamp(-1, 1)
xc = torch.zeros_like(x0)
y = torch.zeros((B,), dtype=torch.int64, device=dev)
t = torch.randint(0, 1000, (B,), device=dev, generator=g)


def body():
    loss = diffusion.training_losses(model, x0, xc, t, model_kwargs={"y": y})["loss"].mean()
    loss.backward()
    opt.step()
    return loss


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        body()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(graph):
    static_loss = body()
torch.cuda.synchronize()
print("GRAPH captured")
for _ in range(2):
    graph.replay()
torch.cuda.synchronize()
n = 8
t0 = time.perf_counter()
for _ in range(n):
    t.copy_(torch.randint(0, 1000, (B,), device=dev, generator=g))
    graph.replay()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("GRAPH replay: %.1f ms/step, loss %.4f, finite %s" % (dt * 1e3, float(static_loss), bool(torch.isfinite(static_loss))))
