"""UNet training step at the reference's configuration (README.md:104: production F4 network, microbatch 2, MSE loss):
GaussianDiffusion.training_losses -> backward through the HIP kernels (unet_train.py) -> AdamW step.
    python scripts/unet_train_bench.py [iters] [batch] [twin]      (twin: the same step through the PyTorch-op twin on MIOpen / rocBLAS)
HL_TRAIN_ARITH=fp32|bf16|fp16 selects the arithmetic of the convolutions (unet_train.set_train_arithmetic)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
twin = len(sys.argv) > 3 and sys.argv[3] == "twin"
model, diffusion, _ = bench.build_unet(dev)
model.train()
if os.environ.get('HL_TRAIN_ARITH'):
    from humanliff_amd.improved_diffusion import unet_train as _ut
    _ut.set_train_arithmetic(os.environ['HL_TRAIN_ARITH'])
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.0, fused=os.environ.get('HL_ADAMW_FUSED', '1') == '1')   # one pass over the 497 M parameters instead of PyTorch's ~8 foreach passes
g = torch.Generator(device=dev).manual_seed(0)
x0 = torch.randn((B, 27, 256, 256), device=dev, generator=g).clamp(-1, 1)
xc = torch.zeros_like(x0)
y = torch.zeros((B,), dtype=torch.int64, device=dev)


if twin:      # the torch-autograd statement of the forward lives with the tests since round 5 (tests/unet_autograd_twin.py)
    import functools
    from tests.unet_autograd_twin import forward_autograd
    twin_forward = functools.partial(forward_autograd, model)


def step():
    t = torch.randint(0, 1000, (B,), device=dev, generator=g)
    loss = diffusion.training_losses(twin_forward if twin else model, x0, xc, t, model_kwargs={"y": y})["loss"].mean()
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
    return loss


step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    loss = step()
t_enq = (time.perf_counter() - t0) / iters        # host time to enqueue a step; equal to the wall time = something blocks the host
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(f"host enqueue {t_enq * 1e3:.1f} ms/step of {dt * 1e3:.1f} ms wall")
fl = 3 * 2015.4e9 * B          # forward + backward-data + backward-weights, direct-convolution FLOPs
print(f"UNet training step ({'PyTorch-op twin, MIOpen' if twin else 'HIP kernels, ' + os.environ.get('HL_TRAIN_ARITH', 'fp32')}), batch {B}: {dt * 1e3:.1f} ms/step = {B / dt:.2f} samples/s = {fl / dt / 1e12:.1f} TFLOP/s algorithmic "
      f"(3 x 2015.4 GFLOP per sample); loss {float(loss):.4f}; peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
