"""humanliff_amd: MI355X-native (gfx950) hot paths of HumanLiff behind the reference's Python API.

    humanliff_amd.improved_diffusion   GaussianDiffusion / SpacedDiffusion / UNetModel / script_util
    humanliff_amd.NeRF                 Renderer / render (tri-plane volume renderer)

All numerics run in libhumanliff_hip.so (hand-written HIP, include/humanliff_hip.h); PyTorch only
owns device memory, streams and torch.distributed.
"""
__version__ = "0.1.0"
