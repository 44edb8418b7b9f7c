"""End-to-end example on synthetic weights: layer-conditioned DDIM sampling of several subjects, then orbit renders of every
final tri-plane, sharded over the ranks it is launched with (BASELINE configs[3] / configs[4] at reduced counts).

    python scripts/sample_and_render.py --subjects 2 --layers 2 --ddim 10 --views 4 --res 256
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/sample_and_render.py --subjects 64 ...

Mirrors scripts/triplane_sample_layered.py of the reference: per subject, layer k is sampled with x_cond = the layer k-1 sample
(:124-134); the finished tri-plane is reshaped to (1,3,9,256,256) (:158) and rendered view by view (:159-199); samples / images are
all-gathered at the end (:211-212).  Rays, near/far and the importance uniforms are made on the device (NeRF.render_view).
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from humanliff_amd import distributed as hd, synthetic as syn
from humanliff_amd.NeRF import Renderer, render_view
from humanliff_amd.improved_diffusion.script_util import create_model_and_diffusion


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--subjects", type=int, default=2)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--ddim", type=int, default=10)
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--batch", type=int, default=4, help="subjects sampled together on one GPU")
    ap.add_argument("--float-images", action="store_true", help="gather fp32 images instead of uint8")
    args = ap.parse_args()
    rank, world, dev = hd.init_distributed()
    cfg = dict(bench.F4, timestep_respacing=f"ddim{args.ddim}")
    model, diffusion = create_model_and_diffusion(**cfg)
    keys = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    model.load_state_dict(syn.state_from_shapes(keys, seed=1))
    model = model.to(dev).eval()
    r = Renderer(use_canonical_space=False, triplane_dim=256, triplane_ch=27, smpl_type="smpl", test=True)
    r.load_state_dict(syn.render_mlp_state(3), strict=False)
    r = r.to(dev)
    shape = (27, 256, 256)

    def sample_fn(x_cond, layer, ids):
        g = torch.Generator().manual_seed(1000 * layer + ids[0])
        noise = torch.randn((len(ids),) + shape, generator=g).to(dev)
        y = torch.full((len(ids),), layer, dtype=torch.int64, device=dev)
        return diffusion.ddim_sample_loop(model, (len(ids),) + shape, x_cond=x_cond, noise=noise, clip_denoised=True,
                                          model_kwargs={"y": y}, device=dev)

    tp = {"world_bounds": torch.tensor(syn.WORLD_BOUNDS)[None].to(dev)}
    H = W = args.res

    def render_fn(sid, sample, v):
        planes = sample.clamp(-1, 1).reshape(1, 3, 9, 256, 256)                   # triplane_sample_layered.py:158
        K, c2w, cam = syn.orbit_camera(v, args.views, H, W)
        R = c2w.T.copy(); T = (-R @ cam).reshape(3, 1)
        return render_view(H, W, K, R, T, planes, tp, r, n_samples=128, n_importance=128)[0]

    torch.cuda.synchronize(); t0 = time.perf_counter()
    # sampling, rendering and the gathers (samples: one all-gather; images: per subject, uint8, asynchronous behind the next
    # subject's renders) - humanliff_amd.distributed.sample_and_render, the same code the world-size-2 CPU test drives
    samples, images = hd.sample_and_render(sample_fn, render_fn, args.subjects, args.layers, shape, args.batch, args.views, (H, W, 3), dev,
                                           as_uint8=not args.float_images, images_root=0)   # only rank 0 writes images (:214-219)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    t1 = t0
    if rank == 0:
        steps = args.subjects * args.layers * args.ddim
        rays = args.subjects * args.views * H * W
        print(f"ranks {world}: {args.subjects} subjects x {args.layers} layers x DDIM-{args.ddim} ({steps} denoise steps) + "
              f"{args.subjects * args.views} views {H}x{W} ({rays / 1e6:.1f} Mrays) + gathers in {t2 - t0:.2f} s; "
              f"samples {tuple(samples.shape)} images {tuple(images.shape)} {images.dtype} mean {float(images.float().mean()):.4f}")


if __name__ == "__main__":
    main()
