#!/usr/bin/env python
"""bench.py - headline benchmark of the two hot paths on MI355X.

    python bench.py --gpus 1 --steps 8 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Primary metric (BASELINE.json configs[1]): denoise-steps/sec of the production 256x256x27 tri-plane UNet
(497M params) inside the 1000-step DDPM p_sample loop at batch 4 per GPU.  A "step" is one p_sample call:
UNet forward + fused posterior update + noise draw for the whole batch; value = ranks * 4 * K / seconds.
Secondary metric (configs[2]) in the "render" object: Mrays/sec of the tri-plane renderer, 512x512 views,
128 coarse + 128 importance samples per ray.  Weak scaling: every rank runs its own subjects / views; the
only collective is the final all-gather of samples and images (north star), inside the timed region.
"fit" object (not a BASELINE metric): tri-plane fitting iterations/sec at the reference's training
configuration (2 subjects x 2048 rays x 128+128 samples; forward, HIP backward, Adam).

Prints ONE JSON line on rank 0.  `roofline` is measured live with HIP events (per-kernel-category events
inside hl_unet_forward for the conv kernels; torch events around the stage launches for the ray-march
kernel) on one extra pass after the timed region; `cpu_baseline` times the oracle (a PyTorch-CPU
restatement of the reference's algorithm, kind "port") on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from bench_legs import *          # noqa: E402,F401,F403  (the legs; tests and scripts also reach build_unet / F4 through this module)
from bench_legs import UNET_GFLOP_PER_SAMPLE_STEP, dist_setup, bench_unet, collective_probe, e2e_chain, ddim50_parity, cpu_baseline_unet, e2e_slice, bench_train, \
    bench_render, bench_fit, cpu_baseline_render, cpu_baseline_fit   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4, help="subjects per GPU in the denoise loop (configs[1]: 4)")
    ap.add_argument("--views", type=int, default=36, help="512x512 views per GPU in the render leg (configs[2]: the whole 36-view orbit)")
    ap.add_argument("--sustained-steps", type=int, default=200,
                    help="extra untimed-for-`value` steps of the same loop after the timed region: sustained steps/s + shader clock (N=1 only; 0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity legs (end-to-end chain vs the reference's vectors, HIP vs oracle samples)")
    ap.add_argument("--no-batch-sweep", action="store_true", help="skip the B = 1 / 8 / 4 eager-vs-graph legs of the denoise loop")
    ap.add_argument("--no-render", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the per-GPU slice of configs[3]/[4] (4 layers x DDIM-50 -> 185 views 512x512)")
    ap.add_argument("--e2e-views", type=int, default=185)
    ap.add_argument("--e2e-ddim", type=int, default=50)
    ap.add_argument("--e2e-layers", type=int, default=4)
    ap.add_argument("--e2e-subjects", type=int, default=8, help="subjects per GPU in the e2e slice (configs[3]/[4]: 64 subjects over 8 GPUs)")
    ap.add_argument("--e2e-batch", type=int, default=8, help="subjects sampled at a time in the e2e slice")
    ap.add_argument("--no-train", action="store_true", help="skip the UNet training-step leg (SURVEY 8(f) rank 4)")
    ap.add_argument("--no-fit", action="store_true", help="skip the tri-plane fitting leg (SURVEY 8(f) rank 4)")
    ap.add_argument("--no-bf16x3-leg", action="store_true", help="skip the extra measurement of the opt-in bf16x3 conv mode")
    ap.add_argument("--no-overlap", action="store_true",
                    help="issue the control encoder on the caller's stream instead of the side stream (used for the "
                         "per-kernel rocprof trace: concurrent kernels stretch each other's durations)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become one - one rank per GPU under torch.distributed.run on this node (RCCL over
        # xGMI), same arguments; the ranks' output (rank 0 prints the JSON line) passes straight through.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this driver (RCCL / tensor sharing across processes)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    from humanliff_amd import _lib
    _lib.lib()   # fail loudly if the HIP library was not built
    rank, local, world = dist_setup(args.gpus)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    dev = torch.device("cuda", local)

    secs, roof, sd, model = bench_unet(args, rank, world, dev)
    value = world * args.batch * args.steps / secs
    rccl = collective_probe(args, rank, world, dev) if world > 1 else None
    parity = None
    if rank == 0 and world == 1 and not args.no_parity:
        chain = e2e_chain(model, dev)
        parity = {"psnr_db": min(l["image_psnr_db"] for l in chain["layers"]), "max_abs": max(l["image_max_abs"] for l in chain["layers"]),
                  "triplane_psnr_db": min(l["triplane_psnr_db"] for l in chain["layers"]),
                  "triplane_max_abs": max(l["triplane_max_abs"] for l in chain["layers"]), "end_to_end": chain,
                  "ddim50_vs_reference": ddim50_parity(model, dev)}
    cpu = step_parity = None
    threads = min(len(os.sched_getaffinity(0)), 32)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # PyTorch-CPU stops scaling (and then collapses) beyond ~32 threads on this path: measured on the
        # MI355X host (256 logical CPUs) 3x3 conv 192->192@256^2: 49/47/40/90/208 ms at 8/16/32/64/128 threads
        cpu, step_parity = cpu_baseline_unet(sd, threads, None if args.no_parity else model, dev)
        if parity is not None:
            parity["denoise_steps_vs_oracle"] = step_parity
    e2e = None
    if not args.no_e2e:
        e2e = e2e_slice(model, dev, rank, world, n_layers=args.e2e_layers, ddim=args.e2e_ddim, n_views=args.e2e_views,
                        oracle=(world == 1 and not args.no_parity), subjects_per_gpu=args.e2e_subjects, batch=min(args.e2e_batch, args.e2e_subjects))
    train = None
    if not args.no_train:
        from humanliff_amd.improved_diffusion.script_util import create_gaussian_diffusion
        train = bench_train(model, create_gaussian_diffusion(steps=1000), dev, rank, world)
    del model
    torch.cuda.empty_cache()
    render = None
    if not args.no_render:
        rsecs, rroof, rays_per_rank = bench_render(args, rank, world, dev)
        render = {"metric": "Mrays/sec@256spp", "value": round(world * rays_per_rank / rsecs / 1e6, 4), "unit": "Mrays/s",
                  "dtype": "f32 results; MLP products = fp16x2 (two fp16 planes per operand, three partial products, fp32 accumulation - fp32 tolerance, same test "
                           "bounds as the fp32-MFMA kernel whose figure is roofline.fp32_products)",
                  "views_per_gpu": args.views, "ms_per_view": round(rsecs * 1e3 / args.views, 3), "host_inclusive": rroof.pop("host_inclusive", None), "roofline": rroof,
                  "config": {"workload": "configs[2]: tri-plane NeRF render 512x512, n_samples=128 + n_importance=128, "
                                         "views of a 36-view orbit, random 256x256x27 tri-plane", "rays_per_view": 512 * 512}}
    fit = None
    if not args.no_fit:
        torch.cuda.empty_cache()
        fit = bench_fit(args, rank, world, dev)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        if render is not None:
            render["cpu_baseline"], render["parity"] = cpu_baseline_render(threads, dev=None if args.no_parity else dev)
        if fit is not None:
            fit["cpu_baseline"] = cpu_baseline_fit(threads)
    if rank == 0:
        line = {
            "metric": "denoise-steps/sec", "value": round(value, 3), "unit": "denoise-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(secs * 1e3 / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32",
            "dtype_note": "fp32 tensors and accumulators; conv / attention products from two fp16 planes per operand (fp32-class error); fp32_mfma_mode = all-fp32 pipe",
            "data": "synthetic",
            "config": {"workload": "configs[1]: 256x256x27 tri-plane UNet (controlnet, 497M params), 1000-step DDPM "
                                   "p_sample_loop, batch=4 per GPU", "global_batch": world * args.batch,
                       "parallelism": f"replicas x{world} (subjects sharded, final all-gather only)",
                       "gflop_per_sample_step": UNET_GFLOP_PER_SAMPLE_STEP},
            "step_tflops": round(world * args.batch * args.steps * UNET_GFLOP_PER_SAMPLE_STEP / secs / 1e3, 2),
            "roofline": roof, "cpu_baseline": cpu, "parity": parity, "e2e": e2e, "render": render, "fit": fit, "train": train, "rccl": rccl,
        }
        # flat copies of the other headline figures: inside `roofline` (a driver that keeps its scalars keeps these) and as the LAST key of the line
        sweep = (roof or {}).get("batch_sweep") or {}
        summary = {"denoise_steps_per_s_b4": round(value / world, 3),
                   "denoise_steps_per_s_b1": (sweep.get("batch1") or {}).get("eager", {}).get("value"),
                   "denoise_steps_per_s_b8": (sweep.get("batch8") or {}).get("eager", {}).get("value"),
                   "e2e_seconds_per_subject": (e2e or {}).get("seconds_per_subject"),
                   "e2e_psnr_db_vs_oracle": ((e2e or {}).get("parity") or {}).get("psnr_db"),
                   "render_mrays_per_s": (render or {}).get("value"),
                   "render_host_inclusive_mrays_per_s": ((render or {}).get("host_inclusive") or {}).get("value"),
                   "fit_iters_per_s": (fit or {}).get("value"), "n_gpus": world,
                   "rccl_backend": (rccl or {}).get("backend"), "rccl_world_size": (rccl or {}).get("world_size")}
        for k, v in summary.items():
            if isinstance(roof, dict) and v is not None and k not in ("n_gpus",):
                roof["hl_" + k] = v
        line["summary"] = summary
        # Everything measured goes to bench_detail.json; stdout gets ONE compact line (bench_line.py: < 6 KB, no long strings) - round 5's
        # 20 KB line was more than the driver's log reader kept.
        import bench_line
        detail = json.dumps(line, indent=1)
        for d in (os.path.join(ROOT, "gpurun_out"), ROOT):
            try:
                os.makedirs(d, exist_ok=True)
                with open(os.path.join(d, "bench_detail.json"), "w") as f:
                    f.write(detail + "\n")
            except OSError:
                pass
        sys.stdout.flush()
        print(json.dumps(bench_line.compact_line(line)), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
