"""bench_legs.py - the measurement legs behind bench.py (not a script).

bench.py holds the driver contract: argument parsing, the N > 1 self-launch, the order of the legs and the ONE compact JSON line (bench_line.py).  Everything that
measures lives here: the headline denoise loop (`bench_unet`: configs[1], with the live roofline of the dominant convolution kernel), the batch sweep, the parity legs
against the reference's vectors and the oracle, the configs[3]/[4] per-GPU slice (`e2e_slice`), the render leg (configs[2]: both schedules, three product modes), the
SURVEY 8(f) legs (fit, train) and the `cpu_baseline` legs (the ONLY place besides tests/ and smoke() that imports oracle/).
"""
import ctypes as C
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_{f16,bf16}, dense
PEAK_F16_FULL_MANTISSA_TFLOPS = 1655.0   # measured: independent v_mfma_f32_32x32x16_f16 on normal fp16 operands with random 10-bit mantissas (zeros: 2481) - scripts/microbench/mfma_data_power.hip
UNET_GFLOP_PER_SAMPLE_STEP = 2015.4
RENDER_FLOP_PER_RAY = 128 * 79616 + 256 * 132608   # 44 138 496 at 128+128
FULL_FLOP_PER_POINT = 132608            # density + colour MLP at one sample point
FINE_FLOP_PER_RAY = 256 * 132608
COARSE_FLOP_PER_RAY = 128 * 79616
FINE_FLOP_PER_RAY_TOTAL = FINE_FLOP_PER_RAY + COARSE_FLOP_PER_RAY   # the reference's schedule: 44 138 496 FLOP per ray
# Fabric-side bytes per launch of the dominant kernels from the separate rocprofv3 PMC passes (FETCH_SIZE doubled per the gfx950
# calibration, WRITE_SIZE as is).  Not measured by this script - PMC collection needs its own runs (scripts/refresh_profiles_r4.sh).
PMC_TRAFFIC = {"k_conv_avg_launch_b4": 267e6, "k_march_fine_512x512": 4.1e9, "k_march_eval_512x512": 0.72e9, "k_march_b3w_eval_512x512": 0.726e9,
               "source": "profiles/r05_pmc_hbm_traffic.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; k_conv_h2s: (122.75 GB read "
                         "+ 48.69 GB written) / 642 launches of all its shapes = 267 MB per launch; the whole conv path 198.0 + 81.3 GB per 6 forwards), profiles/r05_pmc_render_traffic.md (k_march_plw<2>: 188.8 MB read + 536.9 MB written "
                         "per launch; k_march_plw<3> the same bytes) and profiles/notes_design_rounds_1_to_3.md (k_march<true,true>: (1.50 + 4.29 GB) / 8 launches)"}

F4 = dict(image_size=256, in_channels=27, out_channels=27, num_channels=192, num_res_blocks=3, num_heads=4,
          num_heads_upsample=-1, attention_resolutions="32,16,8", dropout=0.0, learn_sigma=False, sigma_small=False,
          class_cond=True, diffusion_steps=1000, noise_schedule="linear", timestep_respacing="", use_kl=False,
          predict_xstart=False, rescale_timesteps=False, rescale_learned_sigmas=True, use_checkpoint=False,
          use_scale_shift_norm=True, cond_type="controlnet", use_3d_aware=False)


def dist_setup(n_gpus):
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("HL_BENCH_BACKEND", "nccl")    # "gloo": rehearsal of the N > 1 code path with all ranks on one GPU (no RCCL there)
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))   # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend)
    return rank, local, world


def barrier(world):
    torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(seconds, world, dev):
    if world == 1:
        return seconds
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class ClockSampler:
    """Samples the shader clock (sclk) of the busiest GPU from sysfs while a leg runs (a thread reading
    /sys/class/drm/card*/device/pp_dpm_sclk every 50 ms; the active level is the line marked '*')."""

    def __init__(self, period=0.05):
        import glob
        self.files = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
        self.period, self.samples, self._stop, self._th = period, [], False, None

    @staticmethod
    def _read(path):
        try:
            for line in open(path).read().splitlines():
                if line.rstrip().endswith("*"):
                    return float(line.split(":")[1].strip().split("M")[0])
        except Exception:
            return None
        return None

    def _run(self):
        while not self._stop:
            vals = [v for v in (self._read(f) for f in self.files) if v is not None]
            if vals:
                self.samples.append(max(vals))
            time.sleep(self.period)

    def start(self):
        import threading
        self._th = threading.Thread(target=self._run, daemon=True)
        self._th.start()

    def stop(self):
        self._stop = True
        if self._th is not None:
            self._th.join()
        if not self.samples:
            return {"sclk_mhz_mean": None, "sclk_mhz_min": None, "sclk_samples": 0, "sclk_source": "pp_dpm_sclk not readable on this box"}
        return {"sclk_mhz_mean": round(sum(self.samples) / len(self.samples), 1), "sclk_mhz_min": min(self.samples),
                "sclk_mhz_max": max(self.samples), "sclk_samples": len(self.samples),
                "sclk_source": "max over /sys/class/drm/card*/device/pp_dpm_sclk (active level), sampled every 50 ms during the leg"}


def build_unet(dev, seed=1):
    from humanliff_amd import synthetic as syn
    from humanliff_amd.improved_diffusion.script_util import create_model_and_diffusion
    model, diffusion = create_model_and_diffusion(**F4)
    keys = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    sd = syn.state_from_shapes(keys, seed=seed)   # every zero_module'd tensor re-randomised (SURVEY 8(c) rule 1)
    model.load_state_dict(sd)
    return model.to(dev).eval(), diffusion, sd


def bench_unet(args, rank, world, dev):
    from humanliff_amd import _lib
    B = args.batch
    model, diffusion, sd = build_unet(dev)
    g = torch.Generator().manual_seed(7 + rank)
    x_T = torch.randn((B, 27, 256, 256), generator=g).to(dev)
    x_cond = torch.zeros((B, 27, 256, 256), device=dev)          # layer 0: zeros (triplane_sample_layered.py:124-129)
    y = torch.zeros((B,), dtype=torch.int64, device=dev)
    it = diffusion.p_sample_loop_progressive(model, (B, 27, 256, 256), x_cond=x_cond, noise=x_T, clip_denoised=True,
                                             model_kwargs={"y": y}, device=dev)
    out = next(it)   # first step binds the state_dict (hl_unet_create) and sizes the workspace
    if args.no_overlap:
        _lib.check(_lib.lib().hl_unet_set_overlap(model._hip[0], 0))
    for _ in range(max(args.warmup - 1, 0)):
        out = next(it)
    gathered = torch.empty((world * B, 27, 256, 256), device=dev) if world > 1 else None   # rank-major, like the reference's gather
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = next(it)
    if world > 1:   # final gather of the samples (triplane_sample_layered.py:211-212): one collective into the preallocated result
        import torch.distributed as dist
        dist.all_gather_into_tensor(gathered, out["sample"].contiguous())
    barrier(world)
    secs = max_over_ranks(time.perf_counter() - t0, world, dev)
    assert torch.isfinite(out["sample"]).all()
    # ---- roofline leg: one more step with per-launch HIP events inside hl_unet_forward ----
    L = _lib.lib()
    handle = model._hip[0]
    _lib.check(L.hl_unet_profile(handle, 1))
    next(it)
    ms, fl, xf, nl = (C.c_double * 4)(), (C.c_double * 4)(), (C.c_double * 4)(), (C.c_int64 * 4)()
    _lib.check(L.hl_unet_profile_read_ex(handle, ms, fl, xf, nl))
    dv, dk = (C.c_double * 4)(), (C.c_int * 5)()
    _lib.check(L.hl_unet_profile_dominant(handle, dv, dk))
    _lib.check(L.hl_unet_profile(handle, 0))
    fam = {0: "k_conv_dma / k_conv (direct implicit GEMM)", 1: "k_conv_wino (Winograd F(2x2,3x3))", 2: "k_conv_bf3 (bf16x3)",
           3: "k_conv_wino4w / k_conv_wino4 (Winograd F(4x4,3x3))", 5: "k_conv_h16 / k_conv1_h16 (16-bit operands)",
           6: "k_conv_h2s / k_conv1_h2s (direct convolution, fp32 products from two fp16 planes per operand: three fp16 MFMAs per product, fp32 accumulation)"}.get(dk[0], f"path {dk[0]}")
    dom_ms = dv[0] / max(dv[3], 1.0)
    dom_exec = dv[2] / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    dom_peak, dom_note = PEAK_F32_MFMA_TFLOPS, None
    if dk[0] == 6:      # fp16x2 kernels: three fp16 partial products per fp32 product, issued on the 16-bit matrix pipe
        dom_exec = 3.0 * dv[1] / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        dom_peak = PEAK_BF16_MFMA_TFLOPS       # (v_mfma_f32_32x32x16_f16 and _bf16 share the dense 16-bit peak)
        dom_note = ("fp16 FLOPs ISSUED (three partial products per fp32 product of the direct convolution) against the dense fp16 matrix peak; the same launches in the "
                    "path's own unit: `algorithmic` (fp32-equivalent work).  The 2.5 PFLOP/s peak is reached with zero operands only: at the same reported 2.4 GHz a chain of "
                    "independent v_mfma_f32_32x32x16_f16 on full-mantissa fp16 operands runs at 1655 TFLOP/s (`peak_full_mantissa`, scripts/microbench/mfma_data_power.hip, "
                    "profiles/r06_unet_regression.md); MFMAs alone take 0.73 of the kernel's time (profiles/r05_unet_fill_experiments.md, sections 6 - 8)")
    dominant = {"kernel": fam, "layers": f"{dk[4]}x{dk[4]} convolutions with {dk[3]} output channels @{256 >> dk[1]}x{256 >> dk[1]}, batch {B}"
                                         + (" behind a nearest-x2 upsample" if dk[2] else "") + " (all input channel counts: one rocprofv3 kernel / grid row)",
                "launches_per_step": int(dv[3]), "avg_launch_ms": round(dom_ms, 4), "total_ms_per_step": round(dv[0], 3),
                "executed_gflop_per_launch": round(dv[2] / 1e9, 2), "algorithmic_gflop_per_launch": round(dv[1] / 1e9, 2),
                "executed_tflops": round(dom_exec, 2), "algorithmic_tflops": round(dv[1] / (dom_ms * 1e-3) / 1e12, 2) if dom_ms > 0 else None}
    conv_ms, conv_fl, conv_n = ms[0], fl[0], nl[0]
    achieved = conv_fl / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    executed = xf[0] / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    # `achieved` / `frac`: FLOPs the matrix pipe actually EXECUTES (Winograd F(4x4,3x3) issues 36 of the 144 multiplies of a direct 3x3
    # convolution per 4x4 outputs, F(2x2,3x3) 16 of 36 per 2x2) over the conv-path time, against the fp32 MFMA peak - a real fraction (<= 1).  The algorithmic figure (direct-convolution
    # FLOPs of SURVEY 8(d) over the same time) is reported next to it as `algorithmic`; it can exceed the peak.
    # `roofline` is the DOMINANT KERNEL's: FLOPs its launches issue to the matrix pipe (Winograd F(4x4,3x3) issues 36 of the 144 multiplies of
    # a direct 3x3 convolution per 4x4 outputs) / its own launch time (HIP events inside hl_unet_forward, behind the GroupNorm pre-pass) /
    # the fp32 MFMA peak - a real fraction.  `algorithmic` is the same launch priced in direct-convolution FLOPs (exceeds the
    # peak); `conv_path` the same two figures for ALL convolution launches of the step with their pre / post passes (round 2's `frac`).
    roof = {"bound": "mfma", "kernel": dominant["kernel"] + ": " + dominant["layers"],
            "achieved": round(dom_exec, 2), "peak": dom_peak, "unit": "TFLOP/s", "frac": round(dom_exec / dom_peak, 4),
            "avg_launch_ms": dominant["avg_launch_ms"], "launches_per_step": dominant["launches_per_step"], "ms_per_step_in_this_kernel": dominant["total_ms_per_step"],
            "peak_full_mantissa": PEAK_F16_FULL_MANTISSA_TFLOPS if dk[0] == 6 else None,
            "frac_of_full_mantissa_peak": round(dom_exec / PEAK_F16_FULL_MANTISSA_TFLOPS, 4) if dk[0] == 6 else None,
            "algorithmic": {"tflops": dominant["algorithmic_tflops"], "x_peak": round((dominant["algorithmic_tflops"] or 0.0) / PEAK_F32_MFMA_TFLOPS, 4),
                            "note": "direct-convolution FLOPs (SURVEY 8(d): 2*M*Cout*Cin*taps) of the same launches over the same time; not a roofline fraction"},
            "note": dom_note or ("fp32 MFMA and the vector ALU share the SIMD's fp32 lanes (scripts/microbench/mfma_fill.hip): the kernel's own input transform (VALU) is "
                                 "added to its MFMA time, so 1.0 is not reachable for a Winograd kernel - MFMAs alone run this launch shape at 0.70 (profiles/r03_wino4w_ablations.md)"),
            "conv_path": {"what": "all convolution launches of one denoise step (k_conv_wino4w / k_conv_wino4 / k_conv_wino / k_conv_dma / k_conv / k_conv_h16<.,2> / k_conv1_h2) with their pre / post "
                                  "passes (k_gn_apply, k_splitk_finish); the fp16x2 kernels' products are counted once (fp32-equivalent work), not three times", "executed_tflops": round(executed, 2), "frac": round(executed / PEAK_F32_MFMA_TFLOPS, 4),
                          "algorithmic_tflops": round(achieved, 2), "algorithmic_x_peak": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                          "launches_per_step": int(conv_n), "gflop_per_step": round(conv_fl / 1e9, 1), "executed_gflop_per_step": round(xf[0] / 1e9, 1),
                          "ms_per_step": round(conv_ms, 3), "avg_launch_ms": round(conv_ms / max(conv_n, 1), 4)},
            "traffic": PMC_TRAFFIC["k_conv_avg_launch_b4"] if B == 4 else None, "traffic_source": PMC_TRAFFIC["source"],
            "other_ms": {"groupnorm": round(ms[1], 3), "attention": round(ms[2], 3), "emb_prep": round(ms[3], 3)}}
    # ---- sustained leg: the loop keeps running for >= 200 more steps; steps/s and the shader clock sampled meanwhile ----
    roof["sustained"] = None
    if world == 1 and args.sustained_steps > 0:
        clk = ClockSampler()
        torch.cuda.synchronize()
        clk.start()
        ts = time.perf_counter()
        for _ in range(args.sustained_steps):
            out = next(it)
        torch.cuda.synchronize()
        dt = time.perf_counter() - ts
        roof["sustained"] = {"steps": args.sustained_steps, "value": round(B * args.sustained_steps / dt, 3), "unit": "denoise-steps/s",
                             "ms_per_step": round(dt * 1e3 / args.sustained_steps, 3), "seconds": round(dt, 2), **clk.stop()}
        assert torch.isfinite(out["sample"]).all()
    del it
    roof["batch_sweep"] = bench_batches(model, dev) if (world == 1 and not args.no_batch_sweep) else None
    # ---- opt-in arithmetic mode (not the headline): fp32 products emulated with three bf16 planes per operand ----
    roof["bf16x3_mode"] = None
    if world == 1 and not args.no_bf16x3_leg:
        xx = torch.randn((B, 27, 256, 256), generator=torch.Generator().manual_seed(99)).to(dev)
        tt = torch.full((B,), 500, dtype=torch.int64, device=dev)
        with torch.no_grad():
            ref = model(xx, tt, x_cond, y=y)
            model.set_conv_mode("bf16x3")
            alt = model(xx, tt, x_cond, y=y)
        it3 = diffusion.p_sample_loop_progressive(model, (B, 27, 256, 256), x_cond=x_cond, noise=x_T, clip_denoised=True,
                                                  model_kwargs={"y": y}, device=dev)
        next(it3); next(it3)
        torch.cuda.synchronize()
        k3 = max(2, min(args.steps, 6))
        t3 = time.perf_counter()
        for _ in range(k3):
            next(it3)
        torch.cuda.synchronize()
        s3 = time.perf_counter() - t3
        del it3
        model.set_conv_mode("fp32")
        roof["bf16x3_mode"] = {
            "what": "UNetModel.set_conv_mode('bf16x3') / HL_CONV_BF16X3: same fp32 tensors and accumulators, products on "
                    "v_mfma_f32_32x32x16_bf16 from exact 3-way bf16 splits (6 partial products, error <= 3*2^-24 per product); "
                    "opt-in, NOT used for `value`",
            "value": round(B * k3 / s3, 3), "unit": "denoise-steps/s", "steps": k3, "ms_per_step": round(s3 * 1e3 / k3, 3),
            "max_abs_diff_vs_fp32_forward": float((alt - ref).abs().max()), "forward_output_mean_abs": float(ref.abs().mean())}
    # ---- the default's dispatch with EVERY product on the fp32 matrix pipe (no fp16x2 kernels): the reference point for `value` ----
    roof["fp32_mfma_mode"] = None
    if world == 1 and not args.no_bf16x3_leg:
        xx = torch.randn((B, 27, 256, 256), generator=torch.Generator().manual_seed(99)).to(dev)
        tt = torch.full((B,), 500, dtype=torch.int64, device=dev)
        with torch.no_grad():
            ref = model(xx, tt, x_cond, y=y)
            model.set_conv_mode("fp32_mfma")
            alt = model(xx, tt, x_cond, y=y)
        it3 = diffusion.p_sample_loop_progressive(model, (B, 27, 256, 256), x_cond=x_cond, noise=x_T, clip_denoised=True,
                                                  model_kwargs={"y": y}, device=dev)
        next(it3); next(it3)
        torch.cuda.synchronize()
        k3 = max(2, min(args.steps, 10))
        t3 = time.perf_counter()
        for _ in range(k3):
            next(it3)
        torch.cuda.synchronize()
        s3 = time.perf_counter() - t3
        del it3
        model.set_conv_mode("fp32")
        roof["fp32_mfma_mode"] = {
            "what": "UNetModel.set_conv_mode('fp32_mfma') / HL_CONV_FP32_MFMA: the default's dispatch without k_conv1_h2 / k_conv_h16<.,2> - every product on "
                    "v_mfma_f32_32x32x2_f32 (the default of rounds 3-4).  `value` uses the default mode, whose 1x1 layers and one-round 3x3 layers form fp32 products "
                    "from two fp16 planes per operand (fp32 accumulation; float64-checked in tests/test_unet_gpu.py, same oracle bounds)",
            "value": round(B * k3 / s3, 3), "unit": "denoise-steps/s", "steps": k3, "ms_per_step": round(s3 * 1e3 / k3, 3),
            "max_abs_diff_default_vs_fp32_mfma_forward": float((alt - ref).abs().max()), "forward_output_mean_abs": float(ref.abs().mean())}
    # ---- opt-in arithmetic mode (not the headline): fp16 operands / fp32 accumulation on the 3x3 layers (k_conv_h16) ----
    roof["fp16_mode"] = None
    if world == 1 and not args.no_bf16x3_leg:
        xx = torch.randn((B, 27, 256, 256), generator=torch.Generator().manual_seed(99)).to(dev)
        tt = torch.full((B,), 500, dtype=torch.int64, device=dev)
        d50 = create_gaussian_diffusion_for_bench("ddim50")
        with torch.no_grad():
            ref = model(xx, tt, x_cond, y=y)
            ref50 = d50.ddim_sample_loop(model, (1, 27, 256, 256), x_cond=x_cond[:1], noise=x_T[:1], clip_denoised=True, model_kwargs={"y": y[:1]}, device=dev)
            model.set_conv_mode("fp16")
            alt = model(xx, tt, x_cond, y=y)
            alt50 = d50.ddim_sample_loop(model, (1, 27, 256, 256), x_cond=x_cond[:1], noise=x_T[:1], clip_denoised=True, model_kwargs={"y": y[:1]}, device=dev)
        it3 = diffusion.p_sample_loop_progressive(model, (B, 27, 256, 256), x_cond=x_cond, noise=x_T, clip_denoised=True,
                                                  model_kwargs={"y": y}, device=dev)
        next(it3); next(it3)
        torch.cuda.synchronize()
        k3 = max(2, min(args.steps, 10))
        t3 = time.perf_counter()
        for _ in range(k3):
            next(it3)
        torch.cuda.synchronize()
        s3 = time.perf_counter() - t3
        del it3
        model.set_conv_mode("fp32")
        psnr = lambda a, b: float(10 * torch.log10(b.abs().max() ** 2 / ((a - b) ** 2).mean()))  # noqa: E731
        roof["fp16_mode"] = {
            "what": "UNetModel.set_conv_mode('fp16') / HL_CONV_FP16: fp16 operands, fp32 accumulation (v_mfma_f32_32x32x16_f16, k_conv_h16) on "
                    "the 3x3 / stride-1 layers, everything else as the fp32 mode - the operand precision of the reference's own TF32 "
                    "convolutions; opt-in, NOT used for `value`, not an fp32-tolerance mode",
            "value": round(B * k3 / s3, 3), "unit": "denoise-steps/s", "steps": k3, "ms_per_step": round(s3 * 1e3 / k3, 3),
            "psnr_db_forward_vs_fp32": round(psnr(alt, ref), 1), "max_abs_diff_vs_fp32_forward": float((alt - ref).abs().max()),
            "psnr_db_ddim50_sample_vs_fp32": round(psnr(alt50, ref50), 1), "max_abs_diff_ddim50_sample": float((alt50 - ref50).abs().max())}
    return secs, roof, sd, model


def collective_probe(args, rank, world, dev, iters=3):
    """What the N > 1 run actually ran on: the process group's backend and size as torch.distributed reports them, and the two gathers of
    the path timed alone - the final sample gather (triplane_sample_layered.py:211-212: B x 27 x 256 x 256 fp32 per rank) and one subject's
    uint8 image gather (185 views of 512 x 512 x 3 per rank) - as all_gather_into_tensor into the preallocated result.  GB/s = bytes every
    rank RECEIVES from the others / max-over-ranks seconds."""
    import torch.distributed as dist
    out = {"world_size": dist.get_world_size(), "rank0_sees_ranks": dist.get_world_size(), "backend": str(dist.get_backend()),
           "device_count_visible": torch.cuda.device_count()}
    for name, shard in (("sample_gather", torch.randn((args.batch, 27, 256, 256), device=dev)),
                        ("image_gather_uint8", torch.zeros((args.e2e_views, 512, 512, 3), dtype=torch.uint8, device=dev))):
        full = torch.empty((world * shard.shape[0],) + tuple(shard.shape[1:]), dtype=shard.dtype, device=dev)
        dist.all_gather_into_tensor(full, shard)          # warm-up (connection set-up)
        barrier(world)
        t0 = time.perf_counter()
        for _ in range(iters):
            dist.all_gather_into_tensor(full, shard)
        barrier(world)
        dt = max_over_ranks(time.perf_counter() - t0, world, dev) / iters
        recv = shard.numel() * shard.element_size() * (world - 1)
        out[name] = {"bytes_per_rank_shard": shard.numel() * shard.element_size(), "ms": round(dt * 1e3, 3), "recv_gb_per_s_per_rank": round(recv / dt / 1e9, 2)}
        del full
    return out


def create_gaussian_diffusion_for_bench(respacing):
    from humanliff_amd.improved_diffusion.script_util import create_gaussian_diffusion
    return create_gaussian_diffusion(steps=1000, timestep_respacing=respacing)


def bench_batches(model, dev, batches=(1, 8, 4), steps=12, warm=3):
    """denoise-steps/s of the same 1000-step DDPM loop at other batch sizes (the shipped sampling script runs --batch_size 1), eager and in
    graph mode (`diffusion.use_hip_graph`: one step captured into a HIP graph and replayed - the per-launch host cost is what bounds small
    batches).  Not used for `value`."""
    from humanliff_amd.improved_diffusion.script_util import create_gaussian_diffusion
    out = {}
    for B in batches:
        row = {}
        for graph in (False, True):
            d = create_gaussian_diffusion(steps=1000, timestep_respacing="")
            d.use_hip_graph = graph
            g = torch.Generator().manual_seed(70 + B)
            x_T = torch.randn((B, 27, 256, 256), generator=g).to(dev)
            xc = torch.zeros_like(x_T)
            y = torch.zeros((B,), dtype=torch.int64, device=dev)
            it = d.p_sample_loop_progressive(model, (B, 27, 256, 256), x_cond=xc, noise=x_T, clip_denoised=True, model_kwargs={"y": y}, device=dev)
            for _ in range(warm):
                o = next(it)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                o = next(it)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            assert torch.isfinite(o["sample"]).all()
            del it
            row["graph" if graph else "eager"] = {"value": round(B * steps / dt, 2), "ms_per_step": round(dt * 1e3 / steps, 3)}
        out[f"batch{B}"] = row
    out["unit"] = "denoise-steps/s"
    out["what"] = f"p_sample_loop of the production net, {steps} timed steps after {warm} warm-up steps per point; graph = diffusion.use_hip_graph (HIP graph replay of one step)"
    return out


def _psnr(a, b):
    import math
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 200.0 if mse == 0 else -10.0 * math.log10(mse)


def e2e_chain(model, dev, golden=None, mlp_fp16=False, mlp_products=None):
    """BASELINE configs[0] / [3] / [4] at one-GPU scale, against the REFERENCE's own outputs (tests/golden/chain_f4_ddim10.npz, made by
    tests/golden/gen_golden_chain.py from /root/reference): the flow of scripts/triplane_sample_layered.py:112-177 on the production
    network - per cloth layer y = layer, x_cond = the previous layer's sample, ddim_sample_loop (DDIM-10, B = 1) on injected noise,
    sample.reshape(1,3,9,256,256), render() of one 128x128 orbit view at 32+32 samples.  Returns per-layer error figures (tri-plane
    values are in [-1,1], colours in [0,1]); used by tests/test_e2e_gpu.py and by the `parity` object of the bench line."""
    import numpy as np
    from humanliff_amd import synthetic as syn
    from humanliff_amd.NeRF import Renderer, render
    from humanliff_amd.improved_diffusion.script_util import create_gaussian_diffusion
    g = np.load(golden or os.path.join(ROOT, "tests", "golden", "chain_f4_ddim10.npz"))
    diffusion = create_gaussian_diffusion(steps=1000, learn_sigma=False, noise_schedule="linear", timestep_respacing="ddim10")
    rend = Renderer(use_canonical_space=False, triplane_dim=256, triplane_ch=27, smpl_type="smpl", test=True)
    rend.load_state_dict(syn.render_mlp_state(3), strict=False)
    rend = rend.to(dev)
    rend.mlp_fp16 = mlp_fp16                      # opt-in mode of the renderer (tests pin it to the reference's images through this flow)
    if mlp_products is not None:
        rend.mlp_products = mlp_products
    IMG, NS, stride = int(g["img"]), int(g["n_samples"]), int(g["stride"])
    rays_o, rays_d, near, far = syn.orbit_rays(int(g["view"]), int(g["n_views"]), IMG, IMG)
    assert np.allclose([float(rays_d.double().sum()), float(rays_d.double().abs().sum())], g["rays_ck"], rtol=0, atol=1e-6)
    tp = {"world_bounds": torch.tensor(syn.WORLD_BOUNDS)[None].to(dev)}
    n = {"i": 0}

    def draw(shape):          # the generator's injected noise stream: draw i = randn from manual_seed(9000 + i)
        gg = torch.Generator().manual_seed(9000 + n["i"])
        n["i"] += 1
        return torch.randn(tuple(shape), generator=gg)

    shape = (1, 27, 256, 256)
    x_cond = torch.zeros(shape, device=dev)
    layers = []
    orig = torch.randn_like
    torch.randn_like = lambda ref: draw(ref.shape).to(ref.device)
    try:
        for layer in range(int(g["n_layers"])):
            y = torch.full((1,), layer, dtype=torch.int64, device=dev)
            x_T = draw(shape).to(dev)
            sample = diffusion.ddim_sample_loop(model, shape, x_cond=x_cond, noise=x_T, clip_denoised=True, model_kwargs={"y": y})
            s = sample.cpu()
            want_sub = torch.from_numpy(g[f"sample{layer}_sub"])
            got_sub = s[:, :, ::stride, ::stride]
            e_tri = max(float((got_sub - want_sub).abs().max()), float((s[0, :, 100, :] - torch.from_numpy(g[f"sample{layer}_row100"])).abs().max()))
            ck = float(s.double().abs().sum())
            tri_planes = sample[0:1].reshape(1, 3, -1, 256, 256)                       # :158
            torch.manual_seed(5)                                                       # sample_pdf's uniforms: CPU generator, like the reference
            rgb, acc, _, depth = render(chunk=IMG * IMG, rays_o=rays_o[None].to(dev), rays_d=rays_d[None].to(dev), near=near[None].to(dev),
                                        far=far[None].to(dev), tri_planes=tri_planes, tp_input=tp, renderer=rend, n_samples=NS, perturb=0.,
                                        n_importance=NS)
            w = lambda k: torch.from_numpy(g[f"{k}{layer}"])  # noqa: E731
            layers.append({"layer": layer, "triplane_max_abs": e_tri, "triplane_psnr_db": round(_psnr(got_sub, want_sub), 2),
                           "triplane_abs_sum_rel": abs(ck - float(g[f"sample{layer}_ck"][1])) / float(g[f"sample{layer}_ck"][1]),
                           "triplane_channel_mean_max_abs": float(np.abs(s.double().mean(dim=(0, 2, 3)).numpy() - g[f"sample{layer}_chmean"]).max()),
                           "image_max_abs": float((rgb[0].cpu() - w("rgb")).abs().max()), "image_psnr_db": round(_psnr(rgb[0].cpu(), w("rgb")), 2),
                           "acc_max_abs": float((acc[0].cpu() - w("acc")).abs().max()), "depth_max_abs": float((depth[0].cpu() - w("depth")).abs().max())})
            x_cond = sample                                                            # :124-134: the next layer is conditioned on this one
    finally:
        torch.randn_like = orig
    return {"layers": layers, "ndraws": n["i"], "ndraws_reference": int(g["ndraws"]),
            "against": "the reference's outputs on identical noise (tests/golden/chain_f4_ddim10.npz <- tests/golden/gen_golden_chain.py)",
            "workload": "F4 net, DDIM-10, B=1, 2 cloth layers chained through x_cond -> reshape(1,3,9,256,256) -> one 128x128 view @32+32"}


def ddim50_parity(model, dev, golden=None, kind="ddim50"):
    """The sampler at the length the shipped scripts use, against the REFERENCE's own outputs (tests/golden/f4_ddim50.npz, made by
    tests/golden/gen_golden_ddim50.py from /root/reference): production network, `timestep_respacing="ddim50"`, B = 1, cloth layer 1
    conditioned on a seeded x_cond, all 50 steps of ddim_sample_loop_progressive on injected noise; compared after steps 1, 10, 25, 40
    and 50 (every 8th pixel + whole-tensor checksums).  Used by tests/test_fullsize_gpu.py and the `parity` object of the bench line."""
    import numpy as np
    from humanliff_amd.improved_diffusion.script_util import create_gaussian_diffusion
    # kind "p250": the SHIPPED configuration (triplane_scripts/SynBody_triplane_sample_layered_*.sh:24-26: --timestep_respacing 250, p_sample_loop,
    # batch 1) against tests/golden/f4_p250.npz (gen_golden_p250.py); states after steps 1, 50, 125, 200, 250
    g = np.load(golden or os.path.join(ROOT, "tests", "golden", "f4_ddim50.npz" if kind == "ddim50" else "f4_p250.npz"))
    diffusion = create_gaussian_diffusion(steps=1000, learn_sigma=False, noise_schedule="linear", timestep_respacing="ddim50" if kind == "ddim50" else "250")
    loop = diffusion.ddim_sample_loop_progressive if kind == "ddim50" else diffusion.p_sample_loop_progressive
    stride, keep, layer = int(g["stride"]), [int(k) for k in g["keep"]], int(g["layer"])
    n = {"i": 0}

    def draw(shape):
        gg = torch.Generator().manual_seed(9100 + n["i"])
        n["i"] += 1
        return torch.randn(tuple(shape), generator=gg)

    shape = (1, 27, 256, 256)
    x_cond = (torch.randn(shape, generator=torch.Generator().manual_seed(77)) * 0.3).clamp_(-1, 1)
    assert abs(float(x_cond.double().abs().sum()) - float(g["x_cond_ck"][1])) < 1e-6 * float(g["x_cond_ck"][1])
    y = torch.full((1,), layer, dtype=torch.int64, device=dev)
    steps = []
    orig = torch.randn_like
    torch.randn_like = lambda ref: draw(ref.shape).to(ref.device)
    try:
        x_T = draw(shape).to(dev)
        for k, out in enumerate(loop(model, shape, x_cond=x_cond.to(dev), noise=x_T, clip_denoised=True, model_kwargs={"y": y}, device=dev), 1):
            if k in keep:
                s_ = out["sample"].cpu()
                want = torch.from_numpy(g[f"step{k}_sub"])
                got = s_[:, :, ::stride, ::stride]
                ck = float(g[f"step{k}_ck"][1])
                steps.append({"step": k, "max_abs": float((got - want).abs().max()), "psnr_db": round(_psnr(got, want), 2),
                              "abs_sum_rel": abs(float(s_.double().abs().sum()) - ck) / ck, "value_abs_max": float(want.abs().max())})
    finally:
        torch.randn_like = orig
    final_row = float((s_[0, :, 100, :] - torch.from_numpy(g["final_row100"])).abs().max())
    chmean = float(np.abs(s_.double().mean(dim=(0, 2, 3)).numpy() - g["final_chmean"]).max())
    return {"steps": steps, "final_row100_max_abs": final_row, "final_channel_mean_max_abs": chmean, "ndraws": n["i"], "ndraws_reference": int(g["ndraws"]),
            "max_abs": max(s["max_abs"] for s in steps), "psnr_db": min(s["psnr_db"] for s in steps),
            "against": "the reference's outputs on identical noise (tests/golden/f4_ddim50.npz <- tests/golden/gen_golden_ddim50.py)" if kind == "ddim50" else
                       "the reference's outputs on identical noise (tests/golden/f4_p250.npz <- tests/golden/gen_golden_p250.py)",
            "workload": "F4 net, DDIM-50 (ddim_sample_loop_progressive), B=1, cloth layer 1 with a seeded x_cond" if kind == "ddim50" else
                        "F4 net, the shipped sampler: timestep_respacing=250, p_sample_loop_progressive, B=1, cloth layer 1 with a seeded x_cond"}


def e2e_slice(model, dev, rank, world, n_layers=4, ddim=50, n_views=185, res=512, n_check_views=3, n_check_rays=1024, oracle=True, subjects_per_gpu=8, batch=8):
    """The real per-GPU slice of BASELINE configs[3] / [4] (scripts/triplane_sample_layered.py:112-213) AT ITS REAL SIZE: per rank
    `subjects_per_gpu` subjects (64 subjects over 8 GPUs = 8 each), sampled `batch` at a time x
    `n_layers` cloth layers chained through x_cond x DDIM-`ddim` on the production network,
    the finished tri-plane reshaped to (1,3,9,256,256) and rendered into `n_views` orbit views of res x res at 128 + 128 samples, the
    samples and the uint8 images gathered over the ranks - humanliff_amd.distributed.sample_and_render, the code the multi-GPU script
    runs.  Timed as a whole and per stage (HIP events around the sampling and the render part); with `oracle`, `n_check_rays` rays of
    each of `n_check_views` views are rendered again by the CPU oracle from the SAME generated tri-plane, rays and uniforms (renderer
    parity decoupled from sampler drift) and compared (PSNR, colours in [0,1])."""
    import numpy as np
    from humanliff_amd import distributed as hd, synthetic as syn
    from humanliff_amd.NeRF import Renderer, render_view
    from humanliff_amd.SynBodyView_datasets import camera_rays
    from humanliff_amd.improved_diffusion.script_util import create_gaussian_diffusion
    diffusion = create_gaussian_diffusion(steps=1000, learn_sigma=False, noise_schedule="linear", timestep_respacing=f"ddim{ddim}")
    rend = Renderer(use_canonical_space=False, triplane_dim=256, triplane_ch=27, smpl_type="smpl", test=True)
    mlp = syn.render_mlp_state(3)
    rend.load_state_dict(mlp, strict=False)
    rend = rend.to(dev)
    shape = (27, 256, 256)
    tp = {"world_bounds": torch.tensor(syn.WORLD_BOUNDS)[None].to(dev)}
    bounds_np = np.asarray(syn.WORLD_BOUNDS, dtype=np.float64)
    u = torch.rand((res * res, 128), device=dev, generator=torch.Generator(device=dev).manual_seed(5 + rank))
    ev = {k: torch.cuda.Event(enable_timing=True) for k in ("s0", "s1", "r1")}
    state = {"first_render": True, "calls": 0}

    def sample_fn(x_cond, layer, ids):
        if state["calls"] == 0:
            ev["s0"].record()
        state["calls"] += 1
        noise = torch.stack([torch.randn(shape, generator=torch.Generator().manual_seed(4000 + 100 * i + layer)) for i in ids]).to(dev)   # per subject: independent of the batching
        y = torch.full((len(ids),), layer, dtype=torch.int64, device=dev)
        return diffusion.ddim_sample_loop(model, (len(ids),) + shape, x_cond=x_cond, noise=noise, clip_denoised=True, model_kwargs={"y": y}, device=dev)

    def cam(v):
        K, c2w, c = syn.orbit_camera(v, n_views, res, res)
        R = c2w.T.copy()
        return K, R, (-R @ c).reshape(3, 1)

    def render_fn(sid, sample, v):
        if state["first_render"]:
            ev["s1"].record()
            state["first_render"] = False
        planes = sample.reshape(1, 3, 9, 256, 256)                                  # triplane_sample_layered.py:158
        K, R, T = cam(v)
        return render_view(res, res, K, R, T, planes, tp, rend, n_samples=128, n_importance=128, u=u)[0]

    # warm-up outside the timed region: one forward at the sampling batch (binds that workspace) and one small view (packs the MLP)
    spg, nsub = subjects_per_gpu, world * subjects_per_gpu
    with torch.no_grad():
        model(torch.zeros((batch,) + shape, device=dev), torch.zeros((batch,), dtype=torch.int64, device=dev), torch.zeros((batch,) + shape, device=dev),
              y=torch.zeros((batch,), dtype=torch.int64, device=dev))
    render_view(*syn_cam64(syn), torch.zeros((1, 3, 9, 256, 256), device=dev), tp, rend, n_samples=128, n_importance=128)
    barrier(world)
    torch.cuda.reset_peak_memory_stats(dev)
    t0 = time.perf_counter()
    samples, images = hd.sample_and_render(sample_fn, render_fn, nsub, n_layers, shape, batch, n_views, (res, res, 3), dev, as_uint8=True)
    ev["r1"].record()
    barrier(world)
    secs = max_over_ranks(time.perf_counter() - t0, world, dev)
    t_sample, t_render = ev["s0"].elapsed_time(ev["s1"]) * 1e-3, ev["s1"].elapsed_time(ev["r1"]) * 1e-3
    steps, rays = nsub * n_layers * ddim, nsub * n_views * res * res
    assert samples.shape == (nsub, n_layers) + shape and images.shape == (nsub, n_views, res, res, 3) and images.dtype == torch.uint8
    assert torch.isfinite(samples).all()
    out = {"workload": f"configs[3]/[4] per-GPU slice at its real size: {spg} subjects per GPU (sampled {batch} at a time) x {n_layers} cloth layers x DDIM-{ddim} "
                       f"(production F4 net, layers chained through x_cond) -> reshape(1,3,9,256,256) -> {n_views} orbit views {res}x{res} @128+128 per subject "
                       "(Renderer.mlp_products = fp16x2) -> gather of samples (fp32) and, per subject and asynchronously, images (uint8); "
                       "humanliff_amd.distributed.sample_and_render",
           "seconds": round(secs, 3), "subjects": nsub, "subjects_per_gpu": spg, "seconds_per_subject": round(secs / spg, 3), "denoise_steps": steps, "rays": rays,
           "sampling": {"seconds_rank0": round(t_sample, 3), "denoise_steps_per_sec_per_gpu": round(spg * n_layers * ddim / t_sample, 2), "batch": batch},
           "rendering": {"seconds_rank0": round(t_render, 3), "mrays_per_sec_per_gpu": round(spg * n_views * res * res / t_render / 1e6, 4),
                         "ms_per_view": round(t_render * 1e3 / (spg * n_views), 3), "views_per_gpu": spg * n_views, "mlp_products": rend.mlp_products,
                         "note": "includes device ray generation (hl_camera_rays), the uint8 conversion and the per-subject image gather"},
           "peak_device_memory_gb": round(torch.cuda.max_memory_allocated(dev) / 1e9, 2), "image_gather_buffer_gb": round(images.numel() / 1e9, 3),
           "subjects_per_hour_per_gpu": round(3600.0 * spg / secs, 1), "image_mean": float(images.float().mean()) / 255.0,
           "sample_abs_max": float(samples.abs().max())}
    if oracle and rank == 0:
        from oracle import render_oracle as ro
        planes = samples[0, -1].reshape(3, 9, 256, 256)
        checks, psnrs = [], []
        for i in range(n_check_views):
            v = (i * n_views) // n_check_views
            K, R, T = cam(v)
            rays_o, rays_d, near, far, _ = camera_rays(res, res, K, R, T, bounds_np, dev, return_mask=False)
            hit = torch.nonzero(far != 1).flatten()                                  # rays that meet the box (the others see the far plane only)
            pick = hit[torch.randperm(hit.numel(), generator=torch.Generator().manual_seed(v))[:n_check_rays].to(dev)]
            got = rend.render(tp, None, None, rays_o[pick][None], rays_d[pick][None], near[pick][None, :, None], far[pick][None, :, None],
                              planes[None], 128, False, n_samples=128, u=u[pick][None])
            with torch.no_grad():
                rgb, acc, depth = ro.render_rays(mlp, planes.cpu(), torch.tensor(syn.WORLD_BOUNDS), rays_o[pick].cpu(), rays_d[pick].cpu(),
                                                 near[pick].cpu(), far[pick].cpu(), 128, 128, u=u[pick].cpu())
            # the same rays inside the full-view launch: the image the slice produced (uint8, truncated like the reference's writer)
            full = images[0, v].reshape(-1, 3)[pick].cpu().float() / 255.0
            assert float((full - (rgb.clamp(0, 1) * 255).to(torch.uint8).float() / 255.0).abs().max()) <= 1.0 / 255.0 + 1e-6
            p_ = _psnr(got["rgb_map"][0].cpu(), rgb)
            psnrs.append(p_)
            checks.append({"view": v, "rays": int(pick.numel()), "psnr_db": round(p_, 2), "max_abs": float((got["rgb_map"][0].cpu() - rgb).abs().max()),
                           "acc_max_abs": float((got["acc_map"][0].cpu() - acc).abs().max()), "depth_max_abs": float((got["depth_map"][0].cpu() - depth).abs().max())})
        out["parity"] = {"psnr_db": round(min(psnrs), 2), "views": checks,
                         "what": f"{n_check_rays} box-hitting rays of each of {n_check_views} views of the GENERATED tri-plane: HIP render vs the CPU oracle on identical "
                                 "rays / uniforms (north-star bar: PSNR >= 45 dB); the uint8 pixels of the slice's own images equal the oracle's within one grey level"}
    if world == 1 and hasattr(model, "set_conv_mode"):
        # the same subject sampled again in the opt-in fp16-operand mode (same noise, same chaining), and what that does to the IMAGES:
        # the views of the parity check rendered from both tri-planes, compared as the uint8 pixels the script would write
        model.set_conv_mode("fp16")
        try:
            with torch.no_grad():
                model(torch.zeros((1,) + shape, device=dev), torch.zeros((1,), dtype=torch.int64, device=dev), torch.zeros((1,) + shape, device=dev),
                      y=torch.zeros((1,), dtype=torch.int64, device=dev))
            torch.cuda.synchronize()
            t16 = time.perf_counter()
            xc, last = torch.zeros((1,) + shape, device=dev), None
            for layer in range(n_layers):
                xc = sample_fn(xc, layer, [0])
            last = xc
            torch.cuda.synchronize()
            t16 = time.perf_counter() - t16
        finally:
            model.set_conv_mode("fp32")
        ps = []
        rend.mlp_fp16 = True                                      # ... and rendered by the fp16-operand MLP (Renderer.mlp_fp16)
        render_fn(0, last, 0)
        torch.cuda.synchronize()
        tr16 = time.perf_counter()
        for v in range(n_views):
            render_fn(0, last, v)
        torch.cuda.synchronize()
        tr16 = time.perf_counter() - tr16
        for i in range(n_check_views):
            v = (i * n_views) // n_check_views
            img16 = render_fn(0, last, v)
            a = (img16.clamp(0, 1) * 255).to(torch.uint8).float().reshape(-1, 3).cpu() if img16.dtype != torch.uint8 else img16.float().reshape(-1, 3).cpu()
            b = images[0, v].reshape(-1, 3).cpu().float()
            mse = float(((a - b) ** 2).mean())
            ps.append(99.0 if mse == 0 else 10 * math.log10(255.0 ** 2 / mse))
        rend.mlp_fp16 = False
        tp_mse = float(((last[0] - samples[0, -1]) ** 2).mean())
        out["fp16_mode"] = {"what": "the same subject sampled in UNetModel.set_conv_mode('fp16') and rendered with Renderer.mlp_fp16 (both opt-in; same noise, layers "
                                    "chained the same way): sampling and rendering time (rendering without the uint8 conversion / gather of the slice), and the "
                                    "PSNR of the uint8 images / of the last layer's tri-plane against the fp32 slice above",
                            "sampling_seconds": round(t16, 3), "denoise_steps_per_sec": round(n_layers * ddim / t16, 2),
                            "rendering_seconds": round(tr16, 3), "mrays_per_sec": round(n_views * res * res / tr16 / 1e6, 3),
                            "seconds_per_subject": round(t16 + tr16, 3),
                            "image_psnr_db_min": round(min(ps), 2), "image_psnr_db_views": [round(v, 2) for v in ps],
                            "triplane_psnr_db": round(10 * math.log10(float(samples[0, -1].abs().max()) ** 2 / tp_mse), 2) if tp_mse > 0 else 99.0}
    return out


def syn_cam64(syn):
    K, c2w, c = syn.orbit_camera(0, 4, 64, 64)
    R = c2w.T.copy()
    return 64, 64, K, R, (-R @ c).reshape(3, 1)


def bench_train(model, diffusion, dev, rank, world, iters=3, B=2):
    """SURVEY 8(f) rank 4, UNet half (not a BASELINE metric): one training step of the production network at the reference's
    microbatch (README.md:104 `--microbatch 2`): GaussianDiffusion.training_losses -> backward through the HIP kernels
    (improved_diffusion/unet_train.py) -> AdamW.  Every rank trains its own replica (no gradient exchange is measured here)."""
    was_training = model.training
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.0, fused=os.environ.get('HL_ADAMW_FUSED', '1') == '1')   # one pass over the 497 M parameters instead of PyTorch's ~8 foreach passes
    g = torch.Generator(device=dev).manual_seed(rank)
    x0 = torch.randn((B, 27, 256, 256), device=dev, generator=g).clamp(-1, 1)
    xc = torch.zeros_like(x0)
    y = torch.zeros((B,), dtype=torch.int64, device=dev)

    def step():
        t = torch.randint(0, 1000, (B,), device=dev, generator=g)
        loss = diffusion.training_losses(model, x0, xc, t, model_kwargs={"y": y})["loss"].mean()
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        return loss

    step()
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(iters):
        loss = step()
    barrier(world)
    dt = max_over_ranks(time.perf_counter() - t0, world, dev) / iters
    assert torch.isfinite(loss.detach()).all()
    # the 16-bit MFMA arithmetic of the convolutions (what the reference's --use_amp True selects; taken automatically under torch.autocast)
    from humanliff_amd.improved_diffusion import unet_train as ut
    dt16 = {}
    for kind in ("bf16", "fp16"):
        ut.set_train_arithmetic(kind)
        try:
            step()
            torch.cuda.synchronize()
            tb = time.perf_counter()
            for _ in range(iters):
                loss16 = step()
            torch.cuda.synchronize()
            dt16[kind] = (time.perf_counter() - tb) / iters
        finally:
            ut.set_train_arithmetic(None)
    assert torch.isfinite(loss16.detach()).all()
    # extension: the whole step (loss -> backward -> optimizer) as ONE HIP graph, replayed (unet_train.GraphedTrainStep): no Python per launch
    graphed = {}
    del loss, loss16          # a live loss keeps the parameters' AccumulateGrad nodes bound to the stream of its step; capture runs on another one
    import gc
    gc.collect()
    for kind in ("fp32", "bf16"):
        ut.set_train_arithmetic(kind)
        try:
            gopt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.0, fused=True, capturable=True)
            tg = torch.randint(0, 1000, (B,), device=dev, generator=g)
            gstep = ut.GraphedTrainStep(diffusion, model, gopt, x0, xc, tg, {"y": y}, warmup=2)
            gstep(x0, xc, tg, {"y": y})
            torch.cuda.synchronize()
            tb = time.perf_counter()
            for _ in range(iters):
                lg = gstep(x0, xc, torch.randint(0, 1000, (B,), device=dev, generator=g), {"y": y})
            torch.cuda.synchronize()
            graphed[kind] = {"ms_per_step": round((time.perf_counter() - tb) / iters * 1e3, 2)}
            assert torch.isfinite(lg).all()
            del gstep, gopt
        finally:
            ut.set_train_arithmetic(None)
    model.train(was_training)
    del opt
    for p in model.parameters():
        p.grad = None
    return {"metric": "UNet training samples/sec", "value": round(world * B / dt, 3), "unit": "samples/s", "ms_per_step": round(dt * 1e3, 2),
            "batch_per_gpu": B, "iterations": iters,
            "bf16_arithmetic": {"ms_per_step": round(dt16["bf16"] * 1e3, 2), "value": round(B / dt16["bf16"], 3), "unit": "samples/s (this rank)"},
            "fp16_arithmetic": {"ms_per_step": round(dt16["fp16"] * 1e3, 2), "value": round(B / dt16["fp16"], 3), "unit": "samples/s (this rank)",
                                "what": "unet_train.set_train_arithmetic('fp16' / 'bf16'), chosen automatically under torch.autocast (train_util.py:214): 16-bit "
                                        "operands / fp32 accumulation on v_mfma_f32_32x32x16 for forward, backward-data (k_conv_h16) and the weight gradients "
                                        "(k_conv_wgrad_h16) of the 3x3 / stride-1 layers; fp32 tensors and master weights; `value` above is the fp32 arithmetic"},
            "hip_graph_step": {"fp32_ms_per_step": graphed["fp32"]["ms_per_step"], "bf16_ms_per_step": graphed["bf16"]["ms_per_step"],
                               "what": "unet_train.GraphedTrainStep: loss + backward + capturable fused AdamW captured once, replayed (extension; NOT used for `value`)"},
            "algorithmic_tflops": round(world * 3 * UNET_GFLOP_PER_SAMPLE_STEP * B / dt / 1e3, 2),
            "config": {"workload": "production F4 UNet, training_losses (MSE) + backward on the HIP kernels + AdamW (torch, fused=True), microbatch 2 (README.md:104)",
                       "flop_count": "3 x 2015.4 GFLOP per sample (forward, backward-data, backward-weights; direct-convolution FLOPs)"}}


def bench_render(args, rank, world, dev):
    from humanliff_amd import _lib, synthetic as syn
    from humanliff_amd.NeRF import Renderer
    H = W = 512
    N = 128
    planes = syn.triplane(seed=11 + rank).to(dev)
    mlp = syn.render_mlp_state(3)
    r = Renderer(use_canonical_space=False, triplane_dim=256, triplane_ch=27, smpl_type='smpl', test=True)
    r.load_state_dict(mlp, strict=False)
    r = r.to(dev)
    tp = {"world_bounds": torch.tensor(syn.WORLD_BOUNDS)[None].to(dev)}
    views = args.views
    rays = [[t.to(dev) for t in syn.orbit_rays((rank * views + v) % 36, 36, H, W)] for v in range(views + 1)]
    gu = torch.Generator(device=dev).manual_seed(5 + rank)
    u = torch.rand((H * W, N), generator=gu, device=dev)

    def one(v):
        ro, rd, nr, fr = rays[v]
        return r.render(tp, None, None, ro[None], rd[None], nr[None], fr[None], planes, N, False, n_samples=N, u=u)

    one(views)   # warm-up view (also packs MLP + planes)
    mine = torch.empty((views, 1, H * W, 3), device=dev)
    gathered = torch.empty((world * views, 1, H * W, 3), device=dev) if world > 1 else None
    barrier(world)
    t0 = time.perf_counter()
    imgs = []
    for v in range(views):
        mine[v] = one(v)["rgb_map"]
        imgs.append(mine[v])
    if world > 1:   # north star: RCCL all-gather of the final images, one collective into the preallocated result
        import torch.distributed as dist
        dist.all_gather_into_tensor(gathered, mine)
    barrier(world)
    secs = max_over_ranks(time.perf_counter() - t0, world, dev)
    assert torch.isfinite(imgs[0]).all()
    # ---- host-inclusive leg: the DROP-IN call, u = None - sample_pdf's uniforms are the reference's (torch.rand of the CPU generator,
    # renderer.py:545), per view: continued on the device from the CPU generator's state (NeRF/cpu_rng.py), and, beside it, drawn on the host ----
    host_incl = None
    if world == 1:
        def one_none(v):
            ro, rd, nr, fr = rays[v]
            return r.render(tp, None, None, ro[None], rd[None], nr[None], fr[None], planes, N, False, n_samples=N, u=None)
        hv = min(views, 8)
        rows = {}
        for host in (False, True):
            r.cpu_uniforms_on_host = host
            torch.manual_seed(5)
            one_none(views)
            torch.cuda.synchronize()
            th0 = time.perf_counter()
            for v in range(hv):
                last = one_none(v)["rgb_map"]
            torch.cuda.synchronize()
            dt = time.perf_counter() - th0
            rows["host_draw" if host else "device_draw"] = {"value": round(hv * H * W / dt / 1e6, 4), "ms_per_view": round(dt * 1e3 / hv, 3)}
            assert torch.isfinite(last).all()
            if host:
                same = torch.equal(last, keep)
            keep = last
        r.cpu_uniforms_on_host = False
        host_incl = {"value": rows["device_draw"]["value"], "unit": "Mrays/s", "views": hv, "ms_per_view": rows["device_draw"]["ms_per_view"],
                     "what": "Renderer.render(..., u=None) per view - the reference's call: sample_pdf's uniforms are torch.rand of the CPU generator (134 MB per "
                             "512x512 view).  Default: the device continues the CPU generator's mt19937 stream bit for bit (hl_mt19937_uniform, beside the "
                             "coarse pass) and the host generator is advanced; host_draw: drawn on the host and uploaded, as the reference does it literally",
                     "host_draw": rows["host_draw"], "images_bit_equal_between_the_two": bool(same)}
    # ---- roofline leg: the four stages of one view (the default evaluate-once schedule) timed with events on the launch stream ----
    L = _lib.lib()
    R = H * W
    T32 = (R + 31) // 32 * 32
    ro, rd, nr, fr = [t.contiguous() for t in rays[0]]
    packed, pp = r._packed_mlp(dev), r._packed_planes(planes[0])
    rec_c, rec_n = torch.empty((T32 * N, 4), device=dev), torch.empty((T32 * N, 4), device=dev)
    z_new = torch.empty(T32 * N, device=dev)
    rgb, acc, dep = torch.empty((R, 3), device=dev), torch.empty(R, device=dev), torch.empty(R, device=dev)
    bd = tp["world_bounds"][0].contiguous()
    p, s = _lib.ptr, _lib.stream_ptr

    def stages(flags):   # the four launches of one view with HIP events on the launch stream
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        for _ in range(2):   # (first pass: warm-up of this mode)
            ev[0].record()
            _lib.check(L.hl_render_eval_products(p(packed), p(pp), 256, 256, p(bd), p(ro), p(rd), p(nr), p(fr), None, 0, R, N, flags, p(rec_c), s()))
            ev[1].record()
            _lib.check(L.hl_render_importance_new(p(rec_c), p(rd), p(nr), p(fr), None, p(u), R, N, N, p(z_new), s()))
            ev[2].record()
            _lib.check(L.hl_render_eval_products(p(packed), p(pp), 256, 256, p(bd), p(ro), p(rd), p(nr), p(fr), p(z_new), 1, R, N, flags, p(rec_n), s()))
            ev[3].record()
            _lib.check(L.hl_render_composite(p(nr), p(fr), None, p(z_new), p(rec_c), p(rec_n), R, N, N, 2, p(rgb), p(acc), p(dep), s()))
            ev[4].record()
            torch.cuda.synchronize()
        return [ev[k].elapsed_time(ev[k + 1]) for k in range(4)]

    eval_flop = R * N * FULL_FLOP_PER_POINT                     # one evaluate launch: full MLP at 128 points per ray
    # default product mode (Renderer.mlp_products = "fp16x2"): k_march_plw<2>.  Its MFMAs: 384 v_mfma_f32_32x32x16_f16 per 32 points and
    # sample (32 chunks x 4 positions x 3 partial products) = 393 216 FLOP per point issued on the 16-bit pipe (dense peak 2.5 PFLOP/s)
    t_a, t_i, t_b, t_c = stages(_lib.HL_RENDER_MLP_FP16X2)
    issued = R * N * 393216.0
    view_ms = t_a + t_i + t_b + t_c
    roof = {"bound": "mfma", "kernel": "k_march_plw<2> (evaluate pass: tri-plane gather + full MLP at 128 depths per ray, every fp32 product as three fp16 partial "
                                       "products of two-plane splits, fp32 accumulation, raw records out), two launches per 512x512 view (coarse depths, importance depths)",
            "products": "fp16x2 (x = h0 + h1 with two fp16 planes, 2^-20 |x|; weight planes of 2^k W per layer, nearest-even, 2^-22 of the layer's largest weight at any "
                        "magnitude; h0 w0 + h0 w1 + h1 w0 on v_mfma_f32_32x32x16_f16, fp32 accumulation) - an fp32-tolerance mode: same test bounds as the fp32-MFMA "
                        "kernel, rgb within 7e-7 of it on full views",
            "achieved": round(issued / (t_b * 1e-3) / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(issued / (t_b * 1e-3) / 1e12 / 2500.0, 4),
            "peak_note": "dense fp16 MFMA peak (MI355X_MICROARCH.md); `achieved` = fp16 FLOPs issued.  In the path's own unit: "
                         f"{eval_flop / (t_b * 1e-3) / 1e12:.1f} TFLOP/s of algorithmic fp32 work = {eval_flop / (t_b * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS:.2f} x the fp32 matrix peak",
            "fp32_equivalent_tflops": round(eval_flop / (t_b * 1e-3) / 1e12, 2), "traffic": PMC_TRAFFIC["k_march_b3w_eval_512x512"],
            "traffic_source": PMC_TRAFFIC["source"] + " (the fp16x2 launch moves the same bytes: same gather, same records)", "launch_ms": round(t_b, 3),
            "view": {"ms": round(view_ms, 3), "eval_coarse_ms": round(t_a, 3), "importance_ms": round(t_i, 3),
                     "eval_importance_ms": round(t_b, 3), "composite_ms": round(t_c, 3),
                     "algorithmic_tflops": round(R * FINE_FLOP_PER_RAY_TOTAL / (view_ms * 1e-3) / 1e12, 2),
                     "note": "algorithmic = the reference's schedule, 128 x 79 616 + 256 x 132 608 FLOP per ray (SURVEY 8(d)); this "
                             "schedule evaluates every point once (256 x 132 608)"}}
    # ---- round 6: the schedule `value` runs on is TWO launches per view - the coarse evaluate (timed above as eval_coarse_ms) and the one-pass fine launch
    # (k_march_plw<2, false, true>: importance depths + their evaluation + depth-ordered compositing); the whole call timed with events, both schedules ----
    ws = r._workspace(L.hl_render_workspace_bytes(R, N, N), dev)

    def whole_view(flags):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = None
        for _ in range(3):
            e0.record()
            _lib.check(L.hl_render_rays(p(packed), p(pp), 256, 256, p(bd), p(ro), p(rd), p(nr), p(fr), None, p(u), R, N, N, flags, p(rgb), p(acc), p(dep), p(ws), s()))
            e1.record()
            torch.cuda.synchronize()
            best = e0.elapsed_time(e1) if best is None else min(best, e0.elapsed_time(e1))
        return best, rgb.clone()
    v2_ms, img2 = whole_view(_lib.HL_RENDER_MLP_FP16X2 | 2)
    v4_ms, img4 = whole_view(_lib.HL_RENDER_MLP_FP16X2 | 2 | _lib.HL_RENDER_FOUR_LAUNCH)
    roof["launches_per_view"] = 2
    roof["hbm_bytes_per_view"] = 3.0e9
    roof["one_pass"] = {"what": "hl_render_rays, default schedule (two launches: coarse evaluate, one-pass fine launch) against HL_RENDER_FOUR_LAUNCH (rounds 2-5: evaluate, "
                                "k_importance, evaluate, k_composite), one 512x512 view each, HIP events on the launch stream, best of 3",
                        "ms_per_view": round(v2_ms, 3), "fine_launch_ms": round(v2_ms - t_a, 3), "four_launch_ms_per_view": round(v4_ms, 3),
                        "images_bit_equal": bool(torch.equal(img2, img4)),
                        "hbm_bytes_per_view": {"one_pass": 3.0e9, "four_launch": 7.0e9,
                                               "source": "profiles/r06_pmc_render_traffic.md (rocprofv3 --pmc FETCH_SIZE x 2 / WRITE_SIZE, separate passes): coarse 168 + 537 MB, "
                                                         "fine launch 2 161 + 143 MB; four launches: 2 x (168 + 537) + 760 + 699 + 4 150 + 5 MB"}}
    del img2, img4
    # the exact-split mode beside it: Renderer.mlp_products = "bf16x3" (k_march_plw<3>: three bf16 planes, six partial products - round 4's default)
    b_a, b_i, b_b, b_c = stages(_lib.HL_RENDER_MLP_BF16X3)
    r.mlp_products = "bf16x3"
    try:
        one(views)
        torch.cuda.synchronize()
        tb_ = time.perf_counter()
        imgs_b3 = [one(v)["rgb_map"] for v in range(views)]
        torch.cuda.synchronize()
        db3 = time.perf_counter() - tb_
    finally:
        r.mlp_products = "fp16x2"
    roof["bf16x3_products"] = {"what": "the same views with Renderer.mlp_products = 'bf16x3' (k_march_plw<3>: EXACT three-way bf16 splits, six partial products)",
                               "value": round(views * R / db3 / 1e6, 4), "unit": "Mrays/s", "ms_per_view": round(db3 * 1e3 / views, 3), "launch_ms": round(b_b, 3),
                               "issued_tflops": round(R * N * 786432.0 / (b_b * 1e-3) / 1e12, 1), "frac_of_bf16_peak": round(R * N * 786432.0 / (b_b * 1e-3) / 1e12 / 2500.0, 4),
                               "rgb_max_abs_vs_fp16x2": max(float((imgs_b3[v][0] - mine[v][0]).abs().max()) for v in range(min(views, 3)))}
    del imgs_b3
    # the native-fp32 figure beside it: Renderer.mlp_products = "fp32" (k_march<true,true,8> on v_mfma_f32_32x32x2_f32)
    f_a, f_i, f_b, f_c = stages(0)
    r.mlp_products = "fp32"
    try:
        one(views)
        torch.cuda.synchronize()
        tf_ = time.perf_counter()
        imgs32 = [one(v)["rgb_map"] for v in range(views)]
        torch.cuda.synchronize()
        d32 = time.perf_counter() - tf_
    finally:
        r.mlp_products = "fp16x2"
    diff = max(float((imgs32[v][0] - mine[v][0]).abs().max()) for v in range(min(views, 3)))
    roof["fp32_products"] = {"what": "the same views with Renderer.mlp_products = 'fp32' (k_march<true,true,8>, v_mfma_f32_32x32x2_f32): the native-fp32 figure",
                             "value": round(views * R / d32 / 1e6, 4), "unit": "Mrays/s", "ms_per_view": round(d32 * 1e3 / views, 3),
                             "launch_ms": round(f_b, 3), "achieved_tflops": round(eval_flop / (f_b * 1e-3) / 1e12, 2),
                             "frac_of_fp32_matrix_peak": round(eval_flop / (f_b * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                             "traffic": PMC_TRAFFIC["k_march_eval_512x512"], "traffic_source": PMC_TRAFFIC["source"],
                             "rgb_max_abs_between_the_two_modes": diff}
    # ---- extract_geometry's density field at the reference's resolution (renderer.py:290-321): 512^3 lattice points
    #      through the tri-plane lookup + density MLP (79 616 FLOP per point), the input of marching cubes ----
    if world == 1:
        r.density_grid(tp, planes, resolution=64)      # warm-up
        torch.cuda.synchronize()
        tg = time.perf_counter()
        grid = r.density_grid(tp, planes, resolution=512)
        torch.cuda.synchronize()
        dg = time.perf_counter() - tg
        roof["density_grid"] = {"resolution": 512, "points": 512 ** 3, "ms": round(dg * 1e3, 2),
                                "tflops": round(512 ** 3 * 79616 / dg / 1e12, 2), "finite": bool(torch.isfinite(grid).all()),
                                "what": "Renderer.density_grid(resolution=512): the field extract_geometry hands to marching cubes, on k_march<false> "
                                        "(coarse density pass), including the host-side launch loop and the untile copies"}
        del grid
        # ---- opt-in: the MLP with fp16 operands / fp32 accumulation (Renderer.mlp_fp16, k_march16); not `value` ----
        ref_imgs = [imgs32[v].clone() for v in range(min(views, 3))]
        r.mlp_fp16 = True
        try:
            one(views)
            torch.cuda.synchronize()
            th_ = time.perf_counter()
            imgs16 = [one(v)["rgb_map"] for v in range(views)]
            torch.cuda.synchronize()
            d16 = time.perf_counter() - th_
        finally:
            r.mlp_fp16 = False
        ps = []
        for v in range(len(ref_imgs)):
            mse = float(((imgs16[v] - ref_imgs[v]) ** 2).mean())
            ps.append(99.0 if mse == 0 else 10 * math.log10(1.0 / mse))
        roof["fp16_mode"] = {"what": "Renderer.mlp_fp16 / HL_RENDER_MLP_FP16 (opt-in): the MLP on v_mfma_f32_32x32x16_f16 - fp16 operands, fp32 accumulation, "
                                     "all weights LDS-resident (k_march16); tri-plane gather, encodings, softplus, importance sampling and compositing "
                                     "stay fp32.  NOT used for `value`",
                             "value": round(views * R / d16 / 1e6, 4), "unit": "Mrays/s", "ms_per_view": round(d16 * 1e3 / views, 3),
                             "psnr_db_vs_fp32_views": [round(x, 1) for x in ps], "finite": bool(torch.isfinite(imgs16[0]).all())}
    roof["host_inclusive"] = host_incl
    return secs, roof, views * R


def bench_fit(args, rank, world, dev, iters=30):
    """one tri-plane fitting iteration at the reference's training configuration
    (recon_NeRF/configs/SynBody.txt: 2 subjects x n_rand 2048 rays x 128+128 stratified samples, density noise, MSE on rgb + 0.1 MSE on
    acc, Adam on MLP and tri-planes; run_nerf_batch.py:236-265).  Every rank fits its own subjects (no exchange)."""
    from humanliff_amd import synthetic as syn
    from humanliff_amd.NeRF import Renderer
    torch.manual_seed(rank)
    r = Renderer(use_canonical_space=False, triplane_dim=256, triplane_ch=27, test=False)
    r.load_state_dict(syn.render_mlp_state(3), strict=False)
    r = r.to(dev)
    tri = torch.nn.Parameter((0.1 * torch.randn((2, 4, 3, 9, 256, 256))).to(dev))
    opt = torch.optim.Adam([{'params': list(r.parameters()), 'lr': 5e-4}, {'params': [tri], 'lr': 1e-2}], betas=(0.9, 0.999),
                           fused=os.environ.get('HL_ADAMW_FUSED', '1') == '1')
    bs, R, N = 2, 2048, 128
    ro, rd, nr, fr = syn.orbit_rays(2, 8, 128, 128)
    pick = torch.nonzero(fr != 1).flatten()
    pick = pick[torch.randperm(pick.numel())[:R]]
    ro, rd, nr, fr = (t[pick].to(dev) for t in (ro, rd, nr, fr))
    target = torch.rand((bs, R, 3), device=dev)
    tp = {"world_bounds": torch.tensor(syn.WORLD_BOUNDS)[None].expand(bs, 2, 3).to(dev)}
    ids, layer = torch.tensor([0, 1], device=dev), torch.tensor([1, 3], device=dev)   # the reference's loop indexes with the batch's device tensors (to_cuda, run_nerf_batch.py:233)
    t = torch.linspace(0., 1., steps=N, device=dev)

    def one():
        z = (nr[:, None] * (1. - t) + fr[:, None] * t)[None].expand(bs, R, N)
        mids = .5 * (z[..., 1:] + z[..., :-1])
        upper, lower = torch.cat([mids, z[..., -1:]], -1), torch.cat([z[..., :1], mids], -1)
        z = lower + (upper - lower) * torch.rand(z.shape, device=dev)
        out = r.render(tp, None, z, ro[None].expand(bs, R, 3), rd[None].expand(bs, R, 3), nr[None, :, None].expand(bs, R, 1),
                       fr[None, :, None].expand(bs, R, 1), tri[ids, layer], N, False)
        loss = ((out["rgb_map"] - target) ** 2).mean() + 0.1 * ((out["acc_map"] - 1.0) ** 2).mean()
        loss.backward()
        opt.step()
        opt.zero_grad()
        return loss

    def timed():
        for _ in range(3):
            one()
        barrier(world)
        t0 = time.perf_counter()
        for _ in range(iters):
            loss = one()
        barrier(world)
        s = max_over_ranks(time.perf_counter() - t0, world, dev)
        assert torch.isfinite(loss.detach()).all()
        return s

    secs = timed()                       # the defaults: sample_pdf's uniforms = the CPU generator's stream continued on the device, subjects on their own streams
    r.subject_streams = False            # one stream for all subjects (rounds 1-4's default)
    secs_one_stream = timed()
    r.subject_streams = True
    r.uniforms_on_device = True          # extension: the device generator's own uniforms (another random stream)
    secs_dev_streams = timed()
    r.uniforms_on_device = False
    pts = bs * R * 2 * N
    stages = fit_stage_times(r, tri[0, 0].detach(), tp["world_bounds"][0].contiguous(), ro, rd, nr, fr, N, dev)
    return {"stages_ms_per_subject": stages["ms"], "roofline": stages["roofline"], "metric": "fitting-iterations/sec", "value": round(world * iters / secs, 2), "unit": "it/s", "ms_per_iteration": round(secs * 1e3 / iters, 3),
            "sample_points_per_sec": round(world * iters * pts / secs), "iterations": iters,
            "one_stream": {"value": round(world * iters / secs_one_stream, 2), "unit": "it/s", "ms_per_iteration": round(secs_one_stream * 1e3 / iters, 3),
                           "what": "Renderer.subject_streams = False: every subject on the caller's stream (rounds 1-4's default; same bits)"},
            "uniforms_on_device": {"value": round(world * iters / secs_dev_streams, 2), "unit": "it/s", "ms_per_iteration": round(secs_dev_streams * 1e3 / iters, 3),
                                   "what": "Renderer.uniforms_on_device = True: sample_pdf's uniforms from the device generator's own stream (same distribution, other numbers; NOT used for `value`)"},
            "config": {"workload": "recon_NeRF SynBody training step: 2 subjects x 2048 rays x (128+128) samples, 256x256x27 tri-planes, "
                                   "forward + HIP backward + Adam", "sample_points_per_iteration": pts}}


def fit_stage_times(r, planes, bounds, ro, rd, nr, fr, N, dev):
    """One subject's forward + backward through the C ABI stage by stage (the calls of NeRF/train.py), HIP events between them."""
    import ctypes as C
    from humanliff_amd import _lib
    from humanliff_amd.NeRF.renderer import untile_rows
    from humanliff_amd.NeRF.train import _row_pad, train_rows
    L = _lib.lib()
    p, st = _lib.ptr, _lib.stream_ptr()
    R = ro.shape[0]
    H, W = planes.shape[-2:]
    T32 = (R + 31) // 32 * 32
    P = T32 * 2 * N
    act_rows, del_rows = train_rows()
    packed, pp = r._packed_mlp(dev), r._packed_planes(planes)
    t = torch.linspace(0., 1., steps=N, device=dev)
    z = (nr[:, None] * (1. - t) + fr[:, None] * t).contiguous()
    u = torch.rand((R, N), device=dev)
    noise = torch.randn((R, 2 * N), device=dev)
    g_rgb, g_acc = torch.randn((R, 3), device=dev) / R, torch.randn((R,), device=dev) / R
    e = lambda n: torch.empty(n, dtype=torch.float32, device=dev)  # noqa: E731
    LD = P + _row_pad()          # row pitch of the two matrices, as NeRF/train.py allocates them
    act, delta = e((act_rows, LD)), e((del_rows, LD))
    vc, vn, zn, d_rec = e(T32 * N * 4), e(T32 * N * 4), e(T32 * N), e((P, 4))
    rgb, acc, dep = e((R, 3)), e(R), e(R)
    scratch = e(L.hl_render_composite_backward_scratch_bytes(R, N, N) // 4)
    mlp = r._mlp_tensors()
    params = _lib.RenderMlpParams(*[C.c_void_p(t_.data_ptr()) for t_ in mlp])
    bwd = e(L.hl_render_mlp_bwd_packed_bytes() // 4)
    flat = torch.zeros(sum(t_.numel() for t_ in mlp), device=dev)
    grads, o = [], 0
    for t_ in mlp:
        grads.append(flat[o:o + t_.numel()])
        o += t_.numel()
    gp = _lib.RenderMlpParams(*[C.c_void_p(g.data_ptr()) for g in grads])
    d_planes = e((27, H, W))
    ro, rd, nr, fr, bd = ro.contiguous(), rd.contiguous(), nr.contiguous(), fr.contiguous(), bounds
    calls = [
        ("eval_acts_coarse", lambda: L.hl_render_eval_acts(p(packed), p(pp), H, W, p(bd), p(ro), p(rd), p(nr), p(fr), p(z), 0, R, N, p(vc), p(act), LD, 0, st)),
        ("importance", lambda: L.hl_render_importance_new(p(vc), p(rd), p(nr), p(fr), p(z), p(u), R, N, N, p(zn), st)),
        ("eval_acts_new", lambda: L.hl_render_eval_acts(p(packed), p(pp), H, W, p(bd), p(ro), p(rd), p(nr), p(fr), p(zn), 1, R, N, p(vn), p(act), LD, T32 * N, st)),
        ("composite", lambda: L.hl_render_composite_noise(p(nr), p(fr), p(z), p(zn), p(vc), p(vn), p(noise), R, N, N, 2, p(rgb), p(acc), p(dep), st)),
        ("composite_backward", lambda: L.hl_render_composite_backward(p(nr), p(fr), p(z), p(zn), p(vc), p(vn), p(noise), p(g_rgb), p(g_acc), R, N, N, 2,
                                                                      p(d_rec[:T32 * N]), p(d_rec[T32 * N:]), p(delta), LD, p(scratch), st)),
        ("pack_bwd", lambda: L.hl_render_mlp_pack_bwd(C.byref(params), p(bwd), st)),
        ("mlp_backward_coarse", lambda: L.hl_render_mlp_backward(p(packed), p(bwd), H, W, p(bd), p(ro), p(rd), p(nr), p(fr), p(z), 0, R, N, p(d_rec[:T32 * N]),
                                                                 p(act), LD, 0, p(delta), LD, 0, st)),
        ("mlp_backward_new", lambda: L.hl_render_mlp_backward(p(packed), p(bwd), H, W, p(bd), p(ro), p(rd), p(nr), p(fr), p(zn), 1, R, N, p(d_rec[T32 * N:]),
                                                              p(act), LD, T32 * N, p(delta), LD, T32 * N, st)),
        ("plane_grads", lambda: L.hl_render_plane_grads(H, W, p(bd), p(ro), p(rd), p(nr), p(fr), p(z), p(untile_rows(zn, R, N).contiguous()), 1, R, N, N,
                                                        p(delta), LD, p(d_planes), p(pg_scratch), st)),
        ("weight_grads", lambda: L.hl_render_weight_grads(p(delta), LD, p(act), LD, P, C.byref(gp), p(wg_scratch), st)),
    ]
    pg_scratch = torch.empty(L.hl_render_plane_grads_scratch_bytes(R) // 4, dtype=torch.float32, device=dev)
    wg_scratch = torch.empty(L.hl_render_weight_grads_scratch_bytes(P) // 4, dtype=torch.float32, device=dev)
    ms = {}
    for rep in range(2):          # second round is the measurement
        for name, fn in calls:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            _lib.check(fn(), name)
            b.record()
            torch.cuda.synchronize()
            ms[name] = round(a.elapsed_time(b), 4)
    # dominant kernel: k_wgrad, bound by reading the two matrices: the seven products touch 1 365 rows of P floats
    wg_bytes = 1365 * P * 4
    ach = wg_bytes / (ms["weight_grads"] * 1e-3) / 1e9
    roof = {"bound": "hbm", "kernel": "k_wgrad (14 parameter gradients = delta rows x activation rows^T over the sample points of one subject)",
            "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": 2863.1e6,
            "traffic_source": "profiles/r01_pmc_fit_traffic.md (rocprofv3 --pmc FETCH_SIZE, corrected x2; = the algorithmic 2.86 GB)",
            "launch_ms": ms["weight_grads"], "mfma_tflops": round(2 * 66304 * P / (ms["weight_grads"] * 1e-3) / 1e12, 2)}
    return {"ms": ms, "roofline": roof}


def cpu_baseline_fit(threads, n_rays=512):
    """Oracle autograd (PyTorch-CPU) of the same loss on a bounded sample of rays; scaled to iterations of 4096 rays."""
    from humanliff_amd import synthetic as syn
    from oracle import render_oracle as ro
    torch.set_num_threads(threads)
    planes = syn.triplane(seed=11)[0].clone().requires_grad_(True)
    mlp = {k: v.clone().requires_grad_(True) for k, v in syn.render_mlp_state(3).items()}
    o, d, nr, fr = syn.orbit_rays(2, 8, 128, 128)
    pick = torch.nonzero(fr != 1).flatten()[:n_rays]
    o, d, nr, fr = o[pick], d[pick], nr[pick], fr[pick]
    N = 128
    t = torch.linspace(0., 1., steps=N)
    z = nr[:, None] * (1. - t) + fr[:, None] * t
    u, noise = torch.rand(n_rays, N), torch.randn(n_rays, 2 * N)
    dt = []
    for _ in range(2):
        t0 = time.perf_counter()
        rgb, acc, _ = ro.render_rays(mlp, planes, torch.tensor(syn.WORLD_BOUNDS), o, d, nr, fr, N, N, u=u, z_vals=z, noise=noise)
        ((rgb ** 2).mean() + 0.1 * ((acc - 1.0) ** 2).mean()).backward()
        dt.append(time.perf_counter() - t0)
    return {"value": round(n_rays / 4096 / dt[-1], 4), "unit": "it/s", "cores": threads, "kind": "port",
            "sample": f"oracle (PyTorch-CPU fp32 restatement) forward + autograd backward of {n_rays} rays at 128+128 samples (second of two runs), "
                      "scaled to the 4096 rays of an iteration; optimizer not included"}


def cpu_baseline_unet(sd, threads, model=None, dev=None):
    """Oracle UNet forward + DDPM update on the host: B=1 (1 warm-up + 3 timed steps, ~13 s) and B=4 - the configuration `value` is
    quoted on - (1 timed step, ~13 s; a B=4 1000-step run would take hours).  With `model` the same four B=1 steps are run on the GPU
    on the same x_T / noise and compared (`parity`)."""
    from oracle import diffusion_oracle as do
    from oracle import unet_oracle as uo
    torch.set_num_threads(threads)
    s = do.Schedule(do.linear_betas(1000), list(range(1000)))
    g = torch.Generator().manual_seed(7)
    x = x_T = torch.randn((1, 27, 256, 256), generator=g)
    xc = torch.zeros_like(x)
    y = torch.zeros((1,), dtype=torch.int64)
    n_timed = 3
    noises = []
    with torch.no_grad():
        for i in range(1 + n_timed):
            if i == 1:
                t0 = time.perf_counter()
            t = torch.tensor([999 - i])
            eps = uo.unet_forward(sd, x, t, xc, y)
            noises.append(torch.randn(x.shape, generator=g))
            x, _ = do.p_sample_step(s, x, t, eps, noises[-1])
        dt = time.perf_counter() - t0
        x4 = torch.randn((4, 27, 256, 256), generator=g)
        t4 = time.perf_counter()
        eps4 = uo.unet_forward(sd, x4, torch.full((4,), 500), torch.zeros_like(x4), torch.zeros((4,), dtype=torch.int64))
        do.p_sample_step(s, x4, torch.full((4,), 500), eps4, torch.randn(x4.shape, generator=g))
        dt4 = time.perf_counter() - t4
    out = {"value": round(n_timed / dt, 4), "unit": "denoise-steps/sec", "cores": threads, "kind": "port",
           "sample": f"oracle (PyTorch-CPU fp32 restatement) p_sample, production UNet, batch 1, {n_timed} timed steps after 1 warm-up",
           "batch4": {"value": round(4 / dt4, 4), "unit": "denoise-steps/sec", "sample": "same, batch 4 (the configuration `value` is quoted on), 1 timed step"}}
    parity = None
    if model is not None:
        # the headline dispatch: the B=4 forward of the oracle above against the HIP forward on the same inputs (kernel selection depends
        # on the batch size - at B=4 the 128-pixel level runs k_conv_wino4, at B=1 it does not; the census says which kernels ran)
        with torch.no_grad():
            got4 = model(x4.to(dev), torch.full((4,), 500, device=dev), torch.zeros_like(x4).to(dev), y=torch.zeros((4,), dtype=torch.int64, device=dev)).cpu()
        census = model.dispatch_census()
        b4 = {"max_abs": float((got4 - eps4).abs().max()), "psnr_db": round(_psnr(got4, eps4), 2), "output_abs_mean": float(eps4.abs().mean()),
              "dispatch": {k: v[:6] for k, v in census.items() if any(v)},
              "what": "one forward of the production net at B=4 (t=500), HIP vs the oracle; dispatch = conv launches per kernel family and "
                      "resolution level (256, 128, 64, 32, 16, 8 pixels)"}
        from humanliff_amd.improved_diffusion.script_util import create_gaussian_diffusion
        d = create_gaussian_diffusion(steps=1000, timestep_respacing="")
        k = {"i": 0}
        orig = torch.randn_like

        def inj(ref):
            k["i"] += 1
            return noises[k["i"] - 1].to(ref.device)
        torch.randn_like = inj
        try:
            xg = x_T.to(dev)
            with torch.no_grad():
                for i in range(1 + n_timed):
                    xg = d.p_sample(model, xg, xc.to(dev), torch.tensor([999 - i], device=dev), model_kwargs={"y": y.to(dev)})["sample"]
        finally:
            torch.randn_like = orig
        parity = {"steps": 1 + n_timed, "max_abs": float((xg.cpu() - x).abs().max()), "psnr_db": round(_psnr(xg.cpu(), x), 2),
                  "value_scale": float(x.abs().max()),
                  "what": "x after 4 recurrent p_sample steps (t = 999..996) of the production net, B=1, HIP vs the oracle on identical x_T / noise",
                  "forward_b4": b4}
    return out, parity


def cpu_baseline_render(threads, n_rays=16384, dev=None):
    """Oracle render of a bounded sample of one 512x512 view; with `dev` the same rays / uniforms go through the HIP renderer and the
    two images are compared (`parity`: PSNR / max-abs, colours in [0,1])."""
    from humanliff_amd import synthetic as syn
    from oracle import render_oracle as ro
    torch.set_num_threads(threads)
    planes = syn.triplane(seed=11)
    mlp = syn.render_mlp_state(3)
    o, d, nr, fr = syn.orbit_rays(0, 36, 512, 512)
    sl = slice(512 * 256, 512 * 256 + n_rays)
    u = syn.importance_u(n_rays, 128, seed=5)
    t0 = time.perf_counter()
    with torch.no_grad():
        rgb, acc, depth = ro.render_rays(mlp, planes[0], torch.tensor(syn.WORLD_BOUNDS), o[sl], d[sl], nr[sl], fr[sl], 128, 128, u=u)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    with torch.no_grad():       # the 4 096-ray point of SURVEY 8(d) (the reference's own chunking granularity class)
        ro.render_rays(mlp, planes[0], torch.tensor(syn.WORLD_BOUNDS), o[sl][:4096], d[sl][:4096], nr[sl][:4096], fr[sl][:4096], 128, 128, u=u[:4096])
    dt4 = time.perf_counter() - t1
    out = {"value": round(n_rays / dt / 1e6, 6), "unit": "Mrays/sec", "cores": threads, "kind": "port",
           "sample": f"oracle (PyTorch-CPU fp32 restatement) render of {n_rays} rays of one 512x512 view at 128+128 samples",
           "rays4096": {"value": round(4096 / dt4 / 1e6, 6), "unit": "Mrays/sec", "sample": "same, 4 096 rays"}}
    parity = None
    if dev is not None:
        from humanliff_amd.NeRF import Renderer
        r = Renderer(use_canonical_space=False, triplane_dim=256, triplane_ch=27, smpl_type='smpl', test=True)
        r.load_state_dict(mlp, strict=False)
        r = r.to(dev)
        got = r.render({"world_bounds": torch.tensor(syn.WORLD_BOUNDS)[None].to(dev)}, None, None, o[sl][None].to(dev), d[sl][None].to(dev),
                       nr[sl][None].to(dev), fr[sl][None].to(dev), planes.to(dev), 128, False, n_samples=128, u=u.to(dev))
        parity = {"rays": n_rays, "psnr_db": round(_psnr(got["rgb_map"][0].cpu(), rgb), 2),
                  "max_abs": float((got["rgb_map"][0].cpu() - rgb).abs().max()), "acc_max_abs": float((got["acc_map"][0].cpu() - acc).abs().max()),
                  "depth_max_abs": float((got["depth_map"][0].cpu() - depth).abs().max()),
                  "what": "rgb of the same 16 384 rays / uniforms at 128+128 samples, HIP vs the oracle (north-star bar: PSNR >= 45 dB)"}
    return out, parity


