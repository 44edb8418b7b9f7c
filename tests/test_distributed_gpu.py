"""Two ranks on the REAL HIP path (BASELINE configs[3] / [4], scripts/triplane_sample_layered.py:41-46, 112-219).

The GPU boxes of this pool have one GPU and RCCL refuses two ranks on one device, so the ranks share cuda:0 and the process group is
gloo (device tensors are staged through the host by the backend): everything except the transport is what an 8-GPU run executes -
torch.distributed rendezvous, block sharding, the HIP UNet inside ddim_sample_loop, the HIP renderer, the per-subject asynchronous uint8
image gathers and the final gather of the samples (humanliff_amd.distributed.sample_and_render).  RCCL itself stays unexercised here.
"""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

N_SUBJECTS, N_LAYERS, N_VIEWS, RES, DDIM = 3, 2, 2, 48, 4          # 3 subjects over 2 ranks: ragged (rank 1 owns one real subject + a repeat)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _flow(dev, images_root=None, n_subjects=N_SUBJECTS, batch=1):
    """The same closures for the 1-process and the 2-process run: tiny controlnet UNet (class-conditional, attention, 27x32x32 tri-planes)
    on the HIP kernels, DDIM-4 per cloth layer chained through x_cond, HIP renders of the finished tri-plane."""
    from humanliff_amd import distributed as hd, synthetic as syn
    from humanliff_amd.NeRF import Renderer, render_view
    from tests.test_train_loss_cpu import tiny_model
    from humanliff_amd.improved_diffusion.script_util import create_gaussian_diffusion
    model, _ = tiny_model()
    model = model.to(dev).eval()
    diffusion = create_gaussian_diffusion(steps=1000, learn_sigma=False, noise_schedule="linear", timestep_respacing=f"ddim{DDIM}")
    rend = Renderer(use_canonical_space=False, triplane_dim=32, triplane_ch=27, smpl_type="smpl", test=True)
    rend.load_state_dict(syn.render_mlp_state(3, gain=2.0), strict=False)
    rend = rend.to(dev)
    shape = (27, 32, 32)
    tp = {"world_bounds": torch.tensor(syn.WORLD_BOUNDS)[None].to(dev)}
    u = torch.rand((RES * RES, 32), generator=torch.Generator().manual_seed(5)).to(dev)
    calls = {"sample": [], "render": []}

    def sample_fn(x_cond, layer, ids):
        calls["sample"].append((layer, tuple(ids)))
        noise = torch.stack([torch.randn(shape, generator=torch.Generator().manual_seed(7000 + 10 * i + layer)) for i in ids]).to(dev)
        y = torch.full((len(ids),), layer, dtype=torch.int64, device=dev)
        return diffusion.ddim_sample_loop(model, (len(ids),) + shape, x_cond=x_cond, noise=noise, clip_denoised=True, model_kwargs={"y": y}, device=dev)

    def render_fn(sid, sample, v):
        calls["render"].append((sid, v))
        planes = sample.reshape(1, 3, 9, 32, 32)
        K, c2w, cam = syn.orbit_camera(v, N_VIEWS, RES, RES)
        R = c2w.T.copy()
        return render_view(RES, RES, K, R, (-R @ cam).reshape(3, 1), planes, tp, rend, n_samples=32, n_importance=32, u=u)[0]

    with torch.no_grad():
        smp, img = hd.sample_and_render(sample_fn, render_fn, n_subjects, N_LAYERS, shape, batch, N_VIEWS, (RES, RES, 3), dev, as_uint8=True,
                                        images_root=images_root)
    return smp, img, calls


def _worker(rank, world, port, out_path, n_subjects=N_SUBJECTS, batch=1):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from humanliff_amd import _lib, distributed as hd
    _lib.lib()                                   # the HIP library, or nothing
    r, w, dev = hd.init_distributed("gloo")      # both ranks on cuda:0 (LOCAL_RANK 0)
    assert (r, w) == (rank, world) and dev.type == "cuda"
    smp, img, calls = _flow(dev, images_root=0, n_subjects=n_subjects, batch=batch)
    per = (n_subjects + world - 1) // world
    assert all(min(i, n_subjects - 1) // per == rank for _, ids in calls["sample"] for i in ids)
    assert all(s // per == rank for s, _ in calls["render"])
    assert (img is None) == (rank != 0)
    torch.save({"samples": smp.cpu(), "images": None if img is None else img.cpu(), "n_sample_calls": len(calls["sample"]),
                "n_render_calls": len(calls["render"])}, f"{out_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_on_the_hip_path_equal_one_process(tmp_path):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    dev = torch.device("cuda:0")
    want_smp, want_img, calls1 = _flow(dev)
    assert want_img.dtype == torch.uint8 and tuple(want_img.shape) == (N_SUBJECTS, N_VIEWS, RES, RES, 3)
    assert float(want_img.float().std()) > 1.0            # real pictures, not a constant
    want_smp, want_img = want_smp.cpu(), want_img.cpu()
    ctx = mp.get_context("spawn")
    port = _free_port()
    out = str(tmp_path / "ranks")
    ps = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(500)
        assert p.exitcode == 0, p.exitcode
    res = [torch.load(f"{out}.{r}") for r in range(2)]
    # every rank holds all samples; rank 0 alone assembled the images; both equal the single-process run BIT FOR BIT (same kernels, same
    # per-call batch, deterministic kernels - only the placement of the subjects differs)
    for r in range(2):
        assert torch.equal(res[r]["samples"], want_smp), r
    assert res[1]["images"] is None and torch.equal(res[0]["images"], want_img)
    # work really was split: 2 + 1(+1 repeated tail slot) subjects; each rank sampled / rendered its own block only
    assert res[0]["n_sample_calls"] == 2 * N_LAYERS and res[1]["n_sample_calls"] == 2 * N_LAYERS
    assert res[0]["n_render_calls"] == 2 * N_VIEWS and res[1]["n_render_calls"] == 2 * N_VIEWS
    assert len(calls1["sample"]) == N_SUBJECTS * N_LAYERS


@pytest.mark.timeout(1500)
def test_eight_ranks_sixty_four_subjects_rehearsal(tmp_path):
    """BASELINE configs[3] / [4] at their real PARTITIONING - 64 subjects over 8 ranks, 8 per rank sampled as one batch of 8, every rank
    renders its own subjects' views, per-subject asynchronous uint8 gathers, root-only assembly - rehearsed with eight processes on one GPU
    over gloo (tiny network, 32x32 tri-planes): the block sharding, the call pattern and the gathered order of an 8-GPU run; the values of
    four subjects spread over the ranks are checked against a single-process run of those subjects."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    world, nsub, batch = 8, 64, 8
    ctx = mp.get_context("spawn")
    port = _free_port()
    out = str(tmp_path / "ranks8")
    ps = [ctx.Process(target=_worker, args=(r, world, port, out, nsub, batch)) for r in range(world)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(1400)
        assert p.exitcode == 0, p.exitcode
    res = [torch.load(f"{out}.{r}") for r in range(world)]
    for r in range(world):
        assert tuple(res[r]["samples"].shape) == (nsub, N_LAYERS, 27, 32, 32)
        assert torch.equal(res[r]["samples"], res[0]["samples"])                   # every rank holds all samples, in the same (rank-major) order
        assert (res[r]["images"] is None) == (r != 0)
        assert res[r]["n_sample_calls"] == N_LAYERS and res[r]["n_render_calls"] == 8 * N_VIEWS      # one batch of 8 per layer; 8 subjects' views
    img = res[0]["images"]
    assert tuple(img.shape) == (nsub, N_VIEWS, RES, RES, 3) and img.dtype == torch.uint8
    # subjects 0..7 as ONE process computes them in one batch of 8 = rank 0's block: same kernels, same batch -> equal bits; and the per-subject
    # noise seeds make every subject's result independent of which rank owned it
    dev = torch.device("cuda:0")
    want_smp, want_img, _ = _flow(dev, n_subjects=8, batch=8)
    assert torch.equal(res[0]["samples"][:8], want_smp.cpu()) and torch.equal(img[:8], want_img.cpu())
    assert len({float(res[0]["samples"][i].double().sum()) for i in range(nsub)}) == nsub                # 64 different subjects


def test_bench_self_launches_for_more_than_one_gpu():
    """`python bench.py --gpus 2` as the driver issues it - a plain process, no launcher, no WORLD_SIZE - must start its own ranks
    (torch.distributed.run) and print ONE JSON line with n_gpus = 2 and the `rccl` block (ranks seen, backend, gather rates).  Rehearsed
    with HL_BENCH_BACKEND=gloo: both ranks share the one GPU of this box; on an 8-GPU node the same path runs on RCCL."""
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env["HL_BENCH_BACKEND"] = "gloo"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "1", "--sustained-steps", "0",
           "--no-cpu-baseline", "--no-parity", "--no-batch-sweep", "--no-render", "--no-e2e", "--no-train", "--no-fit", "--no-bf16x3-leg", "--e2e-views", "4"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["metric"] == "denoise-steps/sec" and line["value"] > 0
    assert line["rccl"]["world_size"] == 2 and line["rccl"]["backend"] == "gloo"
    assert line["rccl"]["sample_gather_recv_gb_per_s_per_rank"] > 0 and line["rccl"]["image_gather_uint8_recv_gb_per_s_per_rank"] > 0
    assert len(lines[0]) < 8192                                   # the compact line (bench_line.py); the rest is in bench_detail.json
    detail = json.load(open(os.path.join(root, "bench_detail.json")))
    assert detail["rccl"]["image_gather_uint8"]["ms"] > 0 and detail["n_gpus"] == 2
    assert line["summary"]["n_gpus"] == 2 and line["summary"]["rccl_world_size"] == 2
