"""One process per GPU; subjects / views are sharded, nothing is exchanged inside the hot paths.

The reference launches one process per GPU with torch.distributed.launch and only all-gathers the finished
samples (scripts/triplane_sample_layered.py:41-46, 211-219).  Same here: backend "nccl" (= RCCL over xGMI on
ROCm) on GPUs, "gloo" on CPU-only hosts (tests).  The gather is a single all_gather of equally sized shards:
on the fully connected xGMI mesh RCCL moves each shard to its 7 peers over 7 links in parallel.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun). Returns (rank, world, device)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    if use_gpu:
        torch.cuda.set_device(local)
    device = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend or ("nccl" if use_gpu else "gloo"), rank=rank, world_size=world)
    return rank, world, device


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_indices(n_items, rank=None, world=None):
    """Subject s runs on rank s mod world; every rank gets ceil(n/world) slots so the
    final all_gather has equal shards - the tail ranks repeat their last subject and the copy is dropped."""
    if rank is None:
        rank, world = world_info()
    per = (n_items + world - 1) // world
    idx = [min(rank + k * world, n_items - 1) for k in range(per)] if n_items > 0 else []
    valid = [rank + k * world < n_items for k in range(per)]
    return idx, valid


def gather_shards(local, n_items):
    """all_gather equally shaped per-rank tensors (per, ...) and restore the global subject order (n_items, ...)."""
    rank, world = world_info()
    if world == 1:
        return local[:n_items]
    parts = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(parts, local.contiguous())
    stacked = torch.stack(parts, dim=1)            # (per, world, ...): slot k of rank r is item r + k*world
    return stacked.reshape(-1, *local.shape[1:])[:n_items]


def sample_layered_sharded(sample_fn, n_subjects, n_layers, shape, batch, device):
    """Layer-conditioned sampling of many subjects (BASELINE configs[3]): a subject's layers are sequential
    (layer k is conditioned on its layer k-1 sample, triplane_sample_layered.py:124-134), subjects are independent.

    sample_fn(x_cond (b,C,H,W), layer int, subject_ids list[int]) -> (b,C,H,W)   e.g. a closure over
    diffusion.ddim_sample_loop.  Returns, on every rank, a tensor (n_subjects, n_layers, C, H, W).
    """
    idx, _ = shard_indices(n_subjects)
    out = torch.empty((len(idx), n_layers) + tuple(shape), dtype=torch.float32, device=device)
    for s0 in range(0, len(idx), batch):
        ids = idx[s0:s0 + batch]
        x_cond = torch.zeros((len(ids),) + tuple(shape), dtype=torch.float32, device=device)
        for layer in range(n_layers):
            x_cond = sample_fn(x_cond, layer, ids)
            out[s0:s0 + len(ids), layer] = x_cond
    return gather_shards(out, n_subjects)


def render_views_sharded(render_fn, n_views, image_shape, device):
    """Render-only sharding (BASELINE configs[2]): view v on rank v mod world, one all_gather of the images."""
    idx, _ = shard_indices(n_views)
    imgs = torch.empty((len(idx),) + tuple(image_shape), dtype=torch.float32, device=device)
    for k, v in enumerate(idx):
        imgs[k] = render_fn(v)
    return gather_shards(imgs, n_views)
