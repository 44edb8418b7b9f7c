"""One process per GPU; subjects / views are sharded, nothing is exchanged inside the hot paths.

The reference launches one process per GPU with torch.distributed.launch and only all-gathers the finished
samples (scripts/triplane_sample_layered.py:41-46, 211-219).  Same here: backend "nccl" (= RCCL over xGMI on
ROCm) on GPUs, "gloo" on CPU-only hosts (tests).

Sharding is by contiguous blocks - rank r owns items [r*per, (r+1)*per), per = ceil(n / world) - which is the order the
reference's gather produces (`all_images.extend(gathered_samples)`, rank-major, :211-213).  The gather is ONE
all_gather_into_tensor of equally sized shards straight into the (world*per, ...) result: no list of per-rank tensors, no
torch.stack, no reorder copy (round 1 kept s mod world and paid two extra copies of a 4.66 GB/GPU image shard).  On the fully
connected xGMI mesh RCCL moves each shard to its 7 peers over 7 links in parallel.  Images can travel as uint8 (a quarter of the
bytes; what the reference writes to PNG / MP4 anyway, :186-199), and per-subject gathers can be issued asynchronously so that the
gather of subject i overlaps the render of subject i+1 (ImageGather).
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
    """Items of this rank: the contiguous block [rank*per, (rank+1)*per), per = ceil(n / world).  Every rank gets `per` slots so the
    final all_gather has equal shards - slots past the end repeat the last item and are dropped after the gather (`valid` marks them)."""
    if rank is None:
        rank, world = world_info()
    per = (n_items + world - 1) // world
    idx = [min(rank * per + k, n_items - 1) for k in range(per)] if n_items > 0 else []
    valid = [rank * per + k < n_items for k in range(per)]
    return idx, valid


def to_uint8(images):
    """[0,1] float images -> uint8 exactly like the reference's writers ((np.clip(x, 0, 1) * 255).astype(np.uint8),
    triplane_sample_layered.py:197: truncation, not rounding)."""
    return (images.clamp(0.0, 1.0) * 255.0).to(torch.uint8)


def gather_shards(local, n_items, as_uint8=False, out=None):
    """all-gather equally shaped per-rank blocks (per, ...) into the global order (n_items, ...): one collective writing straight
    into the result buffer (`out`, (world*per, ...), may be preallocated and reused).  as_uint8: convert float images first."""
    rank, world = world_info()
    if as_uint8 and local.dtype != torch.uint8:
        local = to_uint8(local)
    if world == 1:
        return local[:n_items]
    local = local.contiguous()
    per = local.shape[0]
    if out is None:
        out = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    assert out.shape[0] == world * per and out.dtype == local.dtype and out.is_contiguous()
    dist.all_gather_into_tensor(out, local)
    return out[:n_items]


class ImageGather:
    """Per-item asynchronous gathers: after this rank finishes item k of its block (e.g. all views of its k-th subject) it calls
    put(k, tensor); the collective runs on the communicator's stream while the caller goes on rendering item k+1.  result() waits for
    everything and returns the global (n_items, ...) tensor.  All ranks must put the same k in the same order (they do: same `per`).

    Layout: one buffer (per, world, ...) - the gather of slot k fills [k] - and the result is its (world, per) transpose flattened,
    i.e. global index rank*per + k, materialised once at the end (skipped for world == 1)."""

    def __init__(self, n_items, item_shape, dtype, device, as_uint8=False):
        self.rank, self.world = world_info()
        self.n, self.per = n_items, (n_items + self.world - 1) // self.world
        self.as_uint8 = as_uint8
        self.dtype = torch.uint8 if as_uint8 else dtype
        self.buf = torch.empty((self.per, self.world) + tuple(item_shape), dtype=self.dtype, device=device)
        self.pending = []

    def put(self, k, item):
        if self.as_uint8 and item.dtype != torch.uint8:
            item = to_uint8(item)
        item = item.contiguous()
        if self.world == 1:
            self.buf[k, 0].copy_(item)
            return
        dst = self.buf[k].view((-1,) + tuple(item.shape[1:])) if item.dim() > 0 else self.buf[k]
        self.pending.append((dist.all_gather_into_tensor(dst, item, async_op=True), item))   # keep `item` alive until the wait

    def result(self, root=None):
        """Waits for the gathers; returns the global (n_items, ...) tensor.  root: only that rank materialises it (the transposed copy
        of the whole buffer - 9 GB of uint8 for 64 subjects x 185 views x 512^2 - is what only the rank that writes the images needs,
        rank 0 in the reference, triplane_sample_layered.py:214-219); the other ranks get None."""
        for w, _ in self.pending:
            w.wait()
        self.pending = []
        if root is not None and self.rank != root:
            return None
        if self.world == 1:
            return self.buf[:, 0][:self.n]
        return self.buf.transpose(0, 1).reshape((self.world * self.per,) + tuple(self.buf.shape[2:]))[:self.n]


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


def render_views_sharded(render_fn, n_views, image_shape, device, as_uint8=False):
    """Render-only sharding (BASELINE configs[2]): a contiguous block of views per rank, one all_gather of the images."""
    idx, _ = shard_indices(n_views)
    imgs = torch.empty((len(idx),) + tuple(image_shape), dtype=torch.float32, device=device)
    for k, v in enumerate(idx):
        imgs[k] = render_fn(v)
    return gather_shards(imgs, n_views, as_uint8=as_uint8)


def sample_and_render(sample_fn, render_fn, n_subjects, n_layers, shape, batch, n_views, image_shape, device, as_uint8=True, images_root=None):
    """The end-to-end flow of BASELINE configs[3] / [4] (scripts/triplane_sample_layered.py:112-213) over the ranks it runs on:
    every rank samples the layers of its block of subjects (sample_fn as in sample_layered_sharded), renders all views of each of its
    finished subjects - render_fn(subject_id, final_sample (C,H,W), view) -> image_shape tensor in [0,1] - and hands each subject's
    views to an asynchronous gather that overlaps the next subject's renders.  One gather of the samples, per-subject gathers of the
    images (uint8 by default).  Returns (samples (n_subjects, n_layers, C, H, W), images (n_subjects, n_views, *image_shape));
    images_root = r: only rank r assembles the image tensor, the others return None for it (ImageGather.result)."""
    idx, _ = shard_indices(n_subjects)
    local = torch.empty((len(idx), n_layers) + tuple(shape), dtype=torch.float32, device=device)
    for s0 in range(0, len(idx), batch):
        ids = idx[s0:s0 + batch]
        x_cond = torch.zeros((len(ids),) + tuple(shape), dtype=torch.float32, device=device)
        for layer in range(n_layers):
            x_cond = sample_fn(x_cond, layer, ids)
            local[s0:s0 + len(ids), layer] = x_cond
    gather = ImageGather(n_subjects, (n_views,) + tuple(image_shape), torch.float32, device, as_uint8=as_uint8)
    for k, sid in enumerate(idx):
        views = torch.stack([render_fn(sid, local[k, -1], v) for v in range(n_views)])
        gather.put(k, views)
    samples = gather_shards(local, n_subjects)
    return samples, gather.result(root=images_root)
