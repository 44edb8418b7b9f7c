"""world_size-2 gloo tests (CPU) of the N>1 path: sharding, order-restoring all_gather, layered sampling."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from humanliff_amd import distributed as hd
    r, w, dev = hd.init_distributed("gloo")
    assert (r, w) == (rank, world) and dev.type == "cpu"
    # 1) order-restoring gather with a ragged item count (5 items over 2 ranks)
    idx, valid = hd.shard_indices(5)
    local = torch.tensor([[float(i), float(i) * 10] for i in idx])
    full = hd.gather_shards(local, 5)
    ok1 = torch.equal(full, torch.tensor([[float(i), float(i) * 10] for i in range(5)]))
    # 2) layered sampling: layer k depends on layer k-1 of the same subject only
    calls = []

    def sample_fn(x_cond, layer, ids):
        calls.append((layer, tuple(ids)))
        base = torch.tensor(ids, dtype=torch.float32).view(-1, 1, 1, 1)
        return x_cond * 2 + base + layer       # deterministic stand-in for ddim_sample_loop

    out = hd.sample_layered_sharded(sample_fn, n_subjects=6, n_layers=3, shape=(2, 4, 4), batch=2, device=dev)
    want = torch.empty(6, 3, 2, 4, 4)
    for s in range(6):
        x = torch.zeros(2, 4, 4)
        for k in range(3):
            x = x * 2 + s + k
            want[s, k] = x
    ok2 = torch.equal(out, want)
    per = (6 + world - 1) // world
    ok3 = all(i // per == rank for _, ids in calls for i in ids)       # a subject's layers all run on the rank that owns its block
    # 3) view sharding
    imgs = hd.render_views_sharded(lambda v: torch.full((3, 2, 2), float(v)), 7, (3, 2, 2), dev)
    ok4 = torch.equal(imgs[:, 0, 0, 0], torch.arange(7, dtype=torch.float32))
    # 4) uint8 image gather into a preallocated buffer (truncation like the reference's (clip*255).astype(uint8))
    buf = torch.empty((2 * ((7 + 1) // 2), 2, 2), dtype=torch.uint8)
    idx7, _ = hd.shard_indices(7)
    loc = torch.stack([torch.full((2, 2), v / 10.0 + 0.0039) for v in idx7])
    g8 = hd.gather_shards(loc, 7, as_uint8=True, out=buf)
    ok5 = g8.dtype == torch.uint8 and g8.data_ptr() == buf.data_ptr() and \
        torch.equal(g8[:, 0, 0], torch.tensor([int((v / 10.0 + 0.0039) * 255.0) for v in range(7)], dtype=torch.uint8))
    # 5) the end-to-end flow of scripts/sample_and_render.py (sampling -> per-subject renders -> asynchronous per-subject gathers),
    #    with the HIP network / renderer replaced by arithmetic stand-ins: 5 subjects (ragged), 2 layers, 3 views
    rendered = []

    def render_fn(sid, sample, v):
        rendered.append((sid, v))
        return (sample[:1, :2, :2].permute(1, 2, 0).expand(2, 2, 3) * 0 + (sid * 10 + v) / 100.0).contiguous()

    smp, img = hd.sample_and_render(sample_fn, render_fn, 5, 2, (2, 4, 4), 2, 3, (2, 2, 3), dev, as_uint8=True)
    want_img = torch.tensor([[int(((s * 10 + v) / 100.0) * 255.0) for v in range(3)] for s in range(5)], dtype=torch.uint8)
    ok6 = tuple(smp.shape) == (5, 2, 2, 4, 4) and tuple(img.shape) == (5, 3, 2, 2, 3) and img.dtype == torch.uint8 and \
        torch.equal(img[:, :, 0, 0, 0], want_img) and torch.equal(smp, want[:5, :2])
    per5 = (5 + world - 1) // world
    ok7 = all(min(s, 4) // per5 == rank for s, _ in rendered)          # every rank renders only the subjects it sampled
    smp_f, img_f = hd.sample_and_render(sample_fn, render_fn, 5, 2, (2, 4, 4), 2, 3, (2, 2, 3), dev, as_uint8=False)
    ok8 = img_f.dtype == torch.float32 and torch.allclose(img_f[:, :, 0, 0, 0], torch.tensor([[(s * 10 + v) / 100.0 for v in range(3)] for s in range(5)]))
    # 6) images assembled on rank 0 only (the other ranks skip the transposed copy of the gathered buffer)
    _, img_r = hd.sample_and_render(sample_fn, render_fn, 5, 2, (2, 4, 4), 2, 3, (2, 2, 3), dev, as_uint8=True, images_root=0)
    ok9 = (img_r is None) if rank != 0 else torch.equal(img_r, img)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok1, ok2, ok3, ok4, ok5, ok6, ok7, ok8, ok9))


@pytest.mark.timeout(120)
def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=100) for _ in ps]
    for p in ps:
        p.join(30)
        assert p.exitcode == 0
    for rank, *oks in res:
        assert all(oks), (rank, oks)


def test_single_process_paths():
    from humanliff_amd import distributed as hd
    idx, valid = hd.shard_indices(3, rank=0, world=1)
    assert idx == [0, 1, 2] and all(valid)
    idx, valid = hd.shard_indices(5, rank=1, world=2)
    assert idx == [3, 4, 4] and valid == [True, True, False]          # contiguous blocks, the tail slot repeats the last item
    assert hd.shard_indices(5, rank=0, world=2)[0] == [0, 1, 2]
    g = hd.ImageGather(3, (2,), torch.float32, torch.device("cpu"))
    for k in range(3):
        g.put(k, torch.tensor([k, k + 0.5]))
    assert torch.equal(g.result(), torch.tensor([[0, 0.5], [1, 1.5], [2, 2.5]]))
    t = torch.arange(6.).reshape(3, 2)
    assert torch.equal(hd.gather_shards(t, 3), t)
