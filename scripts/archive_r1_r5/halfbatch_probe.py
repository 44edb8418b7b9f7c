"""Developer probe (round 5): does running a batch as CONCURRENT sub-batches (independent forwards on their own streams, each with its
own two-tower stream pair) fill the chip better than one forward of the whole batch?  Two model objects share the parameters' values
(each packs its own copy: the handles hold per-forward events)."""
import sys, os, time, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench

dev = torch.device("cuda:0")
model, diffusion, sd = bench.build_unet(dev)
NM = int(os.environ.get("NMODELS", "4"))
models = [model] + [copy.deepcopy(model) for _ in range(NM - 1)]
for m in models[1:]:
    m._hip = None; m._ws = {}; m._sd_cache = None
streams = [torch.cuda.Stream() for _ in range(NM)]


def inputs(B, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((B, 27, 256, 256), generator=g).to(dev)
    return x, torch.zeros_like(x), torch.full((B,), 500, dtype=torch.int64, device=dev), torch.zeros((B,), dtype=torch.int64, device=dev)


def run(parts, iters=8, warm=2, skew_ms=0.0):
    """parts: list of sub-batch sizes, one stream + model each. Returns ms per (whole) forward."""
    ins = [inputs(b, 10 + i) for i, b in enumerate(parts)]
    def go(n):
        for _ in range(n):
            for i, (x, xc, t, y) in enumerate(ins):
                with torch.cuda.stream(streams[i]):
                    models[i](x, t, xc, y=y)
    with torch.no_grad():
        go(warm)
        torch.cuda.synchronize()
        if skew_ms > 0:
            for i in range(1, len(parts)):
                with torch.cuda.stream(streams[i]):
                    torch.cuda._sleep(int(skew_ms * i * 2.0e6))   # ~2 GHz counter
        t0 = time.perf_counter()
        go(iters)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    tot = sum(parts)
    ms = 1e3 * (t2 - t0) / iters
    print(f"parts {parts} skew {skew_ms}: {ms:7.3f} ms per forward of {tot} = {tot / ms * 1e3:7.2f} fwd-samples/s  (enqueue {1e3 * (t1 - t0) / iters:.2f} ms)", flush=True)
    return ms


for parts in ([1], [2], [4], [8], [2, 2], [1, 1], [1, 1, 1, 1], [4, 4], [2, 2, 2, 2], [3, 1]):
    if len(parts) > NM:
        continue
    run(parts)
for parts, sk in (([2, 2], 4.0), ([2, 2], 8.0), ([4, 4], 8.0), ([4, 4], 16.0), ([1, 1], 3.0), ([1, 1], 6.0)):
    run(parts, skew_ms=sk)
