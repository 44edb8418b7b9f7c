"""sample_pdf's uniforms: the CPU generator's stream, continued on the device.

The reference draws them with `torch.rand(...)` on the process-wide CPU generator and uploads them (human_diffusion/NeRF/renderer.py:545):
33.5 M numbers per 512 x 512 view at n_importance = 128 - 50 to 80 ms of one host core against ~40 ms of GPU work for the whole view, so the
drop-in call (u = None) was bound by the host generator.  `rand_like_cpu` hands the generator's mt19937 state to hl_mt19937_uniform
(csrc/hl_mt19937.hip), which writes the SAME numbers straight into device memory, and advances the host generator to the state behind the
last one, so whatever the caller draws next continues exactly where `torch.rand` would have left it.

Layout of torch.get_rng_state() for the CPU generator (ATen/CPUGeneratorImpl.cpp, CPUGeneratorImplState, 5056 bytes): seed u64 @0,
left i32 @8, seeded i32 @12, next u64 @16, state 624 x u64 @24 (32-bit words), the cached normal sample behind it.  Between calls
left + next = 625 (left - 1 words of the current block remain); a freshly seeded generator has left = 1, next = 0: block used up.
"""
import numpy as np
import torch

from .. import _lib

_STATE_BYTES = 5056
_N = 624


def parse_cpu_state(state):
    """ByteTensor of torch.get_rng_state() -> (624 uint32 words, position = words of the current block already drawn)."""
    raw = state.numpy()
    if raw.size != _STATE_BYTES:
        raise RuntimeError(f"unexpected CPU generator state of {raw.size} bytes (this build understands torch's mt19937 state of {_STATE_BYTES})")
    left = int(raw[8:12].view(np.int32)[0])
    nxt = int(raw[16:24].view(np.uint64)[0])
    words = raw[24:24 + _N * 8].view(np.uint64).astype(np.uint32)
    if left == 1:
        pos = _N                       # block used up (also: freshly seeded)
    else:
        if left + nxt != _N + 1:
            raise RuntimeError(f"inconsistent mt19937 state (left {left}, next {nxt})")
        pos = nxt
    return words, pos


def advanced_cpu_state(state, words, pos):
    """The generator state `state` with its mt19937 block / position replaced: what torch.get_rng_state() returns after the draws."""
    raw = state.numpy().copy()
    raw[8:12] = np.array([_N + 1 - pos], dtype=np.int32).view(np.uint8)
    raw[16:24] = np.array([pos], dtype=np.uint64).view(np.uint8)
    raw[24:24 + _N * 8] = words.astype(np.uint64).view(np.uint8)
    return torch.from_numpy(raw)


class PendingDraw:
    """The host half of a device draw: `finish()` waits for the generator state behind the last number (it is ready as soon as the
    generator kernel has run, long before the view is rendered) and installs it in the CPU generator."""

    def __init__(self, state, host_out, event):
        self._state, self._host_out, self._event = state, host_out, event

    def finish(self):
        if self._event is None:
            return
        self._event.synchronize()
        out = self._host_out.numpy()
        torch.set_rng_state(advanced_cpu_state(self._state, out[:_N].copy(), int(out[_N])))
        self._event = None


_SIDE = {}


def rand_like_cpu(shape, device, defer_wait=False):
    """`torch.rand(shape)` of the CPU generator, bit for bit, as a device tensor - drawn by the device on a side stream (it overlaps whatever
    the caller has queued; consumers on the current stream wait for it).  Returns (u, pending): call pending.finish() once the rest of
    the work is enqueued; until then the CPU generator has not moved.  defer_wait=True: the current stream is NOT made to wait here - the
    caller orders its consumer behind pending.u_event itself (hl_render_rays_u_event: only the importance-sampling launch waits)."""
    L = _lib.lib()
    n = int(np.prod(shape))
    state = torch.get_rng_state()
    words, pos = parse_cpu_state(state)
    host_in = torch.from_numpy(words.view(np.int32).copy()).pin_memory()
    host_out = torch.empty(_N + 1, dtype=torch.int32).pin_memory()
    cur = torch.cuda.current_stream(device)
    side = _SIDE.get(device)
    if side is None:
        side = _SIDE[device] = torch.cuda.Stream(device=device)
    with _lib.on(device):
        with torch.cuda.stream(side):
            # (everything the generator touches is allocated on ITS stream: it does not wait for what the caller has queued - the draw for
            #  view k + 1 runs beside the fine pass of view k)
            u = torch.empty(shape, dtype=torch.float32, device=device)
            st_in = host_in.to(device, non_blocking=True)
            st_out = torch.empty(_N + 1, dtype=torch.int32, device=device)
            _lib.check(L.hl_mt19937_uniform(_lib.ptr(st_in, torch.int32), pos, _lib.ptr(u), n, _lib.ptr(st_out, torch.int32), _lib.stream_ptr()), "hl_mt19937_uniform")
            u_ev = torch.cuda.Event()
            u_ev.record(side)
            host_out.copy_(st_out, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
        u.record_stream(cur)                        # (consumed on the caller's stream)
        if not defer_wait:
            cur.wait_event(u_ev)                    # consumers of u on the current stream
    pend = PendingDraw(state, host_out, ev)
    pend.u_event = u_ev
    pend._keep = (host_in, st_in, st_out)
    return u, pend
