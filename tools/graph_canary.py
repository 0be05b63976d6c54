"""Root-cause tool for "a torch reduction reads 0 inside a replayed hipGraph" (round-1 swin_bench1/3 logs).

Three experiments, each prints one line per finding:

  A. torch-only graphs (no kernel of this library): BCE-with-logits mean over 1M logits, `.sum(0)` over a
     (100352, 96) tensor, each captured alone and replayed many times -> does the stack itself misbehave?
  B. guard-malloc eager step: every tensor the engine / ops layer allocates gets 64 KB canary zones on both
     sides and is never freed; after one eager train step every zone is checked -> an out-of-bounds WRITE
     by any kernel of this library shows up with its allocation site.
  C. the same under hipGraph capture + replays, with the torch loss inside the graph, comparing the
     replayed loss with the eager one.

    python tools/graph_canary.py [model] [size] [batch]
"""
import sys
import traceback

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
import unet_zoo_amd  # noqa: E402
from unet_zoo_amd import engine as engine_mod, ops as ops_mod  # noqa: E402

GUARD = 64 << 10
CANARY = 0xA5
_allocs = []          # (base uint8 tensor, payload bytes, site)
_real = torch


class _GuardTorch:
    """stands in for the `torch` global of ops.py / engine.py: allocation functions return a payload view
    between two canary zones"""

    def __getattr__(self, name):
        return getattr(_real, name)

    @staticmethod
    def _alloc(shape, dtype, device, zero):
        if isinstance(shape, int):
            shape = (shape,)
        shape = tuple(int(s) for s in shape)
        n = 1
        for s in shape:
            n *= s
        es = _real.empty(0, dtype=dtype).element_size()
        nbytes = (n * es + 255) // 256 * 256
        base = _real.empty(nbytes + 2 * GUARD, dtype=_real.uint8, device=device)
        base[:GUARD].fill_(CANARY)
        base[GUARD + n * es:].fill_(CANARY)       # also covers the round-up slack right behind the payload
        site = "".join(traceback.format_stack(limit=5)[:-2])
        _allocs.append((base, n * es, site))
        t = base[GUARD:GUARD + n * es].view(dtype).view(shape)
        if zero:
            t.zero_()
        return t

    def empty(self, *shape, dtype=None, device=None, **kw):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, _real.Size)):
            shape = tuple(shape[0])
        if device is None or _real.device(device).type != "cuda":
            return _real.empty(*shape, dtype=dtype, device=device, **kw)
        return self._alloc(shape, dtype or _real.float32, device, False)

    def zeros(self, *shape, dtype=None, device=None, **kw):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, _real.Size)):
            shape = tuple(shape[0])
        if device is None or _real.device(device).type != "cuda":
            return _real.zeros(*shape, dtype=dtype, device=device, **kw)
        return self._alloc(shape, dtype or _real.float32, device, True)

    def empty_like(self, t, **kw):
        if not t.is_cuda:
            return _real.empty_like(t, **kw)
        return self._alloc(tuple(t.shape), kw.get("dtype", t.dtype), t.device, False)

    def zeros_like(self, t, **kw):
        if not t.is_cuda:
            return _real.zeros_like(t, **kw)
        return self._alloc(tuple(t.shape), kw.get("dtype", t.dtype), t.device, True)


def guards_on():
    g = _GuardTorch()
    ops_mod.torch = g
    engine_mod.torch = g


def guards_off():
    ops_mod.torch = _real
    engine_mod.torch = _real


def check_guards(tag):
    bad = 0
    for base, nb, site in _allocs:
        lo = base[:GUARD]
        hi = base[GUARD + nb:]
        for name, z in (("below", lo), ("above", hi)):
            ne = (z != CANARY).nonzero()
            if ne.numel():
                bad += 1
                first, last = int(ne[0]), int(ne[-1])
                off = first - GUARD if name == "below" else first
                print(f"[{tag}] GUARD VIOLATION {name} payload of {nb} bytes: {ne.numel()} bytes changed, "
                      f"first at {off:+d} (zone offsets {first}..{last})\n{site}", flush=True)
    print(f"[{tag}] {len(_allocs)} guarded allocations, {bad} violated", flush=True)
    return bad


def experiment_a():
    dev = "cuda"
    logits = torch.randn(16, 1, 256, 256, device=dev)
    mask = (torch.rand(16, 1, 256, 256, device=dev) > 0.5).float()
    rows = torch.randn(100352, 96, device=dev)
    want_loss = F.binary_cross_entropy_with_logits(logits, mask).item()
    want_sum = rows.sum(0)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        F.binary_cross_entropy_with_logits(logits, mask)
        rows.sum(0)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    for mode in ("global", "thread_local"):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode=mode):
            l = F.binary_cross_entropy_with_logits(logits, mask)
            cs = rows.sum(0)
        bad_l = bad_s = 0
        for i in range(300):
            g.replay()
            if i % 10 == 0:
                torch.cuda.synchronize()
            torch.cuda.synchronize()
            if abs(l.item() - want_loss) > 1e-5:
                bad_l += 1
            if (cs - want_sum).abs().max().item() > 1e-2:
                bad_s += 1
        print(f"[A {mode}] torch-only graph, 300 replays: loss wrong {bad_l}x, sum(0) wrong {bad_s}x "
              f"(loss {l.item():.6f} vs {want_loss:.6f})", flush=True)


def experiment_a2():
    """what exactly goes wrong in A: per-replay ratio of the captured sum(0) to the eager one, and a raw
    hipMemsetAsync node (the reduction's semaphore reset) followed by an increment"""
    import ctypes
    dev = "cuda"
    hip = ctypes.CDLL("libamdhip64.so")
    for shape in ((100352, 96), (4096, 96), (1 << 20, 4), (100352, 768)):
        rows = torch.randn(*shape, device=dev)
        want = rows.sum(0)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            rows.sum(0)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            cs = rows.sum(0)
        hist = []
        for i in range(6):
            g.replay()
            torch.cuda.synchronize()
            d = (cs - want).abs().max().item()
            hist.append((round(d, 4), round((cs[0] / want[0]).item(), 4), int((cs == 0).sum())))
        print(f"[A2 sum(0) {shape}] per replay (max abs diff, cs[0]/want[0], zeros): {hist}", flush=True)
    buf = torch.zeros(64, device=dev, dtype=torch.int32)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        buf.add_(1)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = hip.hipMemsetAsync(ctypes.c_void_p(buf.data_ptr()), 0, ctypes.c_size_t(buf.numel() * 4), st)
        buf.add_(1)
    vals = []
    for i in range(5):
        g.replay()
        torch.cuda.synchronize()
        vals.append(int(buf[0]))
    print(f"[A2 memset node + add_(1)] rc {rc}; buf[0] after each replay (1 every time if the memset node runs): {vals}", flush=True)


def torch_loss(out, mask):
    if isinstance(out, dict):
        return sum(F.binary_cross_entropy_with_logits(v, mask) for v in out.values())
    return F.binary_cross_entropy_with_logits(out, mask)


def experiment_bc(name, size, batch):
    from bench import make_model
    torch.manual_seed(0)
    if name == "swin_unet_v2":      # no stochastic depth here: the loss must be the same number every step
        m = unet_zoo_amd.create_model(name, image_size=size, window_size=8 if (size // 4) % 8 == 0 else 7, drop_path_rate=0.0)
    else:
        m, _ = make_model(name, size)
    m = m.cuda().train()
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(batch, 3, size, size, generator=gen).cuda()
    mask = (torch.rand(batch, 1, size, size, generator=gen) > 0.5).float().cuda()
    params = list(m.parameters())
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for p in params:
            p.grad = None
        loss = torch_loss(m(x), mask)       # un-guarded warm-up (pack cache, first-use allocations)
        loss.backward()
        eager_loss = loss.item()
        guards_on()
        for p in params:
            p.grad = None
        loss = torch_loss(m(x), mask)
        loss.backward()
        torch.cuda.synchronize()
        print(f"[B] eager loss {loss.item():.6f} (un-guarded {eager_loss:.6f})", flush=True)
        check_guards("B eager")
        guards_off()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    _allocs.clear()
    used = [p for p in params if p.grad is not None]
    eager_g = torch.cat([p.grad.reshape(-1) for p in used]).clone()
    for p in params:
        p.grad = None
    flat = torch.zeros(sum(p.numel() for p in used), device="cuda")
    off = 0
    for p in used:
        p.grad = flat[off:off + p.numel()].view_as(p)
        off += p.numel()
    m.grads_in_place = True
    for guarded in (False, True):
        if guarded:
            guards_on()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, capture_error_mode="thread_local"):
            out = m(x)
            sl = torch_loss(out, mask)
            sl.backward()
        guards_off()
        wrong = 0
        vals = []
        for i in range(20):
            gr.replay()
            torch.cuda.synchronize()
            v = sl.item()
            vals.append(round(v, 5))
            if abs(v - eager_loss) > 2e-3 * max(1.0, abs(eager_loss)):
                wrong += 1
        with torch.no_grad():
            re = torch_loss(out, mask).item()
        gd = (flat - eager_g).abs().max().item() / eager_g.abs().max().item()
        print(f"[C guarded={guarded}] 20 replays: loss wrong {wrong}x; values {sorted(set(vals))}; loss recomputed "
              f"eagerly from the replay's logits {re:.6f}; eager {eager_loss:.6f}; max grad diff vs eager {gd:.3e}", flush=True)
        if guarded:
            check_guards("C graph")
        del gr


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "swin_unet_v2"
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    experiment_a()
    experiment_a2()
    experiment_bc(name, size, batch)
