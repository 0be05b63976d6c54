"""CPU, world_size 2 over gloo: the bucketed gradient reducer behind RcclDataParallel
(unet_zoo_amd/parallel.py) — the data-parallel path that replaces the reference's
nn.DataParallel (unet_zoo/utils/multi_gpu.py:28-31).  On the GPU box the same code runs over
RCCL; here the reducer is fed (param, grad) pairs the way the engine's backward feeds it."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from unet_zoo_amd.parallel import BucketReducer, RcclDataParallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_params(seed=0):
    g = torch.Generator().manual_seed(seed)
    shapes = [(64, 3, 3, 3), (64,), (64, 64, 3, 3), (64,), (128, 64, 3, 3), (1, 64, 1, 1), (1,)]
    return [nn.Parameter(torch.randn(s, generator=g)) for s in shapes]


def _grads_for(rank, step, params):
    g = torch.Generator().manual_seed(1000 * step + rank)
    return [torch.randn(p.shape, generator=g) for p in params]


def _worker(rank, world, port, bucket_bytes, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        params = _make_params()
        red = BucketReducer(params, bucket_bytes=bucket_bytes)
        ok = True
        for step in range(3):  # step 0 plans the buckets, steps 1.. use the overlapped path
            grads = _grads_for(rank, step, params)
            out = {}
            order = list(reversed(range(len(params))))  # backward produces the last layer first
            for i in order:
                out[params[i]] = grads[i]
                red.push(params[i], grads[i])
            red.finish(out)
            for i, p in enumerate(params):
                expect = sum(_grads_for(r, step, params)[i] for r in range(world)) / world
                ok = ok and torch.allclose(out[p], expect, atol=1e-6)
                ok = ok and out[p].shape == p.shape
        q.put((rank, ok, len(red._buckets)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bucket_bytes", [1 << 10, 150_000, 25 << 20])
def test_bucket_reducer_averages_across_two_ranks(bucket_bytes):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bucket_bytes, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    nb = res[0][2]
    assert nb >= 1 and (bucket_bytes > 1 << 20) == (nb == 1)


class _Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.w = nn.Parameter(torch.ones(4))
        self.register_buffer("stat", torch.zeros(2))
        self._grad_sink = None
        self._grad_sink_done = None


def _wrap_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = _Toy()
        with torch.no_grad():
            m.w.fill_(float(rank + 1))
            m.stat.fill_(float(rank + 5))
        w = RcclDataParallel(m)
        # parameters AND buffers start from rank 0's values, hooks are installed, and the
        # state_dict of the wrapper carries the 'module.' prefix the reference's loader strips
        q.put((rank, m.w.tolist(), m.stat.tolist(), m._grad_sink is not None,
               sorted(w.state_dict().keys())))
    finally:
        dist.destroy_process_group()


def test_wrapper_broadcasts_rank0_state_and_installs_hooks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_wrap_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, w, stat, hooked, keys in res:
        assert w == [1.0] * 4 and stat == [5.0] * 2 and hooked
        assert keys == ["module.stat", "module.w"]


def test_single_process_reducer_is_a_no_op():
    params = _make_params()
    red = BucketReducer(params)
    g = {p: torch.ones_like(p) for p in params}
    for p in params:
        red.push(p, g[p])
    red.finish(g)
    assert all(torch.equal(g[p], torch.ones_like(p)) for p in params)


def _double_push_worker(rank, world, port, q):
    """A parameter that receives TWO contributions per backward (the engine pushes the running sum each time)
    and the in-place mode (grads[p] is None, the gradient lives in p.grad): ADVICE round 1."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        params = _make_params()
        red = BucketReducer(params, bucket_bytes=1 << 10)      # several buckets
        ok = True
        shared = params[2]                                      # pushed twice per backward
        for step in range(3):
            grads = _grads_for(rank, step, params)
            out = {}
            for i in reversed(range(len(params))):
                if params[i] is shared:
                    half = grads[i] * 0.25
                    red.push(shared, half)                      # first contribution: partial value
                    # another parameter of the same bucket arrives in between
                    continue
                out[params[i]] = grads[i]
                red.push(params[i], grads[i])
            out[shared] = grads[2]
            red.push(shared, grads[2])                          # running sum = the final gradient
            red.finish(out)
            for i, p in enumerate(params):
                expect = sum(_grads_for(r, step, params)[i] for r in range(world)) / world
                ok = ok and torch.allclose(out[p], expect, atol=1e-6)
        # in-place mode: gradients sit in p.grad, the dict holds None
        for step in range(3, 5):
            grads = _grads_for(rank, step, params)
            out = {}
            for i in reversed(range(len(params))):
                params[i].grad = grads[i].clone()
                out[params[i]] = None
                if params[i] is shared:
                    red.push(shared, params[i].grad * 0.5)
                red.push(params[i], params[i].grad)
            red.finish(out)
            for i, p in enumerate(params):
                expect = sum(_grads_for(r, step, params)[i] for r in range(world)) / world
                ok = ok and out[p] is None and torch.allclose(p.grad, expect, atol=1e-6)
        # one contribution too many is an error, not a silent early launch
        raised = False
        try:
            for _ in range(3):
                red.push(shared, torch.zeros_like(shared))
        except RuntimeError:
            raised = True
        q.put((rank, ok, raised))
    finally:
        dist.destroy_process_group()


def test_reducer_counts_contributions_and_supports_in_place_grads():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_double_push_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok and raised for _, ok, raised in res), res
