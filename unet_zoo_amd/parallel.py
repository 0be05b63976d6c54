"""Data-parallel training over RCCL/xGMI: one process per GPU, gradients averaged with bucketed
all-reduces that start while the backward tape is still running.

Replaces the reference's single-process ``nn.DataParallel`` (unet_zoo/utils/multi_gpu.py:28-31:
scatter / replicate / gather every step, gradients reduced to GPU 0) with the per-GPU-process
scheme SURVEY.md §8e describes.  Semantics kept from the reference: per-shard BatchNorm statistics
(no SyncBN), buffers are NOT synchronised per step, rank 0 writes checkpoints
(``state_dict`` of the unwrapped module, multi_gpu.py:39-42).

The engine's backward hands each parameter gradient to :meth:`BucketReducer.push` the moment its
kernels are enqueued; a bucket whose members are all present is all-reduced on a side stream
(ordered after the compute stream by an event), so communication hides behind the remaining
backward kernels.  ``BucketReducer`` knows nothing about models and works on CPU tensors with the
``gloo`` backend too, which is how tests cover the world_size > 1 path without GPUs.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist
import torch.nn as nn


class BucketReducer:
    def __init__(self, params: Sequence[nn.Parameter], process_group=None, bucket_bytes: int = 25 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.bucket_bytes = bucket_bytes
        self._order: List[nn.Parameter] = []          # production order seen in the first backward
        self._planned = False
        self._buckets: List[Dict] = []
        self._where: Dict[nn.Parameter, tuple] = {}   # param -> (bucket index, offset)
        self._pending: List = []
        self._first_grads: Dict[nn.Parameter, torch.Tensor] = {}
        # a parameter may receive several contributions in one backward (the engine calls push() with the running
        # sum each time): the planning backward counts them, later backwards launch a bucket only when every
        # member has received its LAST one
        self._expect: Dict[nn.Parameter, int] = {}
        self._got: Dict[nn.Parameter, int] = {}
        dev = self.params[0].device if self.params else torch.device("cpu")
        self._cuda = dev.type == "cuda"
        self._comm_stream = torch.cuda.Stream(device=dev) if self._cuda else None

    # -- planning --------------------------------------------------------------------------
    def _plan(self) -> None:
        """Group parameters into ~bucket_bytes buckets in the order backward produces them."""
        cur: List[nn.Parameter] = []
        size = 0
        groups: List[List[nn.Parameter]] = []
        for p in self._order:
            cur.append(p)
            size += p.numel() * 4
            if size >= self.bucket_bytes:
                groups.append(cur)
                cur, size = [], 0
        if cur:
            groups.append(cur)
        for bi, g in enumerate(groups):
            n = sum(p.numel() for p in g)
            flat = torch.zeros(n, dtype=torch.float32, device=g[0].device)
            off = 0
            for p in g:
                self._where[p] = (bi, off)
                off += p.numel()
            need = sum(self._expect[p] for p in g)
            self._buckets.append({"flat": flat, "members": g, "need": need, "left": need})
        self._planned = True

    # -- per-backward API ------------------------------------------------------------------
    def push(self, p: nn.Parameter, g: torch.Tensor) -> None:
        """A parameter's gradient has been enqueued on the current stream."""
        if self.world == 1:
            return
        if not self._planned:
            if p not in self._first_grads:
                self._order.append(p)
            self._first_grads[p] = g
            self._expect[p] = self._expect.get(p, 0) + 1
            return
        if p not in self._where:
            raise RuntimeError("a parameter received a gradient that did not in the planning backward: the set of "
                               "parameters receiving gradients changed between iterations")
        bi, off = self._where[p]
        b = self._buckets[bi]
        got = self._got.get(p, 0) + 1
        if got > self._expect[p]:
            raise RuntimeError("a parameter received more gradient contributions than in the planning backward")
        self._got[p] = got
        # every contribution is the running sum: the last copy holds the final gradient
        b["flat"][off:off + p.numel()].copy_(g.reshape(-1))
        b["left"] -= 1
        if b["left"] == 0:
            self._launch(b)

    def _launch(self, b: Dict) -> None:
        flat = b["flat"]
        if self._cuda:
            ev = torch.cuda.Event()
            ev.record()                      # after the copies on the compute stream
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(ev)
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                self._pending.append((work, flat))
        else:
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            self._pending.append((work, flat))

    def finish(self, grads: Dict[nn.Parameter, torch.Tensor]) -> None:
        """Backward is fully enqueued: wait for the reductions and replace each entry of `grads`
        by its cross-rank average (views into the bucket buffers)."""
        if self.world == 1:
            return
        if not self._planned:
            # first backward: no plan yet -> reduce everything now, then plan for the next one
            self._plan()
            for p, g in self._first_grads.items():
                bi, off = self._where[p]
                self._buckets[bi]["flat"][off:off + p.numel()].copy_(g.reshape(-1))
            self._first_grads.clear()
            for b in self._buckets:
                self._launch(b)
        else:
            for b in self._buckets:
                if b["left"] != 0:
                    raise RuntimeError("a gradient bucket is incomplete: the set of parameters "
                                       "receiving gradients changed between iterations")
        inv = 1.0 / self.world
        for work, flat in self._pending:
            if self._cuda:
                with torch.cuda.stream(self._comm_stream):
                    work.wait()
                    flat.mul_(inv)
            else:
                work.wait()
                flat.mul_(inv)
        self._pending.clear()
        if self._cuda:
            torch.cuda.current_stream().wait_stream(self._comm_stream)
        # hand out views of a per-bucket copy: the bucket buffers are overwritten by the next
        # backward, while .grad may live on (gradient accumulation)
        self._got.clear()
        outs: List[Optional[torch.Tensor]] = [None] * len(self._buckets)
        for b in self._buckets:
            b["left"] = b["need"]
        for p in list(grads.keys()):
            if p not in self._where:
                continue
            bi, off = self._where[p]
            if grads[p] is None:
                # in-place mode (HipModule.grads_in_place): the local gradient already sits in p.grad and autograd
                # gets None for it, so the average goes back into p.grad -- handing it out would ADD it on top
                if p.grad is not None:
                    p.grad.copy_(self._buckets[bi]["flat"][off:off + p.numel()].view_as(p))
                continue
            if outs[bi] is None:
                outs[bi] = self._buckets[bi]["flat"].clone()
            grads[p] = outs[bi][off:off + p.numel()].view_as(p)


class RcclDataParallel(nn.Module):
    """Wrap a HipModule for one-process-per-GPU data parallelism (``module`` attribute and
    ``state_dict`` prefix behave like DistributedDataParallel's, which the reference's checkpoint
    code already unwraps: multi_gpu.py:13-18, 44-53)."""

    def __init__(self, module: nn.Module, process_group=None, bucket_mb: float = 25.0,
                 broadcast_from_rank0: bool = True):
        """Parameters and buffers are broadcast from rank 0 ONCE here.  BatchNorm running statistics are then
        updated from each rank's own shard and drift apart, exactly as the replicas of the reference's
        nn.DataParallel do (only replica 0's statistics survive there): checkpoints are rank 0's to write."""
        super().__init__()
        self.module = module
        self.reducer = BucketReducer(list(module.parameters()), process_group, int(bucket_mb * (1 << 20)))
        if dist.is_initialized() and broadcast_from_rank0 and self.reducer.world > 1:
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, src=0, group=process_group)
        module._grad_sink = self.reducer.push
        module._grad_sink_done = self.reducer.finish

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)
