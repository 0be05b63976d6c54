"""Whole-network autograd node.

A :class:`HipModule` keeps the reference's ``nn.Module`` contract (parameters, buffers,
``state_dict`` key names, ``train()/eval()``, ``.to(device)``) but its ``forward``
does not call a single torch op on the hot path: it hands the input and the parameters to ONE
``torch.autograd.Function`` whose forward runs the model's ``emit`` program on the HIP engine and
whose backward replays the engine's tape.  PyTorch sees one node, so the loss
(``BCEWithLogitsLoss``, scripts/train.py:135), ``clip_grad_norm_`` and ``AdamW``
(training_loop.py:119-121) work unchanged on top.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib as L
from .engine import Engine, PackCache

_DTYPES = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp32": torch.float32,
           "float32": torch.float32}
_default_dtype = _DTYPES[os.environ.get("UNET_ZOO_AMD_DTYPE", "bf16").lower()]


def set_default_dtype(dtype) -> None:
    """Run dtype of models created afterwards: 'bf16' (throughput) or 'fp32' (exact parity)."""
    global _default_dtype
    _default_dtype = _DTYPES[dtype.lower()] if isinstance(dtype, str) else dtype
    L.dtype_code(_default_dtype)


def get_default_dtype() -> torch.dtype:
    return _default_dtype


class _GraphFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model: "HipModule", plist: List[nn.Parameter], record: bool, x: torch.Tensor,
                *params: torch.Tensor):
        model._pack_cache.refresh(model.run_dtype)
        eng = Engine(model.run_dtype, x.device, model.training, record, model._grad_sink,
                     model._pack_cache, model.grads_in_place)
        outs = model.emit(eng, x)
        eng.finish_forward()
        ctx.eng = eng if record else None
        ctx.plist = plist
        ctx.sink_done = model._grad_sink_done
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grad_outputs):
        eng = ctx.eng
        if eng is None:
            raise RuntimeError("backward through a HipModule forward that ran without gradient recording")
        grads = eng.backward(grad_outputs)
        ctx.eng = None
        if ctx.sink_done is not None:
            ctx.sink_done(grads)
        out = []
        for i, p in enumerate(ctx.plist):
            out.append(grads.get(p) if ctx.needs_input_grad[4 + i] else None)
        # drop every other reference so autograd can adopt the tensors as .grad without copying
        grads.clear()
        return (None, None, None, None, *out)


class HipModule(nn.Module):
    """Base class of the HIP-backed model graphs."""

    def __init__(self):
        super().__init__()
        self.run_dtype: torch.dtype = get_default_dtype()
        # set by parallel.RcclDataParallel: called with (param, grad) as soon as a gradient's
        # kernels are enqueued, and once with the whole dict when backward has been enqueued
        self._grad_sink: Optional[Callable] = None
        self._grad_sink_done: Optional[Callable] = None
        self._pack_cache = PackCache()
        # True: backward OVERWRITES existing p.grad tensors in place (no accumulation, no temporaries;
        # conv biases in front of a train-mode BatchNorm are left untouched = zero).  For pre-allocated
        # flat gradient buffers and hipGraph capture; default False = ordinary autograd semantics.
        self.grads_in_place = False

    def _apply(self, fn, *args, **kwargs):
        # .to()/.cuda()/.float() replace parameter storage: packed copies and their pointer table die
        self._pack_cache.invalidate()
        return super()._apply(fn, *args, **kwargs)

    # -- to be provided by the model --------------------------------------------------------
    def emit(self, eng: Engine, x: torch.Tensor) -> Sequence[torch.Tensor]:
        raise NotImplementedError

    def wrap_outputs(self, outs: Tuple[torch.Tensor, ...]):
        return outs[0]

    # ---------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor):
        L.load()  # fail loudly when the kernel library is missing
        L.require_cuda(x)
        if x.requires_grad:
            raise NotImplementedError("gradients with respect to the input image are not produced")
        plist = list(self.parameters())
        record = torch.is_grad_enabled() and any(p.requires_grad for p in plist)
        outs = _GraphFn.apply(self, plist, record, x, *plist)
        return self.wrap_outputs(outs)


class PhasedStep:
    """Forward, loss and backward of a :class:`HipModule` driven directly (no autograd node), with the
    backward cut into phases at tape positions.

    Purpose: data-parallel training replayed from hipGraphs.  A collective cannot sit
    inside a captured graph, and one all-reduce after the whole backward is fully exposed (124 MB for
    UNet: ~2.6 ms over the single xGMI link of a 2-GPU job).  With phases, phase k is its own graph;
    the gradients it finished are all-reduced (eager RCCL, asynchronous) while the graph of phase k+1
    runs, so only the last — smallest — phase's exchange is exposed.

    Gradients are written in place into the parameters' existing ``.grad`` tensors
    (``model.grads_in_place`` semantics): allocate them (e.g. as views of one flat buffer ordered by
    :meth:`plan`) before calling :meth:`forward`.
    """

    def __init__(self, model: HipModule, loss_fn: Callable):
        self.model = model
        self.loss_fn = loss_fn
        self.eng: Optional[Engine] = None
        self._gouts: Optional[Tuple[Optional[torch.Tensor], ...]] = None

    def forward(self, x: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        self.emit(x)
        return self.loss(target)

    def emit(self, x: torch.Tensor):
        """the model's forward on a fresh engine; returns (and keeps as `outputs`) the wrapped outputs"""
        m = self.model
        L.load()
        L.require_cuda(x)
        m._pack_cache.refresh(m.run_dtype)
        self.eng = Engine(m.run_dtype, x.device, m.training, True, None, m._pack_cache, True)
        with torch.no_grad():
            self._outs = tuple(m.emit(self.eng, x))
            self.eng.finish_forward()
        self.outputs = m.wrap_outputs(tuple(o.detach() for o in self._outs))
        return self.outputs

    def loss(self, target: torch.Tensor) -> torch.Tensor:
        """loss_fn on the outputs of the last emit(); leaves d(loss)/d(output) for backward(..., heads=True)"""
        direct = getattr(self.loss_fn, "direct", None)
        if direct is not None:   # a criterion that hands over its own gradients (the fused BCE + Dice kernel): no autograd node
            loss, gouts = direct(self.model.wrap_outputs(tuple(o.detach() for o in self._outs)), target)
            assert len(gouts) == len(self._outs)
            self._gouts = tuple(gouts)
            return loss.detach()
        leaves = tuple(o.detach().requires_grad_(True) for o in self._outs)
        with torch.enable_grad():
            loss = self.loss_fn(self.model.wrap_outputs(leaves), target)
            self._gouts = torch.autograd.grad(loss, leaves, allow_unused=True)
        return loss.detach()

    def set_output_grads(self, gouts: Sequence[Optional[torch.Tensor]]) -> None:
        """use these tensors as d(loss)/d(output) (static buffers of a captured backward)"""
        self._gouts = tuple(gouts)

    @property
    def n_entries(self) -> int:
        return len(self.eng.tape)

    def backward(self, hi: int, lo: int, heads: bool) -> None:
        """heads (optional), then tape entries hi-1 ... lo"""
        self.eng.backward_range(self._gouts if heads else None, hi, lo)

    def finish(self) -> None:
        self.eng.tape.clear()
        self.eng = None
        self._gouts = None
        self._outs = None

    def plan(self, fractions: Sequence[float] = (0.45, 0.85)):
        """After one complete forward + backward(n_entries, 0, True): cut the backward where the
        cumulative gradient bytes cross `fractions` of the total.  Returns (cuts, groups):
        cuts = [n_entries, c1, ..., 0] (phase k = entries cuts[k]-1 ... cuts[k+1], phase 0 also runs
        the heads) and groups[k] = the parameters whose gradient is complete when phase k ends."""
        log = self.eng.grad_log
        n = self.n_entries
        last: dict = {}
        for pos, p in log:                       # a parameter is complete at its LAST (= lowest) position
            last[p] = pos if p not in last else min(last[p], pos)
        order = sorted(last.items(), key=lambda kv: -kv[1])
        total = sum(p.numel() for p, _ in order)
        cuts, groups, cur, acc, fi = [n], [], [], 0, 0
        fr = list(fractions)
        for idx, (p, pos) in enumerate(order):
            cur.append(p)
            acc += p.numel()
            nxt = order[idx + 1][1] if idx + 1 < len(order) else None
            if fi < len(fr) and acc >= fr[fi] * total and nxt is not None and nxt < pos:
                cuts.append(pos)                  # phase ends after entry `pos` has run
                groups.append(cur)
                cur = []
                while fi < len(fr) and acc >= fr[fi] * total:
                    fi += 1
        cuts.append(0)
        # allocated but not produced in this pass (e.g. analytically-zero conv biases): ride with the last
        # phase; parameters without a .grad tensor (never reached by the graph) stay out
        cur += [p for p in self.model.parameters() if p not in last and p.grad is not None]
        groups.append(cur)
        return cuts, groups
