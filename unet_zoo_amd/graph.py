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
