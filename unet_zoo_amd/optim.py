"""Step tail on flat buffers: ``clip_grad_norm_`` + ``AdamW`` in three kernel launches.

The reference ends every step with ``torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)`` and
``torch.optim.AdamW.step()`` (unet_zoo/utils/training_loop.py:119-121, optimizer from
scripts/train.py:133).  :class:`FlatClipAdamW` keeps the same arithmetic (decoupled weight decay, bias
corrections, eps outside the square root, the clip coefficient ``min(1, max_norm / (norm + 1e-6))``) but on
ONE flat fp32 buffer per role — parameters, gradients, ``exp_avg``, ``exp_avg_sq`` — so that the gradient
buffer is also what a data-parallel all-reduce moves and what the engine's in-place backward writes.

Parameters are re-pointed at views of the flat parameter buffer (``p.data``) and their ``.grad`` at views
of the flat gradient buffer; only the parameters handed in take part (a parameter the graph never reaches
keeps ``.grad is None`` and is skipped exactly as torch skips it).
"""
from __future__ import annotations

from typing import Iterable, List, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib as L


class FlatClipAdamW:
    ALIGN = 64  # elements (256 bytes)

    def __init__(self, params: Iterable[nn.Parameter], lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999),
                 eps: float = 1e-8, weight_decay: float = 1e-2, max_norm: float = 1.0):
        self.params: List[nn.Parameter] = list(params)
        assert self.params, "no parameters"
        dev = self.params[0].device
        L.require_cuda(*self.params)
        self.lr, self.betas, self.eps, self.weight_decay, self.max_norm = lr, betas, eps, weight_decay, max_norm
        # every tensor starts on a 256-byte boundary (kernels read weights / write gradients with 16-byte
        # accesses); the gaps stay zero in all four buffers, so they neither move nor add to the norm
        A = self.ALIGN
        n = sum((p.numel() + A - 1) // A * A for p in self.params)
        self.n = n
        self.flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.step_count = torch.zeros(1, dtype=torch.float32, device=dev)
        off = 0
        self.spans = []
        for p in self.params:
            assert p.dtype == torch.float32
            k = p.numel()
            self.flat_p[off:off + k].copy_(p.detach().reshape(-1))
            p.data = self.flat_p[off:off + k].view_as(p)
            p.grad = self.flat_g[off:off + k].view_as(p)
            self.spans.append((off, off + (k + A - 1) // A * A))
            off += (k + A - 1) // A * A
        wsb = L.load().uz_clip_adamw_workspace_bytes()
        self._ws = torch.empty(wsb // 4, dtype=torch.float32, device=dev)

    def span_of(self, params: Sequence[nn.Parameter]) -> Tuple[int, int]:
        """[begin, end) of a run of consecutive parameters inside the flat buffers"""
        idx = [self.params.index(p) for p in params]
        assert idx == list(range(idx[0], idx[0] + len(idx))), "parameters are not consecutive in the flat order"
        return self.spans[idx[0]][0], self.spans[idx[-1]][1]

    def zero_grad(self) -> None:
        self.flat_g.zero_()

    @torch.no_grad()
    def step(self) -> None:
        L.check(L.load().uz_clip_adamw(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.exp_avg.data_ptr(),
                                       self.exp_avg_sq.data_ptr(), self.n, self.lr, self.betas[0], self.betas[1],
                                       self.eps, self.weight_decay, self.max_norm, self.step_count.data_ptr(),
                                       self._ws.data_ptr(), L.stream_ptr()), "uz_clip_adamw")

    def last_grad_norm(self) -> torch.Tensor:
        """total gradient norm seen by the last step (before clipping), device scalar"""
        return self._ws[2048 + 3]
