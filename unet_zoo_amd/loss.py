"""Loss and Dice metric of the reference's training step on the device.

The reference computes ``criterion(outputs, mask)`` with ``nn.BCEWithLogitsLoss()`` (scripts/train.py:135) and
``dice_coefficient(main_pred_logits, mask)`` (utils/metrics.py:7-24) and reads both back with ``.item()`` every step
(utils/training_loop.py:113-124).  Here one kernel pass produces the loss, its gradient and the Dice value as device
scalars; nothing synchronises with the host until the caller asks for the numbers.
"""
from __future__ import annotations

from typing import Tuple

import torch

from . import _lib as L


class _BceDice(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits: torch.Tensor, target: torch.Tensor):
        L.require_cuda(logits, target)
        assert logits.shape == target.shape, (logits.shape, target.shape)
        x = logits.detach().contiguous().float()
        t = target.detach().contiguous().float()
        n = x.numel()
        lib = L.load()
        need_grad = ctx.needs_input_grad[0]
        dlogits = torch.empty_like(x) if need_grad else None
        out = torch.empty(2, dtype=torch.float32, device=x.device)
        ws = torch.empty(L.check_count(lib.uz_bce_dice_workspace_bytes(n), "uz_bce_dice_workspace_bytes") // 8,
                         dtype=torch.float64, device=x.device)
        L.check(lib.uz_bce_dice(x.data_ptr(), t.data_ptr(), n, dlogits.data_ptr() if need_grad else None, out.data_ptr(),
                                ws.data_ptr(), L.stream_ptr()), "uz_bce_dice")
        ctx.dlogits = dlogits
        ctx.in_dtype = logits.dtype
        dice = out[1]
        ctx.mark_non_differentiable(dice)
        return out[0], dice

    @staticmethod
    def backward(ctx, g_loss, _g_dice):
        return (ctx.dlogits * g_loss).to(ctx.in_dtype), None


def bce_dice_direct(logits: torch.Tensor, target: torch.Tensor):
    """(loss, dice, d(loss)/d(logits)) from ONE kernel pass and nothing else: no autograd node, hence none of the launches
    loss.backward() adds (the ones_like fill, the multiplication by it, a cast) -- the graphed step's form"""
    L.require_cuda(logits, target)
    assert logits.shape == target.shape, (logits.shape, target.shape)
    x = logits.detach().contiguous().float()
    t = target.detach().contiguous().float()
    n = x.numel()
    lib = L.load()
    dlogits = torch.empty_like(x)
    out = torch.empty(2, dtype=torch.float32, device=x.device)
    ws = torch.empty(L.check_count(lib.uz_bce_dice_workspace_bytes(n), "uz_bce_dice_workspace_bytes") // 8,
                     dtype=torch.float64, device=x.device)
    L.check(lib.uz_bce_dice(x.data_ptr(), t.data_ptr(), n, dlogits.data_ptr(), out.data_ptr(), ws.data_ptr(),
                            L.stream_ptr()), "uz_bce_dice")
    return out[0], out[1], dlogits


def loss_and_dice_direct(outputs, target: torch.Tensor):
    """loss_and_dice() plus the gradients of the loss with respect to every output tensor, in the order the model emits
    them (HipModule.wrap_outputs keeps that order for dicts and lists); the summed losses of u2net's seven maps /
    nested_unet's deep-supervision heads have unit weights (training_loop.py:24-32, 60-64), so each map's gradient is its
    own BCE gradient"""
    if isinstance(outputs, dict):
        trip = [bce_dice_direct(v, target) for v in outputs.values()]
        return sum(p[0] for p in trip), trip[0][1], tuple(p[2] for p in trip)
    if isinstance(outputs, (list, tuple)):
        trip = [bce_dice_direct(v, target) for v in outputs]
        return sum(p[0] for p in trip), trip[-1][1], tuple(p[2] for p in trip)
    l, d, g = bce_dice_direct(outputs, target)
    return l, d, (g,)


def bce_dice_with_logits(logits: torch.Tensor, target: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(BCEWithLogitsLoss(logits, target), dice_coefficient(logits, target)) as 0-dim device tensors"""
    return _BceDice.apply(logits, target)


def loss_and_dice(outputs, target: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """The step's loss and main-output Dice for every output container of the zoo: a tensor; u2net's dict
    (sum over d0..d6, Dice of d0; training_loop.py:24-32); nested_unet's deep-supervision list (sum, Dice of the
    last = finest output)."""
    if isinstance(outputs, dict):
        pairs = [bce_dice_with_logits(v, target) for v in outputs.values()]
        return sum(p[0] for p in pairs), pairs[0][1]
    if isinstance(outputs, (list, tuple)):
        pairs = [bce_dice_with_logits(v, target) for v in outputs]
        return sum(p[0] for p in pairs), pairs[-1][1]
    return bce_dice_with_logits(outputs, target)
