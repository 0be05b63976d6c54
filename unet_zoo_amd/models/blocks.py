"""Parameter containers for the conv-family blocks, named exactly like the reference's modules so
``state_dict()`` keys and shapes interchange (e.g.
``down_convolution_1.conv.conv_op.0.weight``).  The ``nn.Conv2d`` / ``nn.BatchNorm2d`` children
only OWN parameters and buffers (and give the reference's default initialisation, in the
reference's construction order, so seed-0 weights are bit-identical); the arithmetic is emitted
onto the HIP engine by each block's ``emit``.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch.nn as nn

from ..engine import Engine
from ..ops import Act


def _conv_bn_relu_x2(cin: int, cout: int) -> nn.Sequential:
    # indices 0,1,(2) and 3,4,(5) as in common_layers.py:27-34 / :46-57
    return nn.Sequential(
        nn.Conv2d(cin, cout, kernel_size=3, padding=1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True),
        nn.Conv2d(cout, cout, kernel_size=3, padding=1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class DoubleConv(nn.Module):
    """[Conv3x3 -> BN -> ReLU] x 2 (reference: common_layers.py:20-37)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.conv_op = _conv_bn_relu_x2(in_channels, out_channels)

    def emit(self, eng: Engine, x: Act, *, out: Optional[Act] = None, pool: bool = False,
             im2col: bool = False, head: Optional[nn.Conv2d] = None) -> Tuple[Act, Optional[Act]]:
        s = self.conv_op
        # the middle tensor has one reader: where the kernels can, it is never written down (Engine.fold_bn_apply)
        mid, _ = eng.conv_bn_relu(x, s[0], s[1], im2col=im2col, defer_apply=s[3])
        # head = the 1x1 convolution that is the ONLY reader of this block's output (OutConv): the same for the output
        return eng.conv_bn_relu(mid, s[3], s[4], out=out, pool=pool, sole_reader=True, defer_apply=head)


class DownSample(nn.Module):
    """DoubleConv then MaxPool2d(2,2); yields (skip, pooled) (reference: common_layers.py:82-95)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.conv = DoubleConv(in_channels, out_channels)
        self.pool = nn.MaxPool2d(kernel_size=2, stride=2)

    def emit(self, eng: Engine, x: Act, skip_slot: Act, *, im2col: bool = False) -> Tuple[Act, Act]:
        # the skip is written directly into its half of the decoder's concat buffer and the
        # pool is fused into the same pass
        return self.conv.emit(eng, x, out=skip_slot, pool=True, im2col=im2col)


class UpSample_UNet(nn.Module):
    """ConvTranspose2d(k2,s2) -> cat([up, skip]) -> DoubleConv (reference: common_layers.py:97-116)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.up = nn.ConvTranspose2d(in_channels, in_channels // 2, kernel_size=2, stride=2)
        self.conv = DoubleConv(in_channels, out_channels)

    def emit(self, eng: Engine, x: Act, cat_full: Act, up_slot: Act, head: Optional[nn.Conv2d] = None) -> Act:
        # odd skip sizes: the transposed convolution lands top-left, the remaining row / column is the
        # reference's F.pad zeros (common_layers.py:110-113); handled inside conv_transpose2x2
        eng.conv_transpose2x2(x, self.up, up_slot, sole_reader=True)
        act, _ = self.conv.emit(eng, cat_full, head=head)
        return act


class OutConv(nn.Module):
    """1x1 convolution to the logits (reference: common_layers.py:118-128)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=1)

    def emit(self, eng: Engine, x: Act):
        return eng.out_conv(x, self.conv, sole_reader=True)   # every model that uses this head feeds it from its last decoder block only
