"""UNet on the HIP engine (reference graph: unet_zoo/models/unet.py:8-43).

Encoder: 4 x (DoubleConv -> fused BN/ReLU/MaxPool), bottleneck DoubleConv(512->1024), decoder:
4 x (ConvTranspose2d k2s2 -> virtual concat [up, skip] -> DoubleConv), 1x1 head.  The four
skip-concats live in buffers allocated up front: the encoder writes each skip into the right
half, the transposed convolution writes the left half, so ``torch.cat`` (and its backward
split) never runs.
"""
from __future__ import annotations

import torch

from .. import ops
from ..engine import Engine
from ..graph import HipModule
from .blocks import DoubleConv, DownSample, OutConv, UpSample_UNet


class UNet(HipModule):
    def __init__(self, in_channels: int, num_classes: int):
        super().__init__()
        self.down_convolution_1 = DownSample(in_channels, 64)
        self.down_convolution_2 = DownSample(64, 128)
        self.down_convolution_3 = DownSample(128, 256)
        self.down_convolution_4 = DownSample(256, 512)

        self.bottle_neck = DoubleConv(512, 1024)

        self.up_convolution_1 = UpSample_UNet(1024, 512)
        self.up_convolution_2 = UpSample_UNet(512, 256)
        self.up_convolution_3 = UpSample_UNet(256, 128)
        self.up_convolution_4 = UpSample_UNet(128, 64)

        self.out = OutConv(in_channels=64, out_channels=num_classes)

    def emit(self, eng: Engine, x: torch.Tensor):
        N, _, H, W = x.shape
        if H < 16 or W < 16:
            raise ValueError(f"UNet needs H, W >= 16 (four 2x2 poolings), got {H}x{W}")
        downs = (self.down_convolution_1, self.down_convolution_2, self.down_convolution_3,
                 self.down_convolution_4)
        ups = (self.up_convolution_4, self.up_convolution_3, self.up_convolution_2,
               self.up_convolution_1)  # indexed by encoder level
        widths = (64, 128, 256, 512)

        cats = []
        with ops.profile_scope("doubleconv_l1"):   # (label of bench.py's block-level figure; no effect outside a profile)
            cur = eng.input_im2col(x)
        for lvl, (down, c) in enumerate(zip(downs, widths)):
            h, w = H >> lvl, W >> lvl
            full, (up_slot, skip_slot) = eng.new_cat(N, h, w, (c, c))  # cat([up, skip], 1)
            cats.append((full, up_slot))
            if lvl == 0:
                with ops.profile_scope("doubleconv_l1"):
                    _, cur = down.emit(eng, cur, skip_slot, im2col=True)
            else:
                _, cur = down.emit(eng, cur, skip_slot, im2col=False)
        cur, _ = self.bottle_neck.emit(eng, cur)
        for lvl in (3, 2, 1, 0):
            full, up_slot = cats[lvl]
            # the last block's output feeds the head only: where the kernels can, it is never written down either
            cur = ups[lvl].emit(eng, cur, full, up_slot, head=self.out.conv if lvl == 0 else None)
        return (self.out.emit(eng, cur),)
