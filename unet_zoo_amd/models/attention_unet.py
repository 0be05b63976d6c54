"""Attention U-Net on the HIP engine (reference graph: unet_zoo/models/attention_unet.py:42-110).

Encoder: 5 x ConvBlock with a fused 2x2 max-pool after the first four; decoder per level:
UpConvBlock (nearest x2 folded into the 3x3 convolution's LDS-DMA source address) -> additive
attention gate on the skip -> virtual concat (gated_skip, up) -> ConvBlock; 1x1 head.
Module / parameter names follow the reference so state_dicts interchange; ``depth`` is accepted and
unused exactly as in the reference (attention_unet.py:43).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from ..engine import Engine
from ..graph import HipModule
from ..ops import Act


class ConvBlock(nn.Module):
    """[Conv3x3 -> BN -> ReLU] x 2 (reference: common_layers.py:39-61; attribute name ``conv``)."""

    def __init__(self, ch_in: int, ch_out: int):
        super().__init__()
        self.conv = nn.Sequential(
            nn.Conv2d(ch_in, ch_out, kernel_size=3, stride=1, padding=1, bias=True),
            nn.BatchNorm2d(ch_out), nn.ReLU(inplace=True),
            nn.Conv2d(ch_out, ch_out, kernel_size=3, stride=1, padding=1, bias=True),
            nn.BatchNorm2d(ch_out), nn.ReLU(inplace=True))

    def emit(self, eng: Engine, x: Act, *, pool: bool = False, im2col: bool = False, head: Optional[nn.Conv2d] = None):
        s = self.conv
        mid, _ = eng.conv_bn_relu(x, s[0], s[1], im2col=im2col, defer_apply=s[3])   # read through BN + ReLU where the kernels can (Engine.fold_bn_apply)
        # head: the 1x1 convolution that alone reads this block's output (the same for the output, Engine.fold_bn_apply_head)
        return eng.conv_bn_relu(mid, s[3], s[4], pool=pool, sole_reader=True, defer_apply=head)


class UpConvBlock(nn.Module):
    """nearest x2 -> Conv3x3 -> BN -> ReLU (reference: common_layers.py:63-80; attribute ``up``)."""

    def __init__(self, ch_in: int, ch_out: int):
        super().__init__()
        self.up = nn.Sequential(
            nn.Upsample(scale_factor=2),
            nn.Conv2d(ch_in, ch_out, kernel_size=3, stride=1, padding=1, bias=True),
            nn.BatchNorm2d(ch_out), nn.ReLU(inplace=True))

    def emit(self, eng: Engine, x: Act, out: Act) -> Act:
        act, _ = eng.conv_bn_relu(x, self.up[1], self.up[2], out=out, upsample=True)
        return act


class AttentionBlock(nn.Module):
    """psi * x with psi = sigmoid(BN(W_psi relu(BN(W_g g) + BN(W_x x)))) (attention_unet.py:6-40)."""

    def __init__(self, f_g: int, f_l: int, f_int: int):
        super().__init__()
        self.w_g = nn.Sequential(nn.Conv2d(f_g, f_int, kernel_size=1, stride=1, padding=0, bias=True),
                                 nn.BatchNorm2d(f_int))
        self.w_x = nn.Sequential(nn.Conv2d(f_l, f_int, kernel_size=1, stride=1, padding=0, bias=True),
                                 nn.BatchNorm2d(f_int))
        self.psi = nn.Sequential(nn.Conv2d(f_int, 1, kernel_size=1, stride=1, padding=0, bias=True),
                                 nn.BatchNorm2d(1), nn.Sigmoid())
        self.relu = nn.ReLU(inplace=True)

    def emit(self, eng: Engine, g: Act, x: Act, out: Act) -> Act:
        return eng.attention_gate(g, x, self, out)


class AttentionUNet(HipModule):
    def __init__(self, in_channels: int = 3, num_classes: int = 1, depth: int = 5):
        super().__init__()
        self.maxpool = nn.MaxPool2d(kernel_size=2, stride=2)

        self.conv1 = ConvBlock(ch_in=in_channels, ch_out=64)
        self.conv2 = ConvBlock(ch_in=64, ch_out=128)
        self.conv3 = ConvBlock(ch_in=128, ch_out=256)
        self.conv4 = ConvBlock(ch_in=256, ch_out=512)
        self.conv5 = ConvBlock(ch_in=512, ch_out=1024)

        self.up5 = UpConvBlock(ch_in=1024, ch_out=512)
        self.att5 = AttentionBlock(f_g=512, f_l=512, f_int=256)
        self.upconv5 = ConvBlock(ch_in=1024, ch_out=512)

        self.up4 = UpConvBlock(ch_in=512, ch_out=256)
        self.att4 = AttentionBlock(f_g=256, f_l=256, f_int=128)
        self.upconv4 = ConvBlock(ch_in=512, ch_out=256)

        self.up3 = UpConvBlock(ch_in=256, ch_out=128)
        self.att3 = AttentionBlock(f_g=128, f_l=128, f_int=64)
        self.upconv3 = ConvBlock(ch_in=256, ch_out=128)

        self.up2 = UpConvBlock(ch_in=128, ch_out=64)
        self.att2 = AttentionBlock(f_g=64, f_l=64, f_int=32)
        self.upconv2 = ConvBlock(ch_in=128, ch_out=64)

        self.conv_1x1 = nn.Conv2d(64, num_classes, kernel_size=1, stride=1, padding=0)

    def emit(self, eng: Engine, x: torch.Tensor):
        N, _, H, W = x.shape
        if H % 16 or W % 16:
            raise ValueError(f"AttentionUNet needs H and W divisible by 16, got {H}x{W}")
        enc = (self.conv1, self.conv2, self.conv3, self.conv4)
        skips = []
        cur = eng.input_im2col(x)
        for lvl, blk in enumerate(enc):
            skip, cur = blk.emit(eng, cur, pool=True, im2col=(lvl == 0))
            skips.append(skip)
        cur, _ = self.conv5.emit(eng, cur)

        dec = ((self.up5, self.att5, self.upconv5), (self.up4, self.att4, self.upconv4),
               (self.up3, self.att3, self.upconv3), (self.up2, self.att2, self.upconv2))
        for lvl, (up, att, conv) in zip((3, 2, 1, 0), dec):
            skip = skips[lvl]
            c = skip.C
            full, (gated_slot, up_slot) = eng.new_cat(N, skip.H, skip.W, (c, c))   # cat((gated_skip, d), 1)
            d = up.emit(eng, cur, up_slot)
            att.emit(eng, d, skip, gated_slot)
            cur, _ = conv.emit(eng, full, head=self.conv_1x1 if lvl == 0 else None)
        return (eng.out_conv(cur, self.conv_1x1, sole_reader=True),)
