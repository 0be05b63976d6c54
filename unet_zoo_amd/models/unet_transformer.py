"""U-Transformer on the HIP engine (reference graph: unet_zoo/models/unet_transformer.py:255-283).

``inc`` (DoubleConv) and three ``Down`` blocks, ``MultiHeadSelfAttention`` on the 1/8-resolution map (:118-137) and
three ``TransformerUp`` blocks (:230-253), each a ``MultiHeadCrossAttention`` (:139-228) -- the skip S is max-pooled and
projected to values, the upsampled path Y to queries and keys, both resampled to a fixed attention grid, softmax over
the QUERY axis (``nn.Softmax(dim=1)``, as the reference has it), the result resized and concatenated with a second
projection of Y -- followed by two Conv3x3 + BN + ReLU.

On the HIP kernels: every convolution (3x3, 1x1, with or without BatchNorm), the pools (fused into the producing
BN/ReLU pass; the stand-alone one of ``Sconv_process`` through the no-ReLU pool kernel), the bilinear resizes
(align_corners=True) written into their concat slots, the position-encoding adds, the adaptive average pools onto the
attention grid, and the attention cores themselves (``Engine.token_attention``: batched products on the LDS-DMA GEMM,
column softmax in place on the (attention-grid)^2 x (attention-grid)^2 score matrices, one-tap weight-gradient kernel for
the products that contract over the queries).
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..engine import Engine
from ..graph import HipModule
from ..ops import Act
from .blocks import DoubleConv, OutConv
from .transatt_unet import Down


class MultiHeadDense(nn.Module):
    """a (d, d) matrix applied to every token (unet_transformer.py:10-32; `bias` is not supported there either)"""

    def __init__(self, d, bias=False):
        super().__init__()
        self.weight = nn.Parameter(torch.Tensor(d, d))
        if bias:
            raise NotImplementedError()
        self.register_parameter('bias', None)
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))


class PositionalEncoding2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        channels = int(np.ceil(channels / 2))
        self.channels = channels
        inv_freq = 1. / (10000 ** (torch.arange(0, channels, 2).float() / channels))
        self.register_buffer('inv_freq', inv_freq)

    def table(self, H: int, W: int, orig_ch: int) -> torch.Tensor:
        """the (H*W, orig_ch) encoding the reference adds to a (b, ch, H, W) map (unet_transformer.py:83-103 seen
        through PositionalEncodingPermute2D, :114-116): first half of the channels from the row index, second half
        from the column index"""
        inv = self.inv_freq.float().cpu()
        sx = torch.arange(H, dtype=torch.float32).unsqueeze(1) * inv.unsqueeze(0)
        sy = torch.arange(W, dtype=torch.float32).unsqueeze(1) * inv.unsqueeze(0)
        emb_x = torch.cat((sx.sin(), sx.cos()), dim=-1)          # (H, channels)
        emb_y = torch.cat((sy.sin(), sy.cos()), dim=-1)          # (W, channels)
        emb = torch.zeros(H, W, self.channels * 2)
        emb[:, :, :self.channels] = emb_x.unsqueeze(1)
        emb[:, :, self.channels:2 * self.channels] = emb_y
        return emb[:, :, :orig_ch].reshape(H * W, orig_ch).contiguous()


class PositionalEncodingPermute2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.penc = PositionalEncoding2D(channels)
        self._cache = {}

    def emit(self, eng: Engine, x: Act) -> Act:
        key = (x.H, x.W, x.C, str(eng.device))
        if key not in self._cache:
            self._cache = {key: self.penc.table(x.H, x.W, x.C).to(eng.device)}
        return eng.add_const(x, self._cache[key])


def _token_attention(eng: Engine, xq: Act, xv: Act, mod: nn.Module) -> Act:
    """softmax(Q K^T / sqrt(c), dim=1) V (unet_transformer.py:127-137, :208-219) on Engine.token_attention; token maps
    whose H * W is not a multiple of 8 run on a grid widened by zero tokens that are masked out of the softmax"""
    ws = (mod.query.weight, mod.key.weight, mod.value.weight)
    Wp = eng.padded_width(xq.H, xq.W)
    if Wp == xq.W:
        return eng.token_attention(xq, xv, *ws, eng.new_act(xq.N, xq.H, xq.W, xq.C))
    xqp = eng.pad_w(xq, Wp)
    xvp = xqp if xq is xv else eng.pad_w(xv, Wp)
    op = eng.token_attention(xqp, xvp, *ws, eng.new_act(xq.N, xq.H, Wp, xq.C), valid_w=xq.W)
    return eng.crop_w(op, xq.W)


class MultiHeadSelfAttention(nn.Module):
    def __init__(self, channel):
        super().__init__()
        self.query = MultiHeadDense(channel, bias=False)
        self.key = MultiHeadDense(channel, bias=False)
        self.value = MultiHeadDense(channel, bias=False)
        self.softmax = nn.Softmax(dim=1)
        self.pe = PositionalEncodingPermute2D(channel)

    def emit(self, eng: Engine, x: Act) -> Act:
        xp = self.pe.emit(eng, x)
        return _token_attention(eng, xp, xp, self)


class MultiHeadCrossAttention(nn.Module):
    def __init__(self, channelY, channelS, common_attn_res_for_QK_V=(64, 64)):
        super().__init__()
        self.common_attn_channels = channelS
        self.common_attn_res_for_QK_V = tuple(common_attn_res_for_QK_V)
        c = channelS
        self.Sconv_process = nn.Sequential(nn.MaxPool2d(2), nn.Conv2d(channelS, c, kernel_size=1), nn.BatchNorm2d(c),
                                           nn.ReLU(inplace=True))
        self.Yconv_process = nn.Sequential(nn.Conv2d(channelY, c, kernel_size=1), nn.BatchNorm2d(c), nn.ReLU(inplace=True))
        self.query = MultiHeadDense(c, bias=False)
        self.key = MultiHeadDense(c, bias=False)
        self.value = MultiHeadDense(c, bias=False)
        self.conv_after_attention = nn.Sequential(nn.Conv2d(c, c, kernel_size=1), nn.BatchNorm2d(c), nn.ReLU(inplace=True))
        self.Yconv2_process = nn.Sequential(
            nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True),
            nn.Conv2d(channelY, channelY, kernel_size=3, padding=1),
            nn.Conv2d(channelY, c, kernel_size=1), nn.BatchNorm2d(c), nn.ReLU(inplace=True))
        self.softmax = nn.Softmax(dim=1)
        self.Spe = PositionalEncodingPermute2D(channelS)
        self.Ype = PositionalEncodingPermute2D(channelY)

    def emit(self, eng: Engine, Y: Act, S: Act) -> Act:
        c = self.common_attn_channels
        Ha, Wa = self.common_attn_res_for_QK_V
        S_pe = self.Spe.emit(eng, S)
        Sp, _ = eng.conv_bn_relu(eng.max_pool2x2(S_pe), self.Sconv_process[1], self.Sconv_process[2])
        Y_pe = self.Ype.emit(eng, Y)
        Yp, _ = eng.conv_bn_relu(Y_pe, self.Yconv_process[0], self.Yconv_process[1])
        qk = eng.adaptive_avg_pool(Yp, Ha, Wa)
        low = _token_attention(eng, qk, eng.adaptive_avg_pool(Sp, Ha, Wa), self)
        Ho, Wo = 2 * Y.H, 2 * Y.W
        full, (z_slot, y2_slot) = eng.new_cat(Y.N, Ho, Wo, (c, c))              # cat([Z_attn, Y2_processed], 1)
        z = eng.resize_bilinear(low, eng.new_act(Y.N, Ho, Wo, c), align_corners=True)
        eng.conv_bn_relu(z, self.conv_after_attention[0], self.conv_after_attention[1], out=z_slot)
        yu = eng.resize_bilinear(Y_pe, eng.new_act(Y.N, Ho, Wo, Y.C), align_corners=True)
        y3 = eng.conv_plain(yu, self.Yconv2_process[1])
        eng.conv_bn_relu(y3, self.Yconv2_process[2], self.Yconv2_process[3], out=y2_slot)
        return full


class TransformerUp(nn.Module):
    def __init__(self, Ychannels, Schannels, common_attn_res_for_QK_V=(64, 64)):
        super().__init__()
        self.MHCA = MultiHeadCrossAttention(Ychannels, Schannels, common_attn_res_for_QK_V)
        self.conv = nn.Sequential(
            nn.Conv2d(Schannels * 2, Schannels, kernel_size=3, stride=1, padding=1, bias=True), nn.BatchNorm2d(Schannels),
            nn.ReLU(inplace=True),
            nn.Conv2d(Schannels, Schannels, kernel_size=3, stride=1, padding=1, bias=True), nn.BatchNorm2d(Schannels),
            nn.ReLU(inplace=True))

    def emit(self, eng: Engine, Y: Act, S: Act) -> Act:
        x = self.MHCA.emit(eng, Y, S)
        x, _ = eng.conv_bn_relu(x, self.conv[0], self.conv[1])
        x, _ = eng.conv_bn_relu(x, self.conv[3], self.conv[4])
        return x


class U_Transformer(HipModule):
    def __init__(self, in_channels, num_classes, bilinear=True, common_attn_res_for_QK_V=(64, 64), **kwargs):
        super().__init__()
        self.in_channels, self.classes, self.bilinear = in_channels, num_classes, bilinear
        self.inc = DoubleConv(in_channels, 64)
        self.down1 = Down(64, 128)
        self.down2 = Down(128, 256)
        self.down3 = Down(256, 512)
        self.MHSA = MultiHeadSelfAttention(512)
        self.up1 = TransformerUp(512, 256, common_attn_res_for_QK_V)
        self.up2 = TransformerUp(256, 128, common_attn_res_for_QK_V)
        self.up3 = TransformerUp(128, 64, common_attn_res_for_QK_V)
        self.outc = OutConv(64, num_classes)

    def emit(self, eng: Engine, x: torch.Tensor):
        N, _, H, W = x.shape
        if H % 8 or W % 8:
            raise ValueError(f"U_Transformer needs H, W divisible by 8 (three 2x2 poolings and x2 upsamplings), got {H}x{W}")
        x1, p1 = self.inc.emit(eng, eng.input_im2col(x), pool=True, im2col=True)
        x2, p2 = self.down1.emit(eng, p1, pool=True)
        x3, p3 = self.down2.emit(eng, p2, pool=True)
        x4, _ = self.down3.emit(eng, p3)
        y = self.MHSA.emit(eng, x4)
        y = self.up1.emit(eng, y, x3)
        y = self.up2.emit(eng, y, x2)
        y = self.up3.emit(eng, y, x1)
        return (self.outc.emit(eng, y),)
