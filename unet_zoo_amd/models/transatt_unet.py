"""TransAttUNet on the HIP engine (reference graph: unet_zoo/models/transatt_unet.py:109-164 over
common_layers.py:130-180).

Encoder ``inc`` + four ``Down`` (MaxPool2d(2) then DoubleConvo), a bottleneck that adds a learned position
embedding and sums two attention branches -- ``PAM_Module`` (position attention: 1x1 q / k / v convolutions,
softmax(q^T k), transatt_unet.py:29-53) and ``ScaledDotProductAttention`` (channel attention with train-mode
dropout, :84-107) --, four ``Up`` (bilinear x2 with align_corners=True, ``cat([skip, up])``, DoubleConvo through
``in // 2`` channels) and a 1x1 head.

Every convolution, BatchNorm, pooling, resize and concat runs on the HIP kernels (the pool is fused into the BN/ReLU
pass of the producing block, the skip and the upsampled tensor are written straight into their halves of one concat
buffer).  The two attention cores at the 1/16-resolution bottleneck -- batched matrix products + softmax on (h*w) x (h*w) and
512 x 512 matrices -- run on the library's batched GEMM, row softmax and one-tap weight-gradient kernels
(``Engine.row_attention``, ``Engine.channel_attention``); only a bottleneck whose token count is not a multiple of 8
(inputs such as 48 x 48) runs on a token grid widened by masked zero tokens (``Engine.pad_w``).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..engine import Engine
from ..graph import HipModule
from ..ops import Act
from .blocks import OutConv


class DoubleConvo(nn.Module):
    """(Conv3x3 -> BN -> ReLU) x 2 through `mid_channels` (common_layers.py:130-146)"""

    def __init__(self, in_channels, out_channels, mid_channels=None):
        super().__init__()
        if not mid_channels:
            mid_channels = out_channels
        self.double_conv = nn.Sequential(
            nn.Conv2d(in_channels, mid_channels, kernel_size=3, padding=1), nn.BatchNorm2d(mid_channels), nn.ReLU(inplace=True),
            nn.Conv2d(mid_channels, out_channels, kernel_size=3, padding=1), nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True))

    def emit(self, eng: Engine, x: Act, *, out: Optional[Act] = None, pool: bool = False,
             im2col: bool = False, head: Optional[nn.Conv2d] = None) -> Tuple[Act, Optional[Act]]:
        s = self.double_conv
        mid, _ = eng.conv_bn_relu(x, s[0], s[1], im2col=im2col, defer_apply=s[3])   # Engine.fold_bn_apply
        # head: the 1x1 convolution that alone reads this block's output (Engine.fold_bn_apply_head)
        return eng.conv_bn_relu(mid, s[3], s[4], out=out, pool=pool, sole_reader=True, defer_apply=head)


class Down(nn.Module):
    """MaxPool2d(2) then DoubleConvo (common_layers.py:148-158); the pool itself is fused into the PRODUCER of this
    block's input, so emit() takes the pooled tensor"""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), DoubleConvo(in_channels, out_channels))

    def emit(self, eng: Engine, pooled: Act, **kw):
        return self.maxpool_conv[1].emit(eng, pooled, **kw)


class Up(nn.Module):
    """bilinear x2 (align_corners=True) or ConvTranspose2d, cat([skip, up]), DoubleConvo (common_layers.py:160-180)"""

    def __init__(self, in_channels, out_channels, bilinear=True):
        super().__init__()
        self.bilinear = bilinear
        if bilinear:
            self.up = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
            self.conv = DoubleConvo(in_channels, out_channels, in_channels // 2)
        else:
            self.up = nn.ConvTranspose2d(in_channels, in_channels // 2, kernel_size=2, stride=2)
            self.conv = DoubleConvo(in_channels, out_channels)

    def emit(self, eng: Engine, x1: Act, cat_full: Act, up_slot: Act, head: Optional[nn.Conv2d] = None) -> Act:
        if self.bilinear:
            eng.resize_bilinear(x1, up_slot, align_corners=True)
        else:
            eng.conv_transpose2x2(x1, self.up, up_slot)
        act, _ = self.conv.emit(eng, cat_full, head=head)
        return act


class PAM_Module(nn.Module):
    def __init__(self, in_dim):
        super().__init__()
        self.chanel_in = in_dim
        self.query_conv = nn.Conv2d(in_dim, in_dim // 8, kernel_size=1)
        self.key_conv = nn.Conv2d(in_dim, in_dim // 8, kernel_size=1)
        self.value_conv = nn.Conv2d(in_dim, in_dim, kernel_size=1)
        self.gamma = nn.Parameter(torch.zeros(1))
        self.softmax = nn.Softmax(dim=-1)

    def emit(self, eng: Engine, x: Act) -> Act:
        q = eng.conv_plain(x, self.query_conv)
        k = eng.conv_plain(x, self.key_conv)
        v = eng.conv_plain(x, self.value_conv)
        Wp = eng.padded_width(x.H, x.W)
        if Wp == x.W:
            att = eng.row_attention(q, k, v, eng.new_act(x.N, x.H, x.W, x.C))
        else:   # token counts that are not a multiple of 16 bytes (inputs like 48 x 48): a widened grid, the added keys masked
            attp = eng.row_attention(eng.pad_w(q, Wp), eng.pad_w(k, Wp), eng.pad_w(v, Wp), eng.new_act(x.N, x.H, Wp, x.C),
                                     valid_w=x.W)
            att = eng.crop_w(attp, x.W)
        return eng.scale_residual(att, self.gamma, x)


class PositionEmbeddingLearned(nn.Module):
    def __init__(self, num_pos_feats=256, len_embedding=32):
        super().__init__()
        self.row_embed = nn.Embedding(len_embedding, num_pos_feats)
        self.col_embed = nn.Embedding(len_embedding, num_pos_feats)
        nn.init.uniform_(self.row_embed.weight)
        nn.init.uniform_(self.col_embed.weight)

    @staticmethod
    def _add(x, row_w, col_w):
        """x + cat([col_embed(i) over rows, row_embed(j) over columns]) (transatt_unet.py:66-82, :144-145)"""
        h, w = x.shape[-2:]
        x_emb, y_emb = col_w[:w], row_w[:h]
        pos = torch.cat([x_emb.unsqueeze(0).expand(h, w, -1), y_emb.unsqueeze(1).expand(h, w, -1)], dim=-1)
        return x + pos.permute(2, 0, 1).unsqueeze(0)

    def emit(self, eng: Engine, x: Act) -> Act:
        if x.H > self.row_embed.num_embeddings or x.W > self.col_embed.num_embeddings:
            raise IndexError(f"position embedding holds {self.row_embed.num_embeddings} rows / columns, the bottleneck "
                             f"map is {x.H}x{x.W} (input larger than 512x512)")
        return eng.add_row_col_embed(x, self.row_embed.weight, self.col_embed.weight)


class ScaledDotProductAttention(nn.Module):
    def __init__(self, temperature, attn_dropout=0.1):
        super().__init__()
        self.temperature = temperature ** 0.5
        self.dropout = nn.Dropout(attn_dropout)

    def emit(self, eng: Engine, x: Act) -> Act:
        Wp = eng.padded_width(x.H, x.W)
        if Wp == x.W:
            return eng.channel_attention(x, self.temperature, self.dropout.p, eng.new_act(x.N, x.H, x.W, x.C))
        # zero tokens add nothing to the channel Gram matrix and receive attn @ 0 = 0: no mask needed
        outp = eng.channel_attention(eng.pad_w(x, Wp), self.temperature, self.dropout.p, eng.new_act(x.N, x.H, Wp, x.C))
        return eng.crop_w(outp, x.W)


class TransAttUNet(HipModule):
    def __init__(self, in_channels=3, num_classes=1, bilinear=True, **kwargs):
        super().__init__()
        self.n_channels, self.n_classes, self.bilinear = in_channels, num_classes, bilinear
        self.inc = DoubleConvo(in_channels, 64)
        self.down1 = Down(64, 128)
        self.down2 = Down(128, 256)
        self.down3 = Down(256, 512)
        factor = 2 if bilinear else 1
        self.down4 = Down(512, 1024 // factor)
        self.up1 = Up((1024 // factor) + 512, 512 // factor, bilinear)
        self.up2 = Up((512 // factor) + 256, 256 // factor, bilinear)
        self.up3 = Up((256 // factor) + 128, 128 // factor, bilinear)
        self.up4 = Up((128 // factor) + 64, 64, bilinear)
        self.outc = OutConv(64, num_classes)
        self.pos = PositionEmbeddingLearned(256)
        self.pam = PAM_Module(512)
        self.sdpa = ScaledDotProductAttention(512)

    def emit(self, eng: Engine, x: torch.Tensor):
        N, _, H, W = x.shape
        if H % 16 or W % 16:
            raise ValueError(f"TransAttUNet needs H, W divisible by 16 (four 2x2 poolings and x2 upsamplings), got {H}x{W}")
        factor = 2 if self.bilinear else 1
        enc = (self.inc, self.down1, self.down2, self.down3)
        skips_c = (64, 128, 256, 512)
        ups = (self.up4, self.up3, self.up2, self.up1)         # indexed by encoder level
        # channels of the tensor that arrives from below at encoder level lvl (after the optional ConvTranspose2d)
        below_c = [128 // factor, 256 // factor, 512 // factor, 1024 // factor]
        up_c = [c if self.bilinear else c // 2 for c in below_c]
        cats = []
        cur = eng.input_im2col(x)
        for lvl, (blk, c) in enumerate(zip(enc, skips_c)):
            full, (skip_slot, up_slot) = eng.new_cat(N, H >> lvl, W >> lvl, (c, up_c[lvl]))   # cat([x2, x1], 1)
            cats.append((full, up_slot))
            _, cur = blk.emit(eng, cur, out=skip_slot, pool=True, im2col=(lvl == 0))
        x5, _ = self.down4.emit(eng, cur)
        if x5.C != 512:
            raise ValueError("the attention bottleneck of TransAttUNet is built for 512 channels: bilinear=True only, "
                             "as in the reference (transatt_unet.py:135-141)")
        x5 = self.pos.emit(eng, x5)
        fused = eng.add(self.sdpa.emit(eng, x5), self.pam.emit(eng, x5))
        cur = fused
        for lvl in (3, 2, 1, 0):
            full, up_slot = cats[lvl]
            cur = ups[lvl].emit(eng, cur, full, up_slot, head=self.outc.conv if lvl == 0 else None)
        return (self.outc.emit(eng, cur),)
