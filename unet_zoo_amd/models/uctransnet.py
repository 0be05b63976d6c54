"""UCTransNet on the HIP engine (reference graph: unet_zoo/models/uctransnet.py:453-496).

A UNet with 16 ... 128 channels whose four skip connections pass through a Channel Transformer (``mtc``, :312-363: patch
embeddings of the four scales onto one token grid, four ``Block_ViT`` layers of channel-wise cross attention and MLPs,
``Reconstruct`` back to each scale) and whose decoder gates each refined skip with a channel attention (``CCA``, :398-427)
before the concat.

On the HIP kernels: ConvBatchNorm / DownBlock / decoder convolutions (Conv3x3 + BN + ReLU, pools fused into the
producer), the patch-embedding convolutions (kernel = stride = patch: space-to-depth + GEMM), ``Reconstruct``'s
Conv1x1 + BN + ReLU -- evaluated on the TOKEN grid: nearest upsampling replicates every pixel, which leaves the batch
mean and the biased variance unchanged and commutes with ReLU; only the unbiased-variance factor of the running
statistics sees the larger count, which ``stat_repeat`` restores --, the global-average / scale-gradient reductions of
CCA, the 1x1 head, and the Channel Transformer itself ((image_size / 32)^2 tokens of 16 ... 240 channels): LayerNorm,
Linear, GELU, dropout and the channel-wise cross attention (``Engine.channel_cross_attention``: token-contracting
products on the one-tap weight-gradient kernel, InstanceNorm + softmax on the score planes, one batched product for the
context of all heads) as engine operations.  Configurations those kernels do not take (attention dropout in training,
widths that are not multiples of 8) are refused with NotImplementedError: the product has one backend.
"""
from __future__ import annotations

import copy
import math
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..engine import Engine
from ..graph import HipModule
from ..ops import Act


class ConfigDict(dict):
    """attribute access on a dict (common_layers.py:6-18)"""

    def __getattr__(self, key):
        if key in self:
            return self[key]
        raise AttributeError(f"'ConfigDict' object has no attribute '{key}'")

    def __setattr__(self, key, value):
        self[key] = value


def get_uctransnet_config():
    """uctransnet.py:12-31"""
    config = ConfigDict()
    config.base_channel = 16
    config.transformer = ConfigDict()
    config.transformer.embeddings_dropout_rate = 0.1
    config.transformer.attention_dropout_rate = 0.0
    config.transformer.dropout_rate = 0.1
    config.transformer.num_heads = 4
    config.transformer.num_layers = 4
    config.KV_size = sum(config.base_channel * (2 ** i) for i in range(4))
    config.patch_sizes = (32, 16, 8, 4)
    config.expand_ratio = 4
    config.vis = False
    return config


class Channel_Embeddings(nn.Module):
    def __init__(self, config, patchsize, img_size, in_channels):
        super().__init__()
        n_patches = (img_size // patchsize) * (img_size // patchsize)
        self.patch_embeddings = nn.Conv2d(in_channels, in_channels, kernel_size=patchsize, stride=patchsize)
        self.position_embeddings = nn.Parameter(torch.zeros(1, n_patches, in_channels))
        self.dropout = nn.Dropout(config.transformer["embeddings_dropout_rate"])


class Reconstruct(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, scale_factor):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, padding=1 if kernel_size == 3 else 0)
        self.norm = nn.BatchNorm2d(out_channels)
        self.activation = nn.ReLU(inplace=True)
        self.scale_factor = scale_factor


class Attention_org(nn.Module):
    def __init__(self, config, vis, channel_num):
        super().__init__()
        self.vis, self.KV_size, self.channel_num = vis, config.KV_size, channel_num
        self.num_attention_heads = config.transformer["num_heads"]
        self.query1, self.query2, self.query3, self.query4 = nn.ModuleList(), nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        self.key, self.value = nn.ModuleList(), nn.ModuleList()
        for _ in range(config.transformer["num_heads"]):
            # constructed in the reference's order (uctransnet.py:104-116): the initialisers draw from the RNG
            qs = [nn.Linear(c, c, bias=False) for c in channel_num]
            key = nn.Linear(self.KV_size, self.KV_size, bias=False)
            value = nn.Linear(self.KV_size, self.KV_size, bias=False)
            for lst, q in zip((self.query1, self.query2, self.query3, self.query4), qs):
                lst.append(copy.deepcopy(q))
            self.key.append(copy.deepcopy(key))
            self.value.append(copy.deepcopy(value))
        self.psi = nn.InstanceNorm2d(self.num_attention_heads)
        self.softmax = nn.Softmax(dim=3)
        self.out1 = nn.Linear(channel_num[0], channel_num[0], bias=False)
        self.out2 = nn.Linear(channel_num[1], channel_num[1], bias=False)
        self.out3 = nn.Linear(channel_num[2], channel_num[2], bias=False)
        self.out4 = nn.Linear(channel_num[3], channel_num[3], bias=False)
        self.attn_dropout = nn.Dropout(config.transformer["attention_dropout_rate"])
        self.proj_dropout = nn.Dropout(config.transformer["attention_dropout_rate"])



class Mlp(nn.Module):
    def __init__(self, config, in_channel, mlp_channel):
        super().__init__()
        self.fc1 = nn.Linear(in_channel, mlp_channel)
        self.fc2 = nn.Linear(mlp_channel, in_channel)
        self.act_fn = nn.GELU()
        self.dropout = nn.Dropout(config.transformer["dropout_rate"])
        nn.init.xavier_uniform_(self.fc1.weight)
        nn.init.xavier_uniform_(self.fc2.weight)
        nn.init.normal_(self.fc1.bias, std=1e-6)
        nn.init.normal_(self.fc2.bias, std=1e-6)



class Block_ViT(nn.Module):
    def __init__(self, config, vis, channel_num):
        super().__init__()
        r = config.expand_ratio
        for i, c in enumerate(channel_num):
            setattr(self, f"attn_norm{i + 1}", nn.LayerNorm(c, eps=1e-6))
        self.attn_norm = nn.LayerNorm(config.KV_size, eps=1e-6)
        self.channel_attn = Attention_org(config, vis, channel_num)
        for i, c in enumerate(channel_num):
            setattr(self, f"ffn_norm{i + 1}", nn.LayerNorm(c, eps=1e-6))
        for i, c in enumerate(channel_num):
            setattr(self, f"ffn{i + 1}", Mlp(config, c, c * r))



class Encoder(nn.Module):
    def __init__(self, config, vis, channel_num):
        super().__init__()
        self.vis = vis
        self.layer = nn.ModuleList()
        for i, c in enumerate(channel_num):
            setattr(self, f"encoder_norm{i + 1}", nn.LayerNorm(c, eps=1e-6))
        for _ in range(config.transformer["num_layers"]):
            self.layer.append(copy.deepcopy(Block_ViT(config, vis, channel_num)))



class ChannelTransformer(nn.Module):
    def __init__(self, config, vis, img_size, channel_num=(64, 128, 256, 512), patchSize=(32, 16, 8, 4)):
        super().__init__()
        self.patch = tuple(patchSize)
        for i in range(4):
            setattr(self, f"embeddings_{i + 1}", Channel_Embeddings(config, patchSize[i], img_size=img_size // (2 ** i),
                                                                    in_channels=channel_num[i]))
        self.vis = vis
        self.attn_weights = []
        self.encoder = Encoder(config, vis, channel_num)
        for i in range(4):
            setattr(self, f"reconstruct_{i + 1}", Reconstruct(channel_num[i], channel_num[i], kernel_size=1,
                                                              scale_factor=(patchSize[i], patchSize[i])))

    def _own_kernels(self, eng: Engine, toks: List[Act]) -> bool:
        """the channel transformer on the library's own kernels: every width a multiple of 8 (16-byte rows), KV <= 1024, no
        dropout on the attention probabilities / projections in training (the reference default 0.0, uctransnet.py:17);
        anything else is refused (emit)"""
        att = self.encoder.layer[0].channel_attn
        if eng.training and (att.attn_dropout.p > 0.0 or att.proj_dropout.p > 0.0):
            return False
        return all(t.C % 8 == 0 for t in toks) and att.KV_size % 8 == 0 and att.KV_size <= 1024

    def _emit_tokens(self, eng: Engine, toks: List[Act]) -> List[Act]:
        """Channel_Embeddings' position embedding + dropout, Encoder.forward over Block_ViT (uctransnet.py:50-55, :260-330) as
        engine operations on (B, h, w, C_i) token maps: LayerNorm, Linear (heads side by side in one activation), the
        channel-wise cross attention (Engine.channel_cross_attention), GELU, dropout, residual sums.  The four scales of
        a block live in the channel slots of ONE concat buffer, so `torch.cat(embs, dim=2)` (:263-268) is never executed."""
        B, h, w = toks[0].N, toks[0].H, toks[0].W
        cn = [t.C for t in toks]
        full, slots = eng.new_cat(B, h, w, cn)
        for i, t in enumerate(toks):
            emb = getattr(self, f"embeddings_{i + 1}")
            eng.dropout(eng.add_param_map(t, emb.position_embeddings), emb.dropout.p, out=slots[i])
        for blk in self.encoder.layer:
            att = blk.channel_attn
            weights = [] if self.vis else None
            H, KV = att.num_attention_heads, att.KV_size
            emb_all = eng.layer_norm(full, blk.attn_norm)
            Kall = eng.linear_heads(emb_all, att.key)
            Vall = eng.linear_heads(emb_all, att.value)
            nfull, nslots = eng.new_cat(B, h, w, cn)
            for i, (queries, proj) in enumerate(zip((att.query1, att.query2, att.query3, att.query4),
                                                    (att.out1, att.out2, att.out3, att.out4))):
                emb_i = slots[i]
                cxn = eng.layer_norm(emb_i, getattr(blk, f"attn_norm{i + 1}"))
                Qi = eng.linear_heads(cxn, queries)
                ctx = eng.channel_cross_attention(Qi, Kall, Vall, H, eps=att.psi.eps, probs_out=weights)
                cx = eng.linear(ctx, proj, residual=emb_i)                          # emb + out_i(context)
                mlp = getattr(blk, f"ffn{i + 1}")
                f = eng.layer_norm(cx, getattr(blk, f"ffn_norm{i + 1}"))
                f = eng.dropout(eng.gelu(eng.linear(f, mlp.fc1)), mlp.dropout.p)
                if eng.training and mlp.dropout.p > 0.0:
                    eng.add(eng.dropout(eng.linear(f, mlp.fc2), mlp.dropout.p), cx, out=nslots[i])
                else:
                    eng.linear(f, mlp.fc2, out=nslots[i], residual=cx)
            full, slots = nfull, nslots
            if self.vis:
                self.attn_weights.append(weights)
        return [eng.layer_norm(slots[i], getattr(self.encoder, f"encoder_norm{i + 1}")) for i in range(4)]

    def emit(self, eng: Engine, ens: List[Act], outs: List[Act]) -> List[Act]:
        toks = []
        for i, en in enumerate(ens):
            emb = getattr(self, f"embeddings_{i + 1}")
            if (en.H // self.patch[i]) * (en.W // self.patch[i]) != emb.position_embeddings.shape[1] or en.H % self.patch[i]:
                raise ValueError(f"UCTransNet was built for img_size {int(math.sqrt(emb.position_embeddings.shape[1])) * self.patch[i] * 2 ** i}"
                                 f" (square); got a {en.H * 2 ** i}x{en.W * 2 ** i} input")
            toks.append(eng.patch_conv(en, emb.patch_embeddings))
        self.attn_weights = []          # vis=True: [layer][scale] -> (B, C_i, KV), as Encoder.forward collects them (:318-322)
        if self._own_kernels(eng, toks):
            enc = self._emit_tokens(eng, toks)
        else:   # (rounds 2-4 ran these through torch GEMMs and autograd: the product has one backend)
            raise NotImplementedError("uctransnet on the HIP engine: the channel transformer's kernels take widths that are "
                                      "multiples of 8, KV_size <= 1024 and no dropout on the attention probabilities / "
                                      "projections in training (the reference's default 0.0, uctransnet.py:17)")
        for i, (e, en, out) in enumerate(zip(enc, ens, outs)):
            rec = getattr(self, f"reconstruct_{i + 1}")
            f = self.patch[i]
            r, _ = eng.conv_bn_relu(e, rec.conv, rec.norm, stat_repeat=f * f)      # on the token grid (module docstring)
            eng.upsample_nearest(r, f, out, add=en)                                 # nn.Upsample(scale_factor=p) ... + en
        return outs


class ConvBatchNorm(nn.Module):
    def __init__(self, in_channels, out_channels, activation='ReLU'):
        super().__init__()
        if activation.lower() != 'relu':
            raise NotImplementedError("UCTransNet is built with ReLU activations (uctransnet.py:365-370 default)")
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1)
        self.norm = nn.BatchNorm2d(out_channels)
        self.activation = nn.ReLU()


def _make_nConv(in_channels, out_channels, nb_Conv, activation='ReLU'):
    layers = [ConvBatchNorm(in_channels, out_channels, activation)]
    for _ in range(nb_Conv - 1):
        layers.append(ConvBatchNorm(out_channels, out_channels, activation))
    return nn.Sequential(*layers)


def _emit_convs(eng: Engine, x: Act, convs: nn.Sequential, *, out: Optional[Act] = None, pool: bool = False, im2col: bool = False):
    pooled = None
    for i, cb in enumerate(convs):
        last = i == len(convs) - 1
        x, pooled = eng.conv_bn_relu(x, cb.conv, cb.norm, out=out if last else None, pool=pool and last,
                                     im2col=im2col and i == 0)
    return x, pooled


class DownBlock(nn.Module):
    def __init__(self, in_channels, out_channels, nb_Conv, activation='ReLU'):
        super().__init__()
        self.maxpool = nn.MaxPool2d(2)
        self.nConvs = _make_nConv(in_channels, out_channels, nb_Conv, activation)


class Flatten(nn.Module):
    def forward(self, x):
        return x.view(x.size(0), -1)


class CCA(nn.Module):
    def __init__(self, F_g, F_x):
        super().__init__()
        self.mlp_x = nn.Sequential(Flatten(), nn.Linear(F_x, F_x))
        self.mlp_g = nn.Sequential(Flatten(), nn.Linear(F_g, F_x))
        self.relu = nn.ReLU(inplace=True)


class UpBlock_attention(nn.Module):
    def __init__(self, in_channels, out_channels, nb_Conv, activation='ReLU'):
        super().__init__()
        self.up = nn.Upsample(scale_factor=2)
        self.coatt = CCA(F_g=in_channels // 2, F_x=in_channels // 2)
        self.nConvs = _make_nConv(in_channels, out_channels, nb_Conv, activation)

    def emit(self, eng: Engine, x: Act, skip: Act) -> Act:
        full, (att_slot, up_slot) = eng.new_cat(skip.N, skip.H, skip.W, (skip.C, x.C))     # cat([skip_x_att, up], 1)
        eng.upsample_nearest(x, 2, up_slot)
        eng.cca_gate(x, skip, self.coatt.mlp_x[1], self.coatt.mlp_g[1], att_slot)
        y, _ = _emit_convs(eng, full, self.nConvs)
        return y


class UCTransNet(HipModule):
    def __init__(self, config, in_channels=3, num_classes=1, img_size=224, vis=False, **kwargs):
        super().__init__()
        self.vis, self.n_channels, self.n_classes, self.img_size = vis, in_channels, num_classes, img_size
        c = config.base_channel
        self.inc = ConvBatchNorm(in_channels, c)
        self.down1 = DownBlock(c, c * 2, nb_Conv=2)
        self.down2 = DownBlock(c * 2, c * 4, nb_Conv=2)
        self.down3 = DownBlock(c * 4, c * 8, nb_Conv=2)
        self.down4 = DownBlock(c * 8, c * 8, nb_Conv=2)
        self.mtc = ChannelTransformer(config, vis, img_size, channel_num=[c, c * 2, c * 4, c * 8], patchSize=config.patch_sizes)
        self.up4 = UpBlock_attention(c * 16, c * 4, nb_Conv=2)
        self.up3 = UpBlock_attention(c * 8, c * 2, nb_Conv=2)
        self.up2 = UpBlock_attention(c * 4, c, nb_Conv=2)
        self.up1 = UpBlock_attention(c * 2, c, nb_Conv=2)
        self.outc = nn.Conv2d(c, num_classes, kernel_size=(1, 1), stride=(1, 1))

    def wrap_outputs(self, outs):
        """`return logits, att_weights` with vis=True (uctransnet.py:493-496); the weights are detached fp32 tensors"""
        return (outs[0], self.mtc.attn_weights) if self.vis else outs[0]

    def emit(self, eng: Engine, x: torch.Tensor):
        N, _, H, W = x.shape
        if (H, W) != (self.img_size, self.img_size):
            raise ValueError(f"UCTransNet was built for {self.img_size}x{self.img_size} inputs (its position embeddings "
                             f"fix the token grid), got {H}x{W}")
        x1, p1 = eng.conv_bn_relu(eng.input_im2col(x), self.inc.conv, self.inc.norm, pool=True, im2col=True)
        x2, p2 = _emit_convs(eng, p1, self.down1.nConvs, pool=True)
        x3, p3 = _emit_convs(eng, p2, self.down2.nConvs, pool=True)
        x4, p4 = _emit_convs(eng, p3, self.down3.nConvs, pool=True)
        x5, _ = _emit_convs(eng, p4, self.down4.nConvs)
        ens = [x1, x2, x3, x4]
        refined = self.mtc.emit(eng, ens, [eng.new_act(e.N, e.H, e.W, e.C) for e in ens])
        y = self.up4.emit(eng, x5, refined[3])
        y = self.up3.emit(eng, y, refined[2])
        y = self.up2.emit(eng, y, refined[1])
        y = self.up1.emit(eng, y, refined[0])
        return (eng.out_conv(y, self.outc),)
