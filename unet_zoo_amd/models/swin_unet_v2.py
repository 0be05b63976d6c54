"""Swin-UNet V2 on the HIP engine (reference graph: unet_zoo/models/swin_unet_v2.py:11-761).

The module tree (names, registration order, default initialisation followed by ``trunc_normal_`` on
every Linear / constants on every LayerNorm through ``self.apply``) is the reference's, so
``state_dict()`` keys, shapes and the seed-0 tensors interchange bit for bit — including the members the
reference constructs but never calls: every block's ``mlp`` and ``norm2`` (its ``forward`` returns after
``shortcut + drop_path(norm1(attn))``, swin_unet_v2.py:264-267).  They own parameters here too and,
exactly as in the reference, receive no gradient.

Lowering (tokens are NHWC activations on the token grid):

* ``PatchEmbed``: patch extraction kernel + GEMM, LayerNorm.
* block: qkv Linear (LDS-DMA GEMM) -> window-attention core kernel (cosine attention, tau, continuous
  position bias, shift mask; ``torch.roll`` / ``window_partition`` / ``window_reverse`` are index
  arithmetic inside it) -> proj Linear -> LayerNorm kernel that also adds the shortcut and applies the
  stochastic-depth factor.
* ``PatchMerging``: LayerNorm(4C) reads the four strided neighbours directly (no concat), Linear.
* ``PatchExpand``: Linear whose GEMM epilogue stores with the 2x2 pixel shuffle, LayerNorm;
  ``FinalPatchExpand_X4``: Linear, LayerNorm reading through the 4x4 rearrangement.
* decoder ``torch.cat([x, skip], -1)``: one buffer, both halves written in place by their producers.
* head: 1x1 convolution to NCHW fp32 logits.

Dropout members are kept for the module tree; non-zero ``drop_rate`` / ``attn_drop_rate`` and the
absolute position embedding are not implemented (the reference's defaults are 0 / off).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from .. import _lib as L
from ..engine import Engine
from ..graph import HipModule
from ..ops import Act


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


class DropPath(nn.Module):
    """Stochastic depth: per-sample Bernoulli(keep) / keep on the residual branch in training
    (what timm.models.layers.DropPath, imported at swin_unet_v2.py:9, does).  The engine takes the
    per-sample factor from :meth:`factor` and applies it inside the LayerNorm kernel."""

    def __init__(self, drop_prob: float = 0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)

    def factor(self, batch: int, device, training: bool) -> Optional[torch.Tensor]:
        if self.drop_prob == 0.0 or not training:
            return None
        drawn, self._drawn = getattr(self, "_drawn", None), None
        if drawn is not None and drawn.shape[0] == batch:
            return drawn                                     # this forward's row of draw_all()
        keep = 1.0 - self.drop_prob
        return torch.empty(batch, dtype=torch.float32, device=device).bernoulli_(keep) / keep

    @staticmethod
    def draw_all(root: nn.Module, batch: int, device, training: bool) -> None:
        """One Bernoulli draw for every DropPath under `root` (independent per layer and sample, as in the
        reference) instead of two tiny kernels per block: each module picks its row up in factor()."""
        mods = [m for m in root.modules() if isinstance(m, DropPath) and m.drop_prob > 0.0]
        if not mods or not training:
            return
        cache = getattr(root, "_drop_path_keep", None)
        if cache is None or cache.device != torch.device(device) or cache.shape[0] != len(mods):
            cache = torch.tensor([1.0 - m.drop_prob for m in mods], dtype=torch.float32, device=device).view(-1, 1)
            root._drop_path_keep = cache
        f = torch.bernoulli(cache.expand(len(mods), batch)) / cache
        for i, m in enumerate(mods):
            m._drawn = f[i]


class Mlp(nn.Module):
    """Constructed by every block, never called by the reference (swin_unet_v2.py:11-28, :264-267)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)


class Mlp_Relu(nn.Module):
    """Linear-ReLU-Linear of the continuous position bias (swin_unet_v2.py:58-72)."""

    def __init__(self, in_features, hidden_features, out_features, dropout=0.0):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.act = nn.ReLU()
        self.dropout = nn.Dropout(dropout)


class WindowAttention(nn.Module):
    """Parameters of the cosine window attention (swin_unet_v2.py:74-159)."""

    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        if dim % num_heads:
            raise ValueError(f"dim {dim} is not divisible by num_heads {num_heads}")
        # head_dim 32 runs on the fused MFMA / VALU kernels; other widths and attention dropout take the engine's
        # library-GEMM window core.  (qk_scale cancels in q.k / (|q||k|) except inside the 1e-6 clamp.)
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        ch, cw = torch.arange(window_size[0]), torch.arange(window_size[1])
        coords = torch.stack(torch.meshgrid([ch, cw], indexing="ij")).flatten(1)
        rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
        self.register_buffer("log_relative_position_index", torch.sign(rel) * torch.log(1.0 + rel.abs()))
        self.cpb = Mlp_Relu(2, 256, num_heads, dropout=0.0)
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.softmax = nn.Softmax(dim=-1)
        n = window_size[0] * window_size[1]
        self.tau = nn.Parameter(torch.ones(num_heads, n, n))


class SwinTransformerBlock(nn.Module):
    """(Shifted-)window attention block (swin_unet_v2.py:177-269)."""

    def __init__(self, dim, input_resolution, num_heads, window_size=7, shift_size=0, mlp_ratio=4.0, qkv_bias=True,
                 qk_scale=None, drop=0.0, attn_drop=0.0, drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim, self.input_resolution, self.num_heads = dim, input_resolution, num_heads
        self.window_size, self.shift_size = window_size, shift_size
        if min(input_resolution) <= window_size:          # one window covers the map: no shift (:199-202)
            self.shift_size = 0
            self.window_size = min(input_resolution)
        assert 0 <= self.shift_size < self.window_size
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, _pair(self.window_size), num_heads, qkv_bias, qk_scale, attn_drop, drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        if self.shift_size > 0:
            # the reference registers the 0 / -100 mask as a buffer (:214-238); kept for state_dict parity,
            # the kernel derives the same mask from the region ids
            H, W = input_resolution
            ws, s = self.window_size, self.shift_size
            img = torch.zeros(1, H, W, 1)
            cnt = 0
            for hs in (slice(0, -ws), slice(-ws, -s), slice(-s, None)):
                for wsl in (slice(0, -ws), slice(-ws, -s), slice(-s, None)):
                    img[:, hs, wsl, :] = cnt
                    cnt += 1
            mw = img.view(1, H // ws, ws, W // ws, ws, 1).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws)
            am = mw.unsqueeze(1) - mw.unsqueeze(2)
            am = am.masked_fill(am != 0, -100.0).masked_fill(am == 0, 0.0)
        else:
            am = None
        self.register_buffer("attn_mask", am)

    def emit(self, eng: Engine, x: Act, out: Optional[Act] = None) -> Act:
        H, W = self.input_resolution
        assert (x.H, x.W, x.C) == (H, W, self.dim), "input feature has wrong size"
        a = eng.window_attention(x, self.attn, self.num_heads, self.window_size, self.shift_size)
        f = self.drop_path.factor(x.N, eng.device, eng.training) if isinstance(self.drop_path, DropPath) else None
        return eng.layer_norm(a, self.norm1, out=out, residual=x, image_scale=f)


class PatchMerging(nn.Module):
    """2x2 neighbourhood concat -> LayerNorm(4C) -> Linear(4C, 2C) (swin_unet_v2.py:298-332)."""

    def __init__(self, input_resolution, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.input_resolution, self.dim = input_resolution, dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = norm_layer(4 * dim)

    def emit(self, eng: Engine, x: Act, out: Optional[Act] = None) -> Act:
        return eng.linear(eng.layer_norm(x, self.norm, mode=L.LN_MERGE), self.reduction, out=out)


class PatchExpand(nn.Module):
    """Linear(C, 2C) -> 2x2 rearrange -> LayerNorm(C/2) (swin_unet_v2.py:342-362)."""

    def __init__(self, input_resolution, dim, dim_scale=2, norm_layer=nn.LayerNorm):
        super().__init__()
        self.input_resolution, self.dim = input_resolution, dim
        if dim_scale != 2:
            raise NotImplementedError("PatchExpand is only used with dim_scale=2")
        self.expand = nn.Linear(dim, 2 * dim, bias=False)
        self.norm = norm_layer(dim // dim_scale)

    def emit(self, eng: Engine, x: Act, out: Optional[Act] = None) -> Act:
        return eng.layer_norm(eng.linear_expand2(x, self.expand), self.norm, out=out)


class FinalPatchExpand_X4(nn.Module):
    """Linear(C, 16C) -> 4x4 rearrange -> LayerNorm(C) (swin_unet_v2.py:364-387)."""

    def __init__(self, input_resolution, dim, dim_scale=4, norm_layer=nn.LayerNorm):
        super().__init__()
        self.input_resolution, self.dim, self.dim_scale = input_resolution, dim, dim_scale
        self.expand = nn.Linear(dim, 16 * dim, bias=False)
        self.output_dim = dim
        self.norm = norm_layer(self.output_dim)

    def emit(self, eng: Engine, x: Act) -> Act:
        return eng.layer_norm(eng.linear(x, self.expand), self.norm, mode=L.LN_EXPAND, r=self.dim_scale)

    def emit_head(self, eng: Engine, x: Act, conv: nn.Conv2d) -> torch.Tensor:
        """this module followed by the 1x1 `conv` to the logits, the normalised map kept in registers"""
        return eng.layer_norm_head(eng.linear(x, self.expand), self.norm, conv, mode=L.LN_EXPAND, r=self.dim_scale)


def _blocks(dim, input_resolution, depth, num_heads, window_size, mlp_ratio, qkv_bias, qk_scale, drop, attn_drop,
            drop_path, norm_layer) -> nn.ModuleList:
    return nn.ModuleList([
        SwinTransformerBlock(dim=dim, input_resolution=input_resolution, num_heads=num_heads, window_size=window_size,
                             shift_size=0 if i % 2 == 0 else window_size // 2, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                             qk_scale=qk_scale, drop=drop, attn_drop=attn_drop,
                             drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path,
                             act_layer=nn.GELU, norm_layer=norm_layer)
        for i in range(depth)])


class BasicLayer(nn.Module):
    """Encoder stage: blocks, then PatchMerging (swin_unet_v2.py:389-455)."""

    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
                 drop=0.0, attn_drop=0.0, drop_path=0.0, norm_layer=nn.LayerNorm, downsample=None, use_checkpoint=False):
        super().__init__()
        self.dim, self.input_resolution, self.depth, self.use_checkpoint = dim, input_resolution, depth, use_checkpoint
        self.blocks = _blocks(dim, input_resolution, depth, num_heads, window_size, mlp_ratio, qkv_bias, qk_scale, drop,
                              attn_drop, drop_path, norm_layer)
        self.downsample = downsample(input_resolution, dim=dim, norm_layer=norm_layer) if downsample is not None else None

    def emit(self, eng: Engine, x: Act, out: Optional[Act] = None) -> Act:
        for i, blk in enumerate(self.blocks):
            last = i == len(self.blocks) - 1 and self.downsample is None
            x = blk.emit(eng, x, out=out if last else None)
        if self.downsample is not None:
            x = self.downsample.emit(eng, x, out=out)
        return x


class BasicLayer_up(nn.Module):
    """Decoder stage: blocks, then PatchExpand (swin_unet_v2.py:464-522)."""

    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
                 drop=0.0, attn_drop=0.0, drop_path=0.0, norm_layer=nn.LayerNorm, upsample=None, use_checkpoint=False):
        super().__init__()
        self.dim, self.input_resolution, self.depth, self.use_checkpoint = dim, input_resolution, depth, use_checkpoint
        self.blocks = _blocks(dim, input_resolution, depth, num_heads, window_size, mlp_ratio, qkv_bias, qk_scale, drop,
                              attn_drop, drop_path, norm_layer)
        self.upsample = PatchExpand(input_resolution, dim=dim, dim_scale=2, norm_layer=norm_layer) \
            if upsample is not None else None

    def emit(self, eng: Engine, x: Act, out: Optional[Act] = None) -> Act:
        for blk in self.blocks:
            x = blk.emit(eng, x)
        if self.upsample is not None:
            x = self.upsample.emit(eng, x, out=out)
        return x


class PatchEmbed(nn.Module):
    """Conv2d(k = s = patch) -> tokens -> LayerNorm (swin_unet_v2.py:524-556)."""

    def __init__(self, img_size=224, patch_size=4, in_chans=3, embed_dim=96, norm_layer=None):
        super().__init__()
        self.img_size, self.patch_size = _pair(img_size), _pair(patch_size)
        self.patches_resolution = [self.img_size[0] // self.patch_size[0], self.img_size[1] // self.patch_size[1]]
        self.num_patches = self.patches_resolution[0] * self.patches_resolution[1]
        self.in_chans, self.embed_dim = in_chans, embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None

    def emit(self, eng: Engine, x: torch.Tensor, out: Optional[Act] = None) -> Act:
        _, _, H, W = x.shape
        if (H, W) != self.img_size:
            raise AssertionError(f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]}).")
        t = eng.patch_embed(x, self.proj)
        if self.norm is None:          # patch_norm=False (swin_unet_v2.py:555: `if self.norm is not None`)
            return t if out is None else eng.copy_into(t, out)
        return eng.layer_norm(t, self.norm, out=out)


class SwinTransformerSys(HipModule):
    """Swin-UNet V2 (swin_unet_v2.py:569-761), same constructor signature."""

    def __init__(self, img_size=224, patch_size=4, in_chans=3, num_classes=1000, embed_dim=96, depths=[2, 2, 2, 2],
                 depths_decoder=[1, 2, 2, 2], num_heads=[3, 6, 12, 24], window_size=7, mlp_ratio=4.0, qkv_bias=True,
                 qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.1, norm_layer=nn.LayerNorm, ape=False,
                 patch_norm=True, use_checkpoint=False, final_upsample="expand_first", **kwargs):
        super().__init__()
        if final_upsample != "expand_first":
            raise NotImplementedError("only the reference's default final_upsample='expand_first' is implemented (with any "
                                      "other value the reference builds no head at all, swin_unet_v2.py:688-692)")
        self.num_classes, self.num_layers, self.embed_dim = num_classes, len(depths), embed_dim
        self.ape, self.patch_norm = ape, patch_norm
        self.num_features = int(embed_dim * 2 ** (self.num_layers - 1))
        self.mlp_ratio, self.final_upsample = mlp_ratio, final_upsample
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                                      norm_layer=norm_layer if patch_norm else None)
        res = self.patch_embed.patches_resolution
        self.patches_resolution = res
        if self.ape:                                  # swin_unet_v2.py:624-626
            self.absolute_pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches, embed_dim))
            nn.init.trunc_normal_(self.absolute_pos_embed, std=.02)
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
        nl = self.num_layers
        self.layers = nn.ModuleList()
        for i in range(nl):
            self.layers.append(BasicLayer(
                dim=int(embed_dim * 2 ** i), input_resolution=(res[0] // 2 ** i, res[1] // 2 ** i), depth=depths[i],
                num_heads=num_heads[i], window_size=window_size, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])],
                norm_layer=norm_layer, downsample=PatchMerging if i < nl - 1 else None, use_checkpoint=use_checkpoint))
        self.layers_up = nn.ModuleList()
        self.concat_back_dim = nn.ModuleList()
        for i in range(nl):
            lvl = nl - 1 - i
            dim = int(embed_dim * 2 ** lvl)
            r = (res[0] // 2 ** lvl, res[1] // 2 ** lvl)
            concat_linear = nn.Linear(2 * dim, dim) if i > 0 else nn.Identity()
            if i == 0:
                layer_up = PatchExpand(input_resolution=r, dim=dim, dim_scale=2, norm_layer=norm_layer)
            else:
                layer_up = BasicLayer_up(
                    dim=dim, input_resolution=r, depth=depths[lvl], num_heads=num_heads[lvl], window_size=window_size,
                    mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate, attn_drop=attn_drop_rate,
                    drop_path=dpr[sum(depths[:lvl]):sum(depths[:lvl + 1])], norm_layer=norm_layer,
                    upsample=PatchExpand if i < nl - 1 else None, use_checkpoint=use_checkpoint)
            self.layers_up.append(layer_up)
            self.concat_back_dim.append(concat_linear)
        self.norm = norm_layer(self.num_features)
        self.norm_up = norm_layer(self.embed_dim)
        self.up = FinalPatchExpand_X4(input_resolution=(img_size // patch_size, img_size // patch_size), dim_scale=4,
                                      dim=embed_dim)
        self.output = nn.Conv2d(in_channels=embed_dim, out_channels=self.num_classes, kernel_size=1, bias=False)
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def emit(self, eng: Engine, x: torch.Tensor):
        nl, E = self.num_layers, self.embed_dim
        N = x.shape[0]
        R = self.patches_resolution
        # decoder level lvl (< nl-1) reads cat([expanded deeper map, encoder-stage input lvl], -1): one buffer
        cats: List = []
        for lvl in range(nl - 1):
            cats.append(eng.new_cat(N, R[0] >> lvl, R[1] >> lvl, (E << lvl, E << lvl)))
        DropPath.draw_all(self, N, eng.device, eng.training)
        # (only the blocks whose window core runs on the fused kernels: Engine.window_attention sends the others --
        # head_dim != 32, attention dropout in training -- through its library-GEMM core, position MLP included)
        eng.position_biases([(m.attn, m.window_size) for m in self.modules() if isinstance(m, SwinTransformerBlock)
                             and m.dim // m.num_heads == 32 and not (eng.training and m.attn.attn_drop.p > 0)])
        skip0 = cats[0][1][1]
        if self.ape or (self.pos_drop.p > 0 and eng.training):     # x + absolute_pos_embed, pos_drop (:713-716)
            t = self.patch_embed.emit(eng, x)
            if self.ape:
                t = eng.add_param_map(t, self.absolute_pos_embed, out=None if self.pos_drop.p > 0 else skip0)
            t = eng.dropout(t, self.pos_drop.p, out=skip0)
        else:
            t = self.patch_embed.emit(eng, x, out=skip0)
        for i, layer in enumerate(self.layers):                 # forward_features (:711-723)
            nxt = cats[i + 1][1][1] if i + 1 < nl - 1 else None  # stage i's output is stage i+1's skip
            t = layer.emit(eng, t, out=nxt)
        t = eng.layer_norm(t, self.norm)
        for inx, layer_up in enumerate(self.layers_up):         # forward_up_features (:725-741)
            lvl = nl - 1 - inx
            dst = cats[lvl - 1][1][0] if lvl >= 1 else None     # left half of the next level's concat buffer
            if inx == 0:
                t = layer_up.emit(eng, t, out=dst)
                continue
            t = eng.linear(cats[lvl][0], self.concat_back_dim[inx])
            t = layer_up.emit(eng, t, out=dst)
        t = eng.layer_norm(t, self.norm_up)
        return (self.up.emit_head(eng, t, self.output),)         # up_x4 + output (:743-754)
