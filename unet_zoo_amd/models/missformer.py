"""MISSFormer on the HIP engine (reference graph: unet_zoo/models/missformer.py:866-939 — MiT-B1 encoder :302-368,
BridegeBlock_4 :765-813 over BridgeLayer_4 :635-702, four SegU_decoder stages :566-633).

Tokens are NHWC activations, so every `permute / reshape / flatten` between the reference's (B, C, H, W) maps and
(B, N, C) token tensors disappears.  What runs on the GPU:

  * Linear layers, the stride-2 3x3 patch embeddings and the r x r spatial-reduction convolutions (space-to-depth +
    GEMM) on the LDS-DMA GEMM kernels; the 7x7 stride-4 embedding of the image as im2col + GEMM;
  * LayerNorm, PatchExpand / FinalPatchExpand_X4 with the rearrange folded into the GEMM store / the LayerNorm
    addressing, and the last 1x1 convolution fused behind the final LayerNorm (the kernels of swin_unet_v2);
  * the attention softmax(q k^T / 8) v with head_dim 64 against the reduced keys on MFMA (`uz_sra_*`);
  * MixFFN_skip: fc1 -> depthwise 3x3 + skip (`uz_dwconv3x3`) -> LayerNorm with the GELU folded in -> fc2.

The bridge's token concat over the four scales (`torch.cat([c1f, c2f, c3f, c4f], -2)`, :681) is ONE buffer stored
scale-major — block s holds the (B, n_s, 64) tokens of scale s — so the per-token layers run once over all
B * sum(n_s) rows, each scale's producers and consumers use their block in place, and the keys of the four scales
are addressed by the attention kernel as four blocks of (B, kps) rows (softmax over a key set does not depend on
the order of the keys).

Module registration order, names and initialisation follow the reference constructor, so `state_dict()` keys and a
seed-0 construction match it tensor for tensor (tests/golden/missformer_manifest.json).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from ..engine import Engine
from ..graph import HipModule
from ..ops import Act
from .swin_unet_v2 import FinalPatchExpand_X4 as _SwinFinalExpand

segformer_settings = {
    'B0': [[32, 64, 160, 256], [2, 2, 2, 2], 256],
    'B1': [[64, 128, 320, 512], [2, 2, 2, 2], 256],
    'B2': [[64, 128, 320, 512], [3, 4, 6, 3], 768],
    'B3': [[64, 128, 320, 512], [3, 4, 18, 3], 768],
    'B4': [[64, 128, 320, 512], [3, 8, 27, 3], 768],
    'B5': [[64, 128, 320, 512], [3, 6, 40, 3], 768],
}


def _head_check(dim: int, head: int) -> None:
    if dim % head or dim // head != 64:
        raise NotImplementedError(f"the attention kernel is built for head_dim 64, got dim={dim}, heads={head}")


class EfficientSelfAtten(nn.Module):
    """missformer.py:7-39"""

    def __init__(self, dim, head, reduction_ratio):
        super().__init__()
        _head_check(dim, head)
        self.head, self.reduction_ratio = head, reduction_ratio
        self.scale = (dim // head) ** -0.5
        self.q = nn.Linear(dim, dim, bias=True)
        self.kv = nn.Linear(dim, dim * 2, bias=True)
        self.proj = nn.Linear(dim, dim)
        if reduction_ratio > 1:
            self.sr = nn.Conv2d(dim, dim, reduction_ratio, reduction_ratio)
            self.norm = nn.LayerNorm(dim)

    def emit(self, eng: Engine, x: Act, residual: Optional[Act] = None) -> Act:
        q = eng.linear(x, self.q)
        red = eng.layer_norm(eng.patch_conv(x, self.sr), self.norm) if self.reduction_ratio > 1 else x
        kv = eng.linear(red, self.kv)
        o = eng.sr_attention(q, kv, x.N, self.head, kv.P // x.N, self.scale)
        return eng.linear(o, self.proj, residual=residual)


class Scale_reduce(nn.Module):
    """missformer.py:65-100 (parameters; M_EfficientSelfAtten.emit does the arithmetic)"""

    def __init__(self, dim, reduction_ratios, patch_resolutions, mi_t_dims):
        super().__init__()
        self.dim, self.reduction_ratios, self.patch_resolutions, self.mi_t_dims = dim, reduction_ratios, patch_resolutions, mi_t_dims
        self.sr_convs = nn.ModuleList()
        for r in reduction_ratios:
            self.sr_convs.append(nn.Conv2d(dim, dim, r, r) if r > 1 else nn.Identity())
        self.norm = nn.LayerNorm(dim)


class M_EfficientSelfAtten(nn.Module):
    """missformer.py:102-128: queries = all tokens of the four scales, keys = their spatially reduced tokens"""

    def __init__(self, dim, head, reduction_ratios, patch_resolutions, mi_t_dims):
        super().__init__()
        _head_check(dim, head)
        self.head = head
        self.scale = (dim // head) ** -0.5
        self.q = nn.Linear(dim, dim, bias=True)
        self.kv = nn.Linear(dim, dim * 2, bias=True)
        self.proj = nn.Linear(dim, dim)
        self.scale_reduce = Scale_reduce(dim, reduction_ratios, patch_resolutions, mi_t_dims)

    def emit(self, eng: Engine, x: Act, B: int, residual: Optional[Act] = None) -> Act:
        sr = self.scale_reduce
        res, ratios = sr.patch_resolutions, sr.reduction_ratios
        shapes = [(B, h, w) for h, w in res]
        red_shapes = [(B, h // r, w // r) for (h, w), r in zip(res, ratios)]
        kps = red_shapes[0][1] * red_shapes[0][2]
        if any(h * w != kps for _, h, w in red_shapes):
            raise NotImplementedError("the bridge attention needs the same number of reduced tokens at every scale "
                                      f"(image size divisible by 32), got {red_shapes}")
        q = eng.linear(x, self.q)
        views = eng.row_views(x, shapes)
        red, slots = eng.new_rows(red_shapes, x.C)
        for v, conv, slot in zip(views, sr.sr_convs, slots):
            if isinstance(conv, nn.Conv2d):
                eng.patch_conv(v, conv, out=slot)
            else:
                eng.copy_into(v, slot)
        kv = eng.linear(eng.layer_norm(red, sr.norm), self.kv)
        segs, r0 = [], 0
        for _, h, w in shapes:
            segs.append((r0, h * w))
            r0 += B * h * w
        o = eng.sr_attention(q, kv, B, self.head, kps, self.scale, segments=segs)
        return eng.linear(o, self.proj, residual=residual)


class DWConv(nn.Module):
    """missformer.py:168-177"""

    def __init__(self, dim):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, 3, 1, 1, groups=dim)


class MixFFN_skip(nn.Module):
    """missformer.py:192-208 (norm2 / norm3 are registered and never used there either)"""

    def __init__(self, c1, c2):
        super().__init__()
        self.fc1 = nn.Linear(c1, c2)
        self.dwconv = DWConv(c2)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(c2, c1)
        self.norm1 = nn.LayerNorm(c2)
        self.norm2 = nn.LayerNorm(c2)
        self.norm3 = nn.LayerNorm(c2)

    def emit(self, eng: Engine, x: Act, out: Optional[Act] = None, residual: Optional[Act] = None) -> Act:
        f = eng.linear(x, self.fc1)
        a = eng.layer_norm(eng.dwconv_skip(f, self.dwconv.dwconv), self.norm1, gelu=True)   # act(norm1(.)) in one kernel
        return eng.linear(a, self.fc2, out=out, residual=residual)


class MixFFN(nn.Module):
    """missformer.py:179-190: fc1 -> depthwise 3x3 -> GELU -> fc2 (token_mlp='mix')"""

    def __init__(self, c1, c2):
        super().__init__()
        self.fc1 = nn.Linear(c1, c2)
        self.dwconv = DWConv(c2)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(c2, c1)

    def emit(self, eng: Engine, x: Act, out: Optional[Act] = None, residual: Optional[Act] = None) -> Act:
        a = eng.gelu(eng.dwconv_skip(eng.linear(x, self.fc1), self.dwconv.dwconv, skip=False))
        return eng.linear(a, self.fc2, out=out, residual=residual)


class MLP_FFN(nn.Module):
    """missformer.py:210-221 (token_mlp='fc').  The reference constructs it, but TransformerBlock.forward calls
    `self.mlp(x, H, W)` (:267) while MLP_FFN.forward takes only x: a TypeError on the first forward -- mirrored."""

    def __init__(self, c1, c2):
        super().__init__()
        self.fc1 = nn.Linear(c1, c2)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(c2, c1)

    def emit(self, eng: Engine, x: Act, out: Optional[Act] = None, residual: Optional[Act] = None) -> Act:
        raise TypeError("MLP_FFN.forward() takes 2 positional arguments but 4 were given "
                        "(token_mlp='fc' fails the same way in the reference: missformer.py:216 vs :267)")


class OverlapPatchEmbeddings(nn.Module):
    """missformer.py:238-250"""

    def __init__(self, img_size=224, patch_size=7, stride=4, padding=1, in_ch=3, dim=768):
        super().__init__()
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_ch, dim, patch_size, stride, padding)
        self.norm = nn.LayerNorm(dim)

    def emit(self, eng: Engine, x) -> Act:
        y = eng.conv_input(x, self.proj) if isinstance(x, torch.Tensor) else eng.conv3x3_s2(x, self.proj)
        return eng.layer_norm(y, self.norm)


class TransformerBlock(nn.Module):
    """missformer.py:252-268"""

    def __init__(self, dim, head, reduction_ratio=1, token_mlp='mix'):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = EfficientSelfAtten(dim, head, reduction_ratio)
        self.norm2 = nn.LayerNorm(dim)
        if token_mlp == 'mix':                       # missformer.py:258-263
            self.mlp = MixFFN(dim, int(dim * 4))
        elif token_mlp == 'mix_skip':
            self.mlp = MixFFN_skip(dim, int(dim * 4))
        else:
            self.mlp = MLP_FFN(dim, int(dim * 4))

    def emit(self, eng: Engine, x: Act) -> Act:
        tx = self.attn.emit(eng, eng.layer_norm(x, self.norm1), residual=x)          # x + attn(norm1(x)): GEMM epilogue
        return self.mlp.emit(eng, eng.layer_norm(tx, self.norm2), residual=tx)       # tx + mlp(norm2(tx))


class MiT(nn.Module):
    """missformer.py:302-368"""

    def __init__(self, image_size, dims, layers, in_ch=3, token_mlp='mix_skip'):
        super().__init__()
        patch_sizes, strides, padding_sizes = [7, 3, 3, 3], [4, 2, 2, 2], [3, 1, 1, 1]
        reduction_ratios, heads = [8, 4, 2, 1], [1, 2, 5, 8]
        size, cin = image_size, in_ch
        for i in range(4):
            setattr(self, f"patch_embed{i + 1}", OverlapPatchEmbeddings(size, patch_sizes[i], strides[i], padding_sizes[i], cin, dims[i]))
            size, cin = size // strides[i], dims[i]
        for i in range(4):
            setattr(self, f"block{i + 1}", nn.ModuleList([TransformerBlock(dims[i], heads[i], reduction_ratios[i], token_mlp)
                                                         for _ in range(layers[i])]))
            setattr(self, f"norm{i + 1}", nn.LayerNorm(dims[i]))

    def emit(self, eng: Engine, x: torch.Tensor) -> List[Act]:
        outs, t = [], x
        for i in range(1, 5):
            t = getattr(self, f"patch_embed{i}").emit(eng, t)
            for blk in getattr(self, f"block{i}"):
                t = blk.emit(eng, t)
            t = eng.layer_norm(t, getattr(self, f"norm{i}"))
            outs.append(t)
        return outs


class PatchExpand(nn.Module):
    """missformer.py:512-537: Linear(dim, 4 dim) -> 2x2 rearrange -> LayerNorm(dim)"""

    def __init__(self, input_resolution, dim, dim_scale=2, norm_layer=nn.LayerNorm):
        super().__init__()
        if dim_scale != 2:
            raise NotImplementedError("PatchExpand is only used with dim_scale=2")
        self.input_resolution, self.dim, self.dim_scale = input_resolution, dim, dim_scale
        self.expand = nn.Linear(dim, dim * dim_scale ** 2, bias=False)
        self.norm = norm_layer(dim)
        self.output_dim = dim

    def emit(self, eng: Engine, x: Act, out: Optional[Act] = None) -> Act:
        assert (x.H, x.W) == tuple(self.input_resolution), "input feature has wrong size"
        return eng.layer_norm(eng.linear_expand2(x, self.expand), self.norm, out=out)


class FinalPatchExpand_X4(_SwinFinalExpand):
    """missformer.py:539-564 (the same layers as swin_unet_v2.py:364-387)"""


class SegU_decoder(nn.Module):
    """missformer.py:566-633"""

    def __init__(self, input_resolution, in_out_chan, heads, reduction_ratios, token_mlp_mode, n_class=9,
                 norm_layer=nn.LayerNorm, is_last=False):
        super().__init__()
        self.input_resolution = input_resolution
        dims, out_dim = in_out_chan
        self.concat_linear = nn.Linear(dims, out_dim)
        if not is_last:
            self.layer_up = PatchExpand(input_resolution=input_resolution, dim=out_dim, dim_scale=2, norm_layer=norm_layer)
            self.last_layer = None
        else:
            self.layer_up = FinalPatchExpand_X4(input_resolution=input_resolution, dim=out_dim, dim_scale=4, norm_layer=norm_layer)
            self.last_layer = nn.Conv2d(out_dim, n_class, 1)
        self.layer_former_1 = TransformerBlock(out_dim, heads, reduction_ratios, token_mlp=token_mlp_mode)
        self.layer_former_2 = TransformerBlock(out_dim, heads, reduction_ratios, token_mlp=token_mlp_mode)
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Conv2d)):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.LayerNorm):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def emit(self, eng: Engine, x: Act, cat: bool, out: Optional[Act] = None):
        """x: the (B, H, W, C) tokens, or with `cat` the buffer holding cat([x1, x2], -1) (:615)"""
        t = eng.linear(x, self.concat_linear) if cat else x
        t = self.layer_former_2.emit(eng, self.layer_former_1.emit(eng, t))
        if self.last_layer is not None:
            return self.layer_up.emit_head(eng, t, self.last_layer)
        return self.layer_up.emit(eng, t, out=out)


class BridgeLayer_4(nn.Module):
    """missformer.py:635-702"""

    def __init__(self, mi_t_dims, head, reduction_ratios, image_size):
        super().__init__()
        self.mi_t_dims = mi_t_dims
        self.common_bridge_dim = d = mi_t_dims[0]
        for i in range(4):
            setattr(self, f"proj_c{i + 1}", nn.Linear(mi_t_dims[i], d))
        self.patch_resolutions = [(image_size // s, image_size // s) for s in (4, 8, 16, 32)]
        self.norm1 = nn.LayerNorm(d)
        self.attn = M_EfficientSelfAtten(d, head, reduction_ratios, self.patch_resolutions, mi_t_dims)
        self.norm2 = nn.LayerNorm(d)
        for i in range(4):
            setattr(self, f"mixffn{i + 1}", MixFFN_skip(d, d * 4))

    def emit(self, eng: Engine, inputs, B: int) -> Act:
        shapes = [(B, h, w) for h, w in self.patch_resolutions]
        if isinstance(inputs, list):
            cat, slots = eng.new_rows(shapes, self.common_bridge_dim)
            for i, (c, slot) in enumerate(zip(inputs, slots)):
                if (c.N, c.H, c.W) != (slot.N, slot.H, slot.W):
                    raise ValueError(f"MISSFormer built for image_size {self.patch_resolutions[0][0] * 4} got a "
                                     f"{c.H}x{c.W} map at scale {i} (expected {slot.H}x{slot.W})")
                eng.linear(c, getattr(self, f"proj_c{i + 1}"), out=slot)
        else:
            cat = inputs
        tx1 = self.attn.emit(eng, eng.layer_norm(cat, self.norm1), B, residual=cat)
        tx = eng.layer_norm(tx1, self.norm2)
        ffn, slots = eng.new_rows(shapes, self.common_bridge_dim)
        for i, (v, slot) in enumerate(zip(eng.row_views(tx, shapes), slots)):
            getattr(self, f"mixffn{i + 1}").emit(eng, v, out=slot)
        return eng.add(tx1, ffn)


class BridegeBlock_4(nn.Module):
    """missformer.py:765-813"""

    def __init__(self, mi_t_dims, head, reduction_ratios, image_size):
        super().__init__()
        self.image_size, self.mi_t_dims = image_size, mi_t_dims
        self.common_bridge_dim = mi_t_dims[0]
        for i in range(4):
            setattr(self, f"bridge_layer{i + 1}", BridgeLayer_4(mi_t_dims, head, reduction_ratios, image_size))
        self.patch_resolutions = [(image_size // s, image_size // s) for s in (4, 8, 16, 32)]
        for i in range(4):
            setattr(self, f"proj_back_c{i + 1}", nn.Linear(self.common_bridge_dim, mi_t_dims[i]))

    def emit(self, eng: Engine, feats: List[Act], outs: Sequence[Optional[Act]]) -> List[Act]:
        """outs[i]: where skip i is to be written (the right half of a decoder's concat buffer) or None"""
        B = feats[0].N
        t = feats
        for i in range(4):
            t = getattr(self, f"bridge_layer{i + 1}").emit(eng, t, B)
        views = eng.row_views(t, [(B, h, w) for h, w in self.patch_resolutions])
        return [eng.linear(v, getattr(self, f"proj_back_c{i + 1}"), out=outs[i]) for i, v in enumerate(views)]


class MISSFormer(HipModule):
    """Same constructor as the reference (missformer.py:866-907)."""

    def __init__(self, num_classes: int = 1, in_channels: int = 3, token_mlp_mode: str = "mix_skip",
                 encoder_pretrained: bool = True, image_size: int = 512, **kwargs):
        super().__init__()
        dims, layers, _ = segformer_settings['B1']
        self.dims, self.image_size = dims, image_size
        self.backbone = MiT(image_size, dims, layers, in_channels, token_mlp_mode)
        reduction_ratios, heads = [8, 4, 2, 1], [1, 2, 5, 8]
        d = image_size // 32
        self.bridge = BridegeBlock_4(dims, heads[0], reduction_ratios, image_size)
        self.decoder_3 = SegU_decoder((d, d), [dims[3], dims[3]], heads[3], reduction_ratios[3], token_mlp_mode,
                                      n_class=num_classes, is_last=False)
        self.decoder_2 = SegU_decoder((d * 2, d * 2), [dims[3] + dims[2], dims[2]], heads[2], reduction_ratios[2],
                                      token_mlp_mode, n_class=num_classes, is_last=False)
        self.decoder_1 = SegU_decoder((d * 4, d * 4), [dims[2] + dims[1], dims[1]], heads[1], reduction_ratios[1],
                                      token_mlp_mode, n_class=num_classes, is_last=False)
        self.decoder_0 = SegU_decoder((d * 8, d * 8), [dims[1] + dims[0], dims[0]], heads[0], reduction_ratios[0],
                                      token_mlp_mode, n_class=num_classes, is_last=True)
        if not encoder_pretrained:
            self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, (nn.Linear, nn.Conv2d)):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.LayerNorm):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)

    def emit(self, eng: Engine, x: torch.Tensor):
        if x.shape[1] == 1:
            x = x.repeat(1, 3, 1, 1)
        N, _, H, W = x.shape
        S = self.image_size
        if (H, W) != (S, S) or S % 32:
            raise ValueError(f"MISSFormer(image_size={S}) takes {S}x{S} inputs with image_size divisible by 32 "
                             f"(the bridge and the decoders are sized by the constructor), got {H}x{W}")
        dims, d = self.dims, S // 32
        feats = self.backbone.emit(eng, x)
        # decoder k reads cat([expanded deeper tokens, bridge skip k], -1): one buffer, both producers write in place
        cats = [eng.new_cat(N, d * 8 >> k, d * 8 >> k, (dims[k + 1], dims[k])) for k in range(3)]
        skips = self.bridge.emit(eng, feats, [cats[0][1][1], cats[1][1][1], cats[2][1][1], None])
        self.decoder_3.emit(eng, skips[3], cat=False, out=cats[2][1][0])
        self.decoder_2.emit(eng, cats[2][0], cat=True, out=cats[1][1][0])
        self.decoder_1.emit(eng, cats[1][0], cat=True, out=cats[0][1][0])
        return (self.decoder_0.emit(eng, cats[0][0], cat=True),)
