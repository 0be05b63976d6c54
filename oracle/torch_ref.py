"""CPU fp32 oracle: a functional restatement of the reference's UNet hot path.

TEST INFRASTRUCTURE — see oracle/__init__.py.  The reference delegates all arithmetic to
``torch.nn`` modules; this file restates the same graph with ``torch.nn.functional`` calls over a
plain ``state_dict`` (reference key names), so it runs anywhere torch runs (the GPU box has no
/root/reference).  It is pinned against golden vectors produced by importing the reference's own
model files (oracle/gen_golden.py -> tests/golden/), see tests/test_oracle_golden.py.

Every function cites the reference lines it follows (paths relative to /root/reference).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

State = Dict[str, torch.Tensor]

# Optional storage-rounding model.  None (default): the reference's fp32 arithmetic.  torch.bfloat16:
# every tensor the HIP engine STORES in bf16 (image, conv weights, conv outputs, activations, resized
# maps) is rounded to bf16 at the same point, arithmetic in between stays fp32 — this separates "the
# product rounds to bf16 here" from "a kernel computes the wrong thing" on networks whose low-sample
# BatchNorms amplify rounding (u2net: see DESIGN.md).  Only conv_bn_relu/upsample_like honour it.
_STORE_DTYPE = None
# Optional jitter in front of every storage rounding: t * (1 + jitter * N(0, 1)).  With jitter of fp32-rounding size
# (1e-6) the oracle becomes "another correct implementation" -- one whose fp32 sums come out in a different order, so that
# the few values within 1e-6 of a bf16 tie round the other way -- and oracle-vs-jittered-oracle measures how far two
# correct bf16-storage implementations of a network end up apart (the yardstick for engine-vs-oracle bounds).
_STORE_JITTER = 0.0
_JITTER_GEN = None


def set_storage_rounding(dtype, jitter: float = 0.0, seed: int = 0) -> None:
    global _STORE_DTYPE, _STORE_JITTER, _JITTER_GEN
    _STORE_DTYPE = dtype
    _STORE_JITTER = float(jitter) if dtype is not None else 0.0
    _JITTER_GEN = torch.Generator().manual_seed(seed) if _STORE_JITTER > 0.0 else None


def _q(t: torch.Tensor) -> torch.Tensor:
    if _STORE_DTYPE is None:
        return t
    if _STORE_JITTER > 0.0:
        t = t * (1.0 + _STORE_JITTER * torch.randn(t.shape, generator=_JITTER_GEN))
    return t.to(_STORE_DTYPE).float()



# ---------------------------------------------------------------------------------------------
# primitives
# ---------------------------------------------------------------------------------------------
def conv_bn_relu(x: torch.Tensor, sd: State, conv: str, bn: str, training: bool,
                 dilation: int = 1, residual: torch.Tensor = None) -> torch.Tensor:
    """Conv2d(k3, padding=dilation) -> BatchNorm2d -> ReLU.
    unet_zoo/models/common_layers.py:28-30 (and :31-33, :47-56; u2net.py:10-17 with dilation).
    Train mode: batch mean / biased variance, running stats updated with momentum 0.1 and the
    unbiased variance (torch.nn.BatchNorm2d defaults, eps 1e-5)."""
    x = _q(F.conv2d(_q(x), _q(sd[conv + ".weight"]), sd.get(conv + ".bias"), padding=dilation, dilation=dilation))
    x = F.batch_norm(x, sd[bn + ".running_mean"], sd[bn + ".running_var"], sd[bn + ".weight"],
                     sd[bn + ".bias"], training=training, momentum=0.1, eps=1e-5)
    if training and (bn + ".num_batches_tracked") in sd:
        sd[bn + ".num_batches_tracked"] += 1
    x = F.relu(x)
    if residual is not None:  # RSU tail `hx1d + hxin` (u2net.py:74)
        x = x + residual
    return _q(x)


def double_conv(x: torch.Tensor, sd: State, prefix: str, training: bool) -> torch.Tensor:
    """DoubleConv.forward — unet_zoo/models/common_layers.py:20-37 (``prefix`` ends in conv_op)."""
    x = conv_bn_relu(x, sd, f"{prefix}.0", f"{prefix}.1", training)
    return conv_bn_relu(x, sd, f"{prefix}.3", f"{prefix}.4", training)


def down_sample(x, sd: State, prefix: str, training: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """DownSample.forward — common_layers.py:92-95: (DoubleConv output, MaxPool2d(2,2) of it)."""
    down = double_conv(x, sd, f"{prefix}.conv.conv_op", training)
    return down, F.max_pool2d(down, kernel_size=2, stride=2)


def up_sample_unet(x1, x2, sd: State, prefix: str, training: bool) -> torch.Tensor:
    """UpSample_UNet.forward — common_layers.py:107-116: ConvTranspose2d(k2,s2), zero-pad to the
    skip's size, cat([up, skip], dim=1), DoubleConv."""
    x1 = _q(F.conv_transpose2d(_q(x1), _q(sd[f"{prefix}.up.weight"]), sd[f"{prefix}.up.bias"], stride=2))
    dy, dx = x2.shape[2] - x1.shape[2], x2.shape[3] - x1.shape[3]
    x1 = F.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    return double_conv(torch.cat([x1, x2], 1), sd, f"{prefix}.conv.conv_op", training)


# ---------------------------------------------------------------------------------------------
# UNet
# ---------------------------------------------------------------------------------------------
def unet_forward(sd: State, x: torch.Tensor, training: bool) -> torch.Tensor:
    """UNet.forward — unet_zoo/models/unet.py:29-43."""
    d1, p1 = down_sample(x, sd, "down_convolution_1", training)
    d2, p2 = down_sample(p1, sd, "down_convolution_2", training)
    d3, p3 = down_sample(p2, sd, "down_convolution_3", training)
    d4, p4 = down_sample(p3, sd, "down_convolution_4", training)
    b = double_conv(p4, sd, "bottle_neck.conv_op", training)
    u1 = up_sample_unet(b, d4, sd, "up_convolution_1", training)
    u2 = up_sample_unet(u1, d3, sd, "up_convolution_2", training)
    u3 = up_sample_unet(u2, d2, sd, "up_convolution_3", training)
    u4 = up_sample_unet(u3, d1, sd, "up_convolution_4", training)
    return F.conv2d(u4, sd["out.conv.weight"], sd["out.conv.bias"])  # OutConv, common_layers.py:125


# ---------------------------------------------------------------------------------------------
# Attention U-Net
# ---------------------------------------------------------------------------------------------
def conv_block(x, sd: State, prefix: str, training: bool) -> torch.Tensor:
    """ConvBlock.forward — common_layers.py:39-61 (``prefix`` ends in .conv)."""
    x = conv_bn_relu(x, sd, f"{prefix}.0", f"{prefix}.1", training)
    return conv_bn_relu(x, sd, f"{prefix}.3", f"{prefix}.4", training)


def up_conv_block(x, sd: State, prefix: str, training: bool) -> torch.Tensor:
    """UpConvBlock.forward — common_layers.py:63-80: nn.Upsample(scale_factor=2) (nearest), Conv3x3, BN, ReLU."""
    x = F.interpolate(x, scale_factor=2, mode="nearest")
    return conv_bn_relu(x, sd, f"{prefix}.up.1", f"{prefix}.up.2", training)


def _conv1x1_bn(x, sd: State, prefix: str, training: bool) -> torch.Tensor:
    x = F.conv2d(_q(x), _q(sd[f"{prefix}.0.weight"]), sd[f"{prefix}.0.bias"])
    if x.shape[1] > 1:
        x = _q(x)                 # the engine stores the raw 1x1 outputs in the run dtype, the 1-channel psi map in fp32
    x = F.batch_norm(x, sd[f"{prefix}.1.running_mean"], sd[f"{prefix}.1.running_var"], sd[f"{prefix}.1.weight"],
                     sd[f"{prefix}.1.bias"], training=training, momentum=0.1, eps=1e-5)
    if training and f"{prefix}.1.num_batches_tracked" in sd:
        sd[f"{prefix}.1.num_batches_tracked"] += 1
    return x


def attention_block(g, x, sd: State, prefix: str, training: bool) -> torch.Tensor:
    """AttentionBlock.forward — attention_unet.py:34-40."""
    g1 = _conv1x1_bn(g, sd, f"{prefix}.w_g", training)
    x1 = _conv1x1_bn(x, sd, f"{prefix}.w_x", training)
    r = F.relu(g1 + x1)
    keep, _STORE = _STORE_DTYPE, None
    set_storage_rounding(None)    # relu(g1 + x1) is never stored: the psi convolution reads it in fp32
    try:
        pre = _conv1x1_bn(r, sd, f"{prefix}.psi", training)
    finally:
        set_storage_rounding(keep)
    return _q(torch.sigmoid(pre) * x)


def attention_unet_forward(sd: State, x: torch.Tensor, training: bool) -> torch.Tensor:
    """AttentionUNet.forward — attention_unet.py:73-110."""
    pool = lambda t: F.max_pool2d(t, kernel_size=2, stride=2)  # noqa: E731  (shared self.maxpool)
    x1 = conv_block(x, sd, "conv1.conv", training)
    x2 = conv_block(pool(x1), sd, "conv2.conv", training)
    x3 = conv_block(pool(x2), sd, "conv3.conv", training)
    x4 = conv_block(pool(x3), sd, "conv4.conv", training)
    x5 = conv_block(pool(x4), sd, "conv5.conv", training)
    d = x5
    for lvl, skip in ((5, x4), (4, x3), (3, x2), (2, x1)):
        d = up_conv_block(d, sd, f"up{lvl}", training)
        gated = attention_block(d, skip, sd, f"att{lvl}", training)
        d = conv_block(torch.cat((gated, d), dim=1), sd, f"upconv{lvl}.conv", training)
    return F.conv2d(d, sd["conv_1x1.weight"], sd["conv_1x1.bias"])


# ---------------------------------------------------------------------------------------------
# U^2-Net (unet_zoo/models/u2net.py)
# ---------------------------------------------------------------------------------------------
def rebnconv(x, sd: State, prefix: str, training: bool, dirate: int = 1, residual=None) -> torch.Tensor:
    """REBNCONV.forward — u2net.py:6-17: Conv3x3(dilation=d, padding=d) -> BN -> ReLU."""
    return conv_bn_relu(x, sd, f"{prefix}.conv_s1", f"{prefix}.bn_s1", training, dilation=dirate,
                        residual=residual)


def upsample_like(src: torch.Tensor, tar: torch.Tensor) -> torch.Tensor:
    """_upsample_like — u2net.py:19-22."""
    return _q(F.interpolate(src, size=tar.shape[2:], mode="bilinear", align_corners=False))


def rsu(x, sd: State, prefix: str, training: bool, depth: int) -> torch.Tensor:
    """RSU7/6/5/4.forward — u2net.py:48-74, 97-119, 139-157, 174-188 (depth = 7, 6, 5, 4):
    in-conv, `depth-1` encoder convs with ceil-mode 2x2 pools between them, one dilation-2 conv at the
    bottom, decoder convs on cat((upsampled, skip), 1), residual with the in-conv output."""
    hxin = rebnconv(x, sd, f"{prefix}.rebnconvin", training)
    skips = []
    hx = hxin
    for i in range(1, depth):
        hx = rebnconv(hx, sd, f"{prefix}.rebnconv{i}", training)
        skips.append(hx)
        if i < depth - 1:
            hx = F.max_pool2d(hx, 2, stride=2, ceil_mode=True)
    hx = rebnconv(skips[-1], sd, f"{prefix}.rebnconv{depth}", training, dirate=2)
    for i in range(depth - 1, 0, -1):
        skip = skips[i - 1]
        if hx.shape[2:] != skip.shape[2:]:
            hx = upsample_like(hx, skip)
        hx = rebnconv(torch.cat((hx, skip), 1), sd, f"{prefix}.rebnconv{i}d", training,
                      residual=hxin if i == 1 else None)   # `hx1d + hxin`
    return hx


def rsu4f(x, sd: State, prefix: str, training: bool) -> torch.Tensor:
    """RSU4F.forward — u2net.py:203-213: dilations 1,2,4,8 down, 4,2,1 up, no resampling."""
    hxin = rebnconv(x, sd, f"{prefix}.rebnconvin", training)
    hx1 = rebnconv(hxin, sd, f"{prefix}.rebnconv1", training, 1)
    hx2 = rebnconv(hx1, sd, f"{prefix}.rebnconv2", training, 2)
    hx3 = rebnconv(hx2, sd, f"{prefix}.rebnconv3", training, 4)
    hx4 = rebnconv(hx3, sd, f"{prefix}.rebnconv4", training, 8)
    hx3d = rebnconv(torch.cat((hx4, hx3), 1), sd, f"{prefix}.rebnconv3d", training, 4)
    hx2d = rebnconv(torch.cat((hx3d, hx2), 1), sd, f"{prefix}.rebnconv2d", training, 2)
    return rebnconv(torch.cat((hx2d, hx1), 1), sd, f"{prefix}.rebnconv1d", training, 1, residual=hxin)  # hx1d + hxin


_U2NET_STAGES = (("stage1", 7), ("stage2", 6), ("stage3", 5), ("stage4", 4), ("stage5", 0), ("stage6", 0))


def _u2_stage(x, sd, name, depth, training):
    return rsu4f(x, sd, name, training) if depth == 0 else rsu(x, sd, name, training, depth)


def u2net_forward(sd: State, x: torch.Tensor, training: bool) -> Dict[str, torch.Tensor]:
    """U2NET.forward — u2net.py:246-298.  Returns the reference's dict of seven logit maps."""
    enc = []
    hx = x
    for i, (name, depth) in enumerate(_U2NET_STAGES):
        h = _u2_stage(hx, sd, name, depth, training)
        enc.append(h)
        if i < 5:
            hx = F.max_pool2d(h, 2, stride=2, ceil_mode=True)
    dec = {6: enc[5]}
    cur = enc[5]
    for lvl, depth in ((5, 0), (4, 4), (3, 5), (2, 6), (1, 7)):
        up = upsample_like(cur, enc[lvl - 1])
        cur = _u2_stage(torch.cat((up, enc[lvl - 1]), 1), sd, f"stage{lvl}d", depth, training)
        dec[lvl] = cur
    sides = []
    for k in range(1, 7):
        d = F.conv2d(dec[k], sd[f"side{k}.weight"], sd[f"side{k}.bias"], padding=1)
        if k > 1:  # fp32 logit planes: never rounded
            d = F.interpolate(d, size=sides[0].shape[2:], mode="bilinear", align_corners=False)
        sides.append(d)
    d0 = F.conv2d(torch.cat(sides, 1), sd["outconv.weight"], sd["outconv.bias"])
    out = {"main": d0}
    for k in range(6):
        out[f"side{k + 1}"] = sides[k]
    return out


def model_loss(outputs, mask: torch.Tensor) -> torch.Tensor:
    """BCEWithLogits on a tensor output; on U2Net's dict the unit-weighted sum over the seven heads
    (unet_zoo/utils/training_loop.py:24-32, 60-64; every head is at the mask's resolution)."""
    if isinstance(outputs, (dict, list, tuple)):   # u2net's seven heads; nested_unet's four under deep supervision
        total = None
        for v in (outputs.values() if isinstance(outputs, dict) else outputs):
            l = F.binary_cross_entropy_with_logits(v, mask)
            total = l if total is None else total + l
        return total
    return F.binary_cross_entropy_with_logits(outputs, mask)


# ---------------------------------------------------------------------------------------------
# Swin-UNet V2 (unet_zoo/models/swin_unet_v2.py)
# ---------------------------------------------------------------------------------------------
def swin_config(sd: State, img_size: int, window_size: int = 7, drop_path_rate: float = 0.1,
                depths=(2, 2, 2, 2), num_heads=(3, 6, 12, 24), patch_size: int = 4) -> dict:
    """Static configuration of SwinTransformerSys (swin_unet_v2.py:590-667); embed_dim from the weights."""
    return {"img": img_size, "ws": window_size, "dpr": drop_path_rate, "depths": tuple(depths),
            "heads": tuple(num_heads), "patch": patch_size, "embed": sd["patch_embed.proj.weight"].shape[0]}


def _layer_norm(x, sd: State, prefix: str):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], 1e-5)


def swin_attention_mask(H: int, W: int, ws: int, shift: int) -> torch.Tensor:
    """SwinTransformerBlock.__init__ — swin_unet_v2.py:214-236: (nW, N, N) of 0 / -100."""
    img = torch.zeros(1, H, W, 1)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, hs, wsl, :] = cnt
            cnt += 1
    mw = img.view(1, H // ws, ws, W // ws, ws, 1).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return am.masked_fill(am != 0, -100.0).masked_fill(am == 0, 0.0)


def swin_window_attention(xw, sd: State, prefix: str, heads: int, mask, qk_scale=None):
    """WindowAttention.forward — swin_unet_v2.py:127-159: cosine attention with a learned per-entry
    temperature tau (clipped at 0.01), continuous position bias from the log-spaced offsets through
    cpb = Linear(2,256)-ReLU-Linear(256,heads), optional shift mask, softmax, @v, proj."""
    B_, N, C = xw.shape
    d = C // heads
    qkv = F.linear(xw, sd[prefix + ".qkv.weight"], sd.get(prefix + ".qkv.bias"))
    qkv = qkv.reshape(B_, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (qk_scale or d ** -0.5), qkv[1], qkv[2]       # self.scale = qk_scale or head_dim ** -0.5 (:94)
    attn = torch.einsum("bhqd,bhkd->bhqk", q, k) / torch.maximum(
        q.norm(dim=-1, keepdim=True) * k.norm(dim=-1, keepdim=True).transpose(-2, -1),
        torch.tensor(1e-6, dtype=q.dtype))
    attn = attn / torch.clip(sd[prefix + ".tau"].unsqueeze(0)[:, :, :N, :N], min=0.01)
    idx = sd[prefix + ".log_relative_position_index"][:N, :N]
    bias = F.linear(F.relu(F.linear(idx, sd[prefix + ".cpb.fc1.weight"], sd[prefix + ".cpb.fc1.bias"])),
                    sd[prefix + ".cpb.fc2.weight"], sd[prefix + ".cpb.fc2.bias"])
    attn = attn + bias.permute(2, 0, 1).unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = (attn.view(B_ // nW, nW, heads, N, N) + mask.unsqueeze(1).unsqueeze(0)).view(-1, heads, N, N)
    attn = attn.softmax(dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(B_, N, C)
    return F.linear(x, sd[prefix + ".proj.weight"], sd[prefix + ".proj.bias"])


def swin_block(x, sd: State, prefix: str, H: int, W: int, heads: int, ws: int, shift: int,
               drop_scale=None, qk_scale=None):
    """SwinTransformerBlock.forward — swin_unet_v2.py:240-269.  NOTE: the reference returns after
    `shortcut + drop_path(norm1(attention))`; its mlp / norm2 members are never called.
    drop_scale: per-sample (B,) factor of the stochastic-depth branch (None = identity)."""
    if min(H, W) <= ws:                       # :199-202
        shift, ws = 0, min(H, W)
    B, L, C = x.shape
    xs = x.view(B, H, W, C)
    if shift > 0:
        xs = torch.roll(xs, shifts=(-shift, -shift), dims=(1, 2))
    xw = xs.view(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)
    mask = swin_attention_mask(H, W, ws, shift) if shift > 0 else None
    aw = swin_window_attention(xw, sd, prefix + ".attn", heads, mask, qk_scale)
    xs = aw.view(B, H // ws, W // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, C)
    if shift > 0:
        xs = torch.roll(xs, shifts=(shift, shift), dims=(1, 2))
    y = _layer_norm(xs.reshape(B, L, C), sd, prefix + ".norm1")
    if drop_scale is not None:
        y = y * drop_scale.view(B, 1, 1)
    return x + y


def swin_patch_merging(x, sd: State, prefix: str, H: int, W: int):
    """PatchMerging.forward — swin_unet_v2.py:315-332."""
    B, L, C = x.shape
    x = x.view(B, H, W, C)
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1).view(B, -1, 4 * C)
    return F.linear(_layer_norm(x, sd, prefix + ".norm"), sd[prefix + ".reduction.weight"])


def swin_patch_expand(x, sd: State, prefix: str, H: int, W: int, r: int):
    """PatchExpand.forward (r=2, :352-362) / FinalPatchExpand_X4.forward (r=4, :375-387):
    Linear, 'b h w (p1 p2 c) -> b (h p1) (w p2) c', LayerNorm."""
    x = F.linear(x, sd[prefix + ".expand.weight"])
    B, L, C = x.shape
    c = C // (r * r)
    x = x.view(B, H, W, r, r, c).permute(0, 1, 3, 2, 4, 5).reshape(B, H * r * W * r, c)
    return _layer_norm(x, sd, prefix + ".norm")


def swin_unet_v2_forward(sd: State, x: torch.Tensor, training: bool, cfg: dict = None,
                         drop_scales: Dict[str, torch.Tensor] = None) -> torch.Tensor:
    """SwinTransformerSys.forward — swin_unet_v2.py:711-758.  Stochastic depth (train mode,
    drop_path_rate > 0) is driven by `drop_scales[block prefix]` = Bernoulli(keep)/keep per sample; with
    drop_scales None it is the identity (eval mode, or drop_path_rate 0)."""
    if cfg is None:
        cfg = swin_config(sd, x.shape[-1])
    E, ws, depths, heads, ps = cfg["embed"], cfg["ws"], cfg["depths"], cfg["heads"], cfg["patch"]
    nl = len(depths)
    B = x.shape[0]
    R = cfg["img"] // ps
    ds = drop_scales or {}
    # PatchEmbed (:548-556): Conv2d(k=s=patch) -> tokens -> LayerNorm
    t = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=ps).flatten(2).transpose(1, 2)
    if "patch_embed.norm.weight" in sd:                       # patch_norm=True (:555-556)
        t = _layer_norm(t, sd, "patch_embed.norm")
    if "absolute_pos_embed" in sd:                            # ape=True (:713-715); pos_drop: rate 0 here
        t = t + sd["absolute_pos_embed"]
    skips = []
    for i in range(nl):                                       # forward_features (:711-723)
        skips.append(t)
        r = R >> i
        for b in range(depths[i]):
            pre = f"layers.{i}.blocks.{b}"
            t = swin_block(t, sd, pre, r, r, heads[i], ws, 0 if b % 2 == 0 else ws // 2, ds.get(pre), cfg.get("qk_scale"))
        if i < nl - 1:
            t = swin_patch_merging(t, sd, f"layers.{i}.downsample", r, r)
    t = _layer_norm(t, sd, "norm")
    for inx in range(nl):                                     # forward_up_features (:725-741)
        lvl = nl - 1 - inx
        r = R >> lvl
        if inx == 0:
            t = swin_patch_expand(t, sd, "layers_up.0", r, r, 2)
            continue
        t = torch.cat([t, skips[lvl]], -1)
        t = F.linear(t, sd[f"concat_back_dim.{inx}.weight"], sd[f"concat_back_dim.{inx}.bias"])
        for b in range(depths[lvl]):
            pre = f"layers_up.{inx}.blocks.{b}"
            t = swin_block(t, sd, pre, r, r, heads[lvl], ws, 0 if b % 2 == 0 else ws // 2, ds.get(pre), cfg.get("qk_scale"))
        if inx < nl - 1:
            t = swin_patch_expand(t, sd, f"layers_up.{inx}.upsample", r, r, 2)
    t = _layer_norm(t, sd, "norm_up")
    t = swin_patch_expand(t, sd, "up", R, R, 4)               # up_x4 (:743-754)
    t = t.view(B, 4 * R, 4 * R, E).permute(0, 3, 1, 2)
    return F.conv2d(t, sd["output.weight"])


# ---------------------------------------------------------------------------------------------
# UNet++ (unet_zoo/models/nested_unet.py)
# ---------------------------------------------------------------------------------------------
def vgg_block(x, sd: State, prefix: str, training: bool) -> torch.Tensor:
    """VGGBlock.forward — nested_unet.py:13-22."""
    x = conv_bn_relu(x, sd, f"{prefix}.conv1", f"{prefix}.bn1", training)
    return conv_bn_relu(x, sd, f"{prefix}.conv2", f"{prefix}.bn2", training)


def nested_unet_forward(sd: State, x: torch.Tensor, training: bool):
    """NestedUNet.forward — nested_unet.py:72-105: x_{i,j} = VGG(cat[x_{i,0..j-1}, up(x_{i+1,j-1})]) with
    up = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) (:32) and pool = MaxPool2d(2, 2) (:31);
    four heads under deep supervision (present iff the state has `final1`), else `final` on x_{0,4}."""
    def up(t):
        return _q(F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=True))

    def pool(t):
        return F.max_pool2d(t, 2, 2)

    n = {}
    n[0, 0] = vgg_block(x, sd, "conv0_0", training)
    n[1, 0] = vgg_block(pool(n[0, 0]), sd, "conv1_0", training)
    n[0, 1] = vgg_block(torch.cat([n[0, 0], up(n[1, 0])], 1), sd, "conv0_1", training)
    n[2, 0] = vgg_block(pool(n[1, 0]), sd, "conv2_0", training)
    n[1, 1] = vgg_block(torch.cat([n[1, 0], up(n[2, 0])], 1), sd, "conv1_1", training)
    n[0, 2] = vgg_block(torch.cat([n[0, 0], n[0, 1], up(n[1, 1])], 1), sd, "conv0_2", training)
    n[3, 0] = vgg_block(pool(n[2, 0]), sd, "conv3_0", training)
    n[2, 1] = vgg_block(torch.cat([n[2, 0], up(n[3, 0])], 1), sd, "conv2_1", training)
    n[1, 2] = vgg_block(torch.cat([n[1, 0], n[1, 1], up(n[2, 1])], 1), sd, "conv1_2", training)
    n[0, 3] = vgg_block(torch.cat([n[0, 0], n[0, 1], n[0, 2], up(n[1, 2])], 1), sd, "conv0_3", training)
    n[4, 0] = vgg_block(pool(n[3, 0]), sd, "conv4_0", training)
    n[3, 1] = vgg_block(torch.cat([n[3, 0], up(n[4, 0])], 1), sd, "conv3_1", training)
    n[2, 2] = vgg_block(torch.cat([n[2, 0], n[2, 1], up(n[3, 1])], 1), sd, "conv2_2", training)
    n[1, 3] = vgg_block(torch.cat([n[1, 0], n[1, 1], n[1, 2], up(n[2, 2])], 1), sd, "conv1_3", training)
    n[0, 4] = vgg_block(torch.cat([n[0, 0], n[0, 1], n[0, 2], n[0, 3], up(n[1, 3])], 1), sd, "conv0_4", training)
    if "final1.weight" in sd:
        return [F.conv2d(n[0, j], sd[f"final{j}.weight"], sd[f"final{j}.bias"]) for j in (1, 2, 3, 4)]
    return F.conv2d(n[0, 4], sd["final.weight"], sd["final.bias"])


# ---------------------------------------------------------------------------------------------
# ResUnet (unet_zoo/models/resunet.py, ResidualConv / UpsampleResUnet of common_layers.py:182-207)
# ---------------------------------------------------------------------------------------------
def _bn(x, sd: State, prefix: str, training: bool) -> torch.Tensor:
    y = F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"],
                     sd[prefix + ".bias"], training=training, momentum=0.1, eps=1e-5)
    if training and (prefix + ".num_batches_tracked") in sd:
        sd[prefix + ".num_batches_tracked"] += 1
    return y


def residual_conv(x, sd: State, prefix: str, training: bool, stride: int) -> torch.Tensor:
    """ResidualConv.forward — common_layers.py:185-199."""
    cb, cs = prefix + ".conv_block", prefix + ".conv_skip"
    h = _q(F.relu(_bn(x, sd, cb + ".0", training)))
    h = _q(F.conv2d(h, _q(sd[cb + ".2.weight"]), None, stride=stride, padding=1))
    h = _q(F.relu(_bn(h, sd, cb + ".3", training)))
    h = _q(F.conv2d(h, _q(sd[cb + ".5.weight"]), None, padding=1))
    s = _q(F.conv2d(_q(x), _q(sd[cs + ".0.weight"]), None, stride=stride))
    s = _q(_bn(s, sd, cs + ".1", training))
    return _q(h + s)


def resunet_forward(sd: State, x: torch.Tensor, training: bool) -> torch.Tensor:
    """ResUnet.forward — resunet.py:54-78."""
    h = conv_bn_relu(x, sd, "input_layer.0", "input_layer.1", training)
    h = _q(F.conv2d(h, _q(sd["input_layer.3.weight"]), sd["input_layer.3.bias"], padding=1))
    sk = _q(F.conv2d(_q(x), _q(sd["input_skip.0.weight"]), sd["input_skip.0.bias"], padding=1))
    x1 = _q(h + sk)
    x2 = residual_conv(x1, sd, "residual_conv_1", training, 2)
    x3 = residual_conv(x2, sd, "residual_conv_2", training, 2)
    x4 = residual_conv(x3, sd, "bridge", training, 2)

    def up(t, name):
        return _q(F.conv_transpose2d(_q(t), _q(sd[name + ".upsample.weight"]), sd[name + ".upsample.bias"], stride=2))

    x6 = residual_conv(torch.cat([up(x4, "upsample_1"), x3], 1), sd, "up_residual_conv1", training, 1)
    x8 = residual_conv(torch.cat([up(x6, "upsample_2"), x2], 1), sd, "up_residual_conv2", training, 1)
    x10 = residual_conv(torch.cat([up(x8, "upsample_3"), x1], 1), sd, "up_residual_conv3", training, 1)
    return F.conv2d(x10, sd["output_layer.0.weight"], sd["output_layer.0.bias"])


# ---------------------------------------------------------------------------------------------
# TransAttUNet (unet_zoo/models/transatt_unet.py over DoubleConvo / Down / Up of common_layers.py:130-180)
# ---------------------------------------------------------------------------------------------
def double_convo(x, sd: State, prefix: str, training: bool) -> torch.Tensor:
    """DoubleConvo.forward — common_layers.py:145-146 (``prefix`` ends in double_conv)."""
    x = conv_bn_relu(x, sd, f"{prefix}.0", f"{prefix}.1", training)
    return conv_bn_relu(x, sd, f"{prefix}.3", f"{prefix}.4", training)


def up_bilinear_cat(x1, x2, sd: State, prefix: str, training: bool) -> torch.Tensor:
    """Up.forward (bilinear) — common_layers.py:171-180: x2 up, centred zero pad, cat([x2, x1]), DoubleConvo."""
    x1 = _q(F.interpolate(x1, scale_factor=2, mode="bilinear", align_corners=True))
    dy, dx = x2.shape[2] - x1.shape[2], x2.shape[3] - x1.shape[3]
    x1 = F.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    return double_convo(torch.cat([x2, x1], 1), sd, f"{prefix}.conv.double_conv", training)


def transatt_unet_forward(sd: State, x: torch.Tensor, training: bool, attn_dropout: float = 0.0) -> torch.Tensor:
    """TransAttUNet.forward — transatt_unet.py:135-164 (bilinear=True).  `attn_dropout`: rate of the train-mode
    dropout on the channel-attention matrix (reference default 0.1, transatt_unet.py:86-88; the goldens use 0)."""
    def down(t, name):
        return double_convo(F.max_pool2d(t, 2), sd, f"{name}.maxpool_conv.1.double_conv", training)

    x1 = double_convo(x, sd, "inc.double_conv", training)
    x2 = down(x1, "down1")
    x3 = down(x2, "down2")
    x4 = down(x3, "down3")
    x5 = down(x4, "down4")
    B, C, h, w = x5.shape
    # PositionEmbeddingLearned.forward — :66-82
    x_emb, y_emb = sd["pos.col_embed.weight"][:w], sd["pos.row_embed.weight"][:h]
    pos = torch.cat([x_emb.unsqueeze(0).repeat(h, 1, 1), y_emb.unsqueeze(1).repeat(1, w, 1)], dim=-1)
    x5 = _q(x5 + pos.permute(2, 0, 1).unsqueeze(0))
    # PAM_Module.forward — :39-53
    q = _q(F.conv2d(x5, _q(sd["pam.query_conv.weight"]), sd["pam.query_conv.bias"])).view(B, -1, h * w).permute(0, 2, 1)
    k = _q(F.conv2d(x5, _q(sd["pam.key_conv.weight"]), sd["pam.key_conv.bias"])).view(B, -1, h * w)
    v = _q(F.conv2d(x5, _q(sd["pam.value_conv.weight"]), sd["pam.value_conv.bias"])).view(B, -1, h * w)
    attention = torch.softmax(torch.bmm(q, k), dim=-1)
    pam = _q(torch.bmm(v, attention.permute(0, 2, 1)).view(B, C, h, w))
    pam = _q(sd["pam.gamma"] * pam + x5)
    # ScaledDotProductAttention.forward — :91-107, temperature = sqrt(512)
    qq = x5.view(B, C, -1)
    attn = torch.matmul(qq / (512 ** 0.5), qq.permute(0, 2, 1))
    attn = F.dropout(F.softmax(attn, dim=-1), attn_dropout, training)
    sdpa = _q(torch.matmul(attn, qq).view(B, C, h, w))
    f = _q(sdpa + pam)
    u = up_bilinear_cat(f, x4, sd, "up1", training)
    u = up_bilinear_cat(u, x3, sd, "up2", training)
    u = up_bilinear_cat(u, x2, sd, "up3", training)
    u = up_bilinear_cat(u, x1, sd, "up4", training)
    return F.conv2d(u, sd["outc.conv.weight"], sd["outc.conv.bias"])


# ---------------------------------------------------------------------------------------------
# U-Transformer (unet_zoo/models/unet_transformer.py)
# ---------------------------------------------------------------------------------------------
def _pos_enc_2d(inv_freq: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """PositionalEncodingPermute2D(x) for x (b, ch, H, W) — unet_transformer.py:83-116: channels [0, n) from the row
    index, [n, 2n) from the column index, n = ceil(ch / 2), truncated to ch."""
    b, ch, H, W = x.shape
    n = 2 * inv_freq.numel()
    sx = torch.einsum("i,j->ij", torch.arange(H, dtype=torch.float32), inv_freq)
    sy = torch.einsum("i,j->ij", torch.arange(W, dtype=torch.float32), inv_freq)
    emb = torch.zeros(H, W, 2 * n)
    emb[:, :, :n] = torch.cat((sx.sin(), sx.cos()), -1).unsqueeze(1)
    emb[:, :, n:2 * n] = torch.cat((sy.sin(), sy.cos()), -1)
    return emb[None, :, :, :ch].repeat(b, 1, 1, 1).permute(0, 3, 1, 2)


def _conv1x1_bn_relu(x, sd: State, conv: str, bn: str, training: bool) -> torch.Tensor:
    y = _q(F.conv2d(_q(x), _q(sd[conv + ".weight"]), sd.get(conv + ".bias")))
    return _q(F.relu(_bn(y, sd, bn, training)))


def _dense_attention(Qs, Ks, Vs, sd: State, prefix: str) -> torch.Tensor:
    """MultiHeadDense q / k / v, softmax over dim=1 (the QUERY axis), A V — unet_transformer.py:127-137, :208-219"""
    b, c, h, w = Qs.shape
    Q = torch.bmm(Qs.flatten(2).permute(0, 2, 1), sd[prefix + ".query.weight"].repeat(b, 1, 1))
    K = torch.bmm(Ks.flatten(2).permute(0, 2, 1), sd[prefix + ".key.weight"].repeat(b, 1, 1))
    V = torch.bmm(Vs.flatten(2).permute(0, 2, 1), sd[prefix + ".value.weight"].repeat(b, 1, 1))
    A = torch.softmax(torch.bmm(Q, K.permute(0, 2, 1)) / (c ** 0.5), dim=1)
    return torch.bmm(A, V).permute(0, 2, 1).reshape(b, c, h, w)


def transformer_up(Y, S, sd: State, prefix: str, training: bool, res=(64, 64)) -> torch.Tensor:
    """TransformerUp.forward — unet_transformer.py:250-253 over MultiHeadCrossAttention.forward :179-228"""
    m = prefix + ".MHCA"
    S_pe = _q(S + _pos_enc_2d(sd[m + ".Spe.penc.inv_freq"], S))
    Sp = _conv1x1_bn_relu(F.max_pool2d(S_pe, 2), sd, m + ".Sconv_process.1", m + ".Sconv_process.2", training)
    Y_pe = _q(Y + _pos_enc_2d(sd[m + ".Ype.penc.inv_freq"], Y))
    Yp = _conv1x1_bn_relu(Y_pe, sd, m + ".Yconv_process.0", m + ".Yconv_process.1", training)
    qk = F.adaptive_avg_pool2d(Yp, res)
    low = _q(_dense_attention(qk, qk, F.adaptive_avg_pool2d(Sp, res), sd, m))
    z = _q(F.interpolate(low, size=(2 * Y.shape[2], 2 * Y.shape[3]), mode="bilinear", align_corners=True))
    z = _conv1x1_bn_relu(z, sd, m + ".conv_after_attention.0", m + ".conv_after_attention.1", training)
    y2 = _q(F.interpolate(Y_pe, scale_factor=2, mode="bilinear", align_corners=True))
    y2 = _q(F.conv2d(y2, _q(sd[m + ".Yconv2_process.1.weight"]), sd[m + ".Yconv2_process.1.bias"], padding=1))
    y2 = _conv1x1_bn_relu(y2, sd, m + ".Yconv2_process.2", m + ".Yconv2_process.3", training)
    x = torch.cat([z, y2], 1)
    x = conv_bn_relu(x, sd, prefix + ".conv.0", prefix + ".conv.1", training)
    return conv_bn_relu(x, sd, prefix + ".conv.3", prefix + ".conv.4", training)


def unet_transformer_forward(sd: State, x: torch.Tensor, training: bool, res=(64, 64)) -> torch.Tensor:
    """U_Transformer.forward — unet_transformer.py:273-283"""
    def down(t, name):
        return double_convo(F.max_pool2d(t, 2), sd, f"{name}.maxpool_conv.1.double_conv", training)

    x1 = double_conv(x, sd, "inc.conv_op", training)
    x2 = down(x1, "down1")
    x3 = down(x2, "down2")
    x4 = down(x3, "down3")
    x4p = _q(x4 + _pos_enc_2d(sd["MHSA.pe.penc.inv_freq"], x4))
    x4 = _q(_dense_attention(x4p, x4p, x4p, sd, "MHSA"))
    y = transformer_up(x4, x3, sd, "up1", training, res)
    y = transformer_up(y, x2, sd, "up2", training, res)
    y = transformer_up(y, x1, sd, "up3", training, res)
    return F.conv2d(y, sd["outc.conv.weight"], sd["outc.conv.bias"])


# ---------------------------------------------------------------------------------------------
# MultiResUNet (unet_zoo/models/multiresunet.py)
# ---------------------------------------------------------------------------------------------
def _bn_plain(x, sd: State, prefix: str, training: bool) -> torch.Tensor:
    """nn.BatchNorm2d(C, affine=False) — multiresunet.py:22, :69, :104, :116"""
    y = F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], None, None, training=training,
                     momentum=0.1, eps=1e-5)
    if training and (prefix + ".num_batches_tracked") in sd:
        sd[prefix + ".num_batches_tracked"] += 1
    return y


def conv2d_batchnorm(x, sd: State, prefix: str, training: bool, relu: bool) -> torch.Tensor:
    """Conv2d_batchnorm.forward — multiresunet.py:24-31"""
    w = sd[prefix + ".conv1.weight"]
    y = _q(F.conv2d(_q(x), _q(w), sd[prefix + ".conv1.bias"], padding=w.shape[-1] // 2))
    y = _bn_plain(y, sd, prefix + ".batchnorm", training)
    return _q(F.relu(y) if relu else y)


def multiresblock(x, sd: State, prefix: str, training: bool) -> torch.Tensor:
    """Multiresblock.forward — multiresunet.py:71-83 (batch_norm1 is applied twice)"""
    temp = conv2d_batchnorm(x, sd, prefix + ".conv2d_bn_1x1", training, False)
    a = conv2d_batchnorm(x, sd, prefix + ".conv2d_bn_3x3", training, True)
    b = conv2d_batchnorm(a, sd, prefix + ".conv2d_bn_5x5", training, True)
    c = conv2d_batchnorm(b, sd, prefix + ".conv2d_bn_7x7", training, True)
    y = _q(_bn_plain(torch.cat([a, b, c], 1), sd, prefix + ".batch_norm1", training))
    y = _q(F.relu(y + temp))
    return _q(_bn_plain(y, sd, prefix + ".batch_norm1", training))


def respath(x, sd: State, prefix: str, training: bool) -> torch.Tensor:
    """Respath.forward — multiresunet.py:121-137"""
    stages = [(prefix + ".conv2d_bn_1x1_initial", prefix + ".conv2d_bn_3x3_initial", prefix + ".batch_norm_initial")]
    i = 0
    while f"{prefix}.blocks.{i}.0.conv1.weight" in sd:
        stages.append((f"{prefix}.blocks.{i}.0", f"{prefix}.blocks.{i}.1", f"{prefix}.blocks.{i}.2"))
        i += 1
    for c1, c3, bn in stages:
        shortcut = conv2d_batchnorm(x, sd, c1, training, False)
        y = conv2d_batchnorm(x, sd, c3, training, True)
        x = _q(_bn_plain(_q(F.relu(y + shortcut)), sd, bn, training))
    return x


def multiresunet_forward(sd: State, x: torch.Tensor, training: bool) -> torch.Tensor:
    """MultiResUnet.forward — multiresunet.py:199-240"""
    m1 = multiresblock(x, sd, "multiresblock1", training)
    r1 = respath(m1, sd, "respath1", training)
    m2 = multiresblock(F.max_pool2d(m1, 2, 2), sd, "multiresblock2", training)
    r2 = respath(m2, sd, "respath2", training)
    m3 = multiresblock(F.max_pool2d(m2, 2, 2), sd, "multiresblock3", training)
    r3 = respath(m3, sd, "respath3", training)
    m4 = multiresblock(F.max_pool2d(m3, 2, 2), sd, "multiresblock4", training)
    r4 = respath(m4, sd, "respath4", training)
    y = multiresblock(F.max_pool2d(m4, 2, 2), sd, "multiresblock5", training)
    for i, skip in ((6, r4), (7, r3), (8, r2), (9, r1)):
        up = _q(F.conv_transpose2d(_q(y), _q(sd[f"upsample{i}.weight"]), sd[f"upsample{i}.bias"], stride=2))
        y = multiresblock(torch.cat([up, skip], 1), sd, f"multiresblock{i}", training)
    return conv2d_batchnorm(y, sd, "conv_final", training, False)


# ---------------------------------------------------------------------------------------------
# UCTransNet (unet_zoo/models/uctransnet.py); transformer dropouts off (rates 0.1 / 0.0 / 0.1 in the reference config)
# ---------------------------------------------------------------------------------------------
def _uct_convs(x, sd: State, prefix: str, training: bool) -> torch.Tensor:
    """_make_nConv(.., nb_Conv=2) of ConvBatchNorm — uctransnet.py:372-395"""
    i = 0
    while f"{prefix}.{i}.conv.weight" in sd:
        x = conv_bn_relu(x, sd, f"{prefix}.{i}.conv", f"{prefix}.{i}.norm", training)
        i += 1
    return x


def _uct_ln(x, sd: State, prefix: str):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], 1e-6)


def _uct_block(embs, sd: State, p: str, heads: int = 4):
    """Block_ViT.forward — uctransnet.py:260-301 over Attention_org.forward :126-226 and Mlp.forward :241-247"""
    kv = sum(e.shape[-1] for e in embs)
    emb_all = _uct_ln(torch.cat(embs, dim=2), sd, p + ".attn_norm")
    a = p + ".channel_attn"
    K = torch.stack([F.linear(emb_all, sd[f"{a}.key.{h}.weight"]) for h in range(heads)], dim=1)
    Vt = torch.stack([F.linear(emb_all, sd[f"{a}.value.{h}.weight"]) for h in range(heads)], dim=1).transpose(-1, -2)
    out = []
    for i, e in enumerate(embs):
        cx = _uct_ln(e, sd, f"{p}.attn_norm{i + 1}")
        Q = torch.stack([F.linear(cx, sd[f"{a}.query{i + 1}.{h}.weight"]) for h in range(heads)], dim=1).transpose(-1, -2)
        scores = torch.matmul(Q, K) / (kv ** 0.5)
        probs = torch.softmax(F.instance_norm(scores), dim=3)                  # nn.InstanceNorm2d(heads), no affine
        ctx = torch.matmul(probs, Vt).permute(0, 3, 2, 1).contiguous().mean(dim=3)
        cx = e + F.linear(ctx, sd[f"{a}.out{i + 1}.weight"])
        h1 = F.gelu(F.linear(_uct_ln(cx, sd, f"{p}.ffn_norm{i + 1}"), sd[f"{p}.ffn{i + 1}.fc1.weight"], sd[f"{p}.ffn{i + 1}.fc1.bias"]))
        out.append(F.linear(h1, sd[f"{p}.ffn{i + 1}.fc2.weight"], sd[f"{p}.ffn{i + 1}.fc2.bias"]) + cx)
    return out


def _uct_up(x, skip, sd: State, prefix: str, training: bool) -> torch.Tensor:
    """UpBlock_attention.forward — uctransnet.py:434-440 with CCA.forward :417-427"""
    up = _q(F.interpolate(x, scale_factor=2, mode="nearest"))
    ax = F.linear(skip.mean((2, 3)), sd[prefix + ".coatt.mlp_x.1.weight"], sd[prefix + ".coatt.mlp_x.1.bias"])
    ag = F.linear(up.mean((2, 3)), sd[prefix + ".coatt.mlp_g.1.weight"], sd[prefix + ".coatt.mlp_g.1.bias"])
    scale = torch.sigmoid((ax + ag) / 2.0)
    att = _q(F.relu(skip * scale[:, :, None, None]))
    return _uct_convs(torch.cat([att, up], 1), sd, prefix + ".nConvs", training)


def uctransnet_forward(sd: State, x: torch.Tensor, training: bool) -> torch.Tensor:
    """UCTransNet.forward — uctransnet.py:476-496 (vis=False), ChannelTransformer.forward :338-363"""
    x1 = conv_bn_relu(x, sd, "inc.conv", "inc.norm", training)
    x2 = _uct_convs(F.max_pool2d(x1, 2), sd, "down1.nConvs", training)
    x3 = _uct_convs(F.max_pool2d(x2, 2), sd, "down2.nConvs", training)
    x4 = _uct_convs(F.max_pool2d(x3, 2), sd, "down3.nConvs", training)
    x5 = _uct_convs(F.max_pool2d(x4, 2), sd, "down4.nConvs", training)
    ens, patch, embs = [x1, x2, x3, x4], (32, 16, 8, 4), []
    for i, (en, p_) in enumerate(zip(ens, patch)):
        e = _q(F.conv2d(_q(en), _q(sd[f"mtc.embeddings_{i + 1}.patch_embeddings.weight"]),
                        sd[f"mtc.embeddings_{i + 1}.patch_embeddings.bias"], stride=p_))
        embs.append(e.flatten(2).transpose(-1, -2) + sd[f"mtc.embeddings_{i + 1}.position_embeddings"])
    li = 0
    while f"mtc.encoder.layer.{li}.attn_norm.weight" in sd:
        embs = _uct_block(embs, sd, f"mtc.encoder.layer.{li}")
        li += 1
    refined = []
    for i, (e, en, p_) in enumerate(zip(embs, ens, patch)):
        e = _q(_uct_ln(e, sd, f"mtc.encoder.encoder_norm{i + 1}"))
        B, n, c = e.shape
        h = int(n ** 0.5)
        t = F.interpolate(e.permute(0, 2, 1).contiguous().view(B, c, h, h), scale_factor=p_, mode="nearest")
        r = f"mtc.reconstruct_{i + 1}"
        t = F.conv2d(t, _q(sd[r + ".conv.weight"]), sd[r + ".conv.bias"])
        t = F.relu(_bn(t, sd, r + ".norm", training))
        refined.append(_q(_q(t) + en))
    y = _uct_up(x5, refined[3], sd, "up4", training)
    y = _uct_up(y, refined[2], sd, "up3", training)
    y = _uct_up(y, refined[1], sd, "up2", training)
    y = _uct_up(y, refined[0], sd, "up1", training)
    return F.conv2d(y, sd["outc.weight"], sd["outc.bias"])


FORWARDS = {"unet": unet_forward, "attention_unet": attention_unet_forward, "u2net": u2net_forward,
            "swin_unet_v2": swin_unet_v2_forward, "nested_unet": nested_unet_forward, "resunet": resunet_forward}


# ------------------------------------------------------------------------------------------------
# MISSFormer (unet_zoo/models/missformer.py), restated on a state dict
def _mf_linear(x, sd: State, prefix: str):
    return _q(F.linear(x, sd[prefix + ".weight"], sd.get(prefix + ".bias")))


def _mf_attention(x, kv_src, sd: State, prefix: str, heads: int):
    """softmax(q k^T / sqrt(d)) v then proj (missformer.py:21-39 / :113-128); kv_src = the (reduced) tokens"""
    B, N, C = x.shape
    d = C // heads
    q = _mf_linear(x, sd, prefix + ".q").reshape(B, N, heads, d).permute(0, 2, 1, 3)
    kv = _mf_linear(kv_src, sd, prefix + ".kv").reshape(B, -1, 2, heads, d).permute(2, 0, 3, 1, 4)
    attn = ((q @ kv[0].transpose(-2, -1)) * d ** -0.5).softmax(dim=-1)
    o = _q((attn @ kv[1]).transpose(1, 2).reshape(B, N, C))
    return _mf_linear(o, sd, prefix + ".proj")


def _mf_reduce(x, sd: State, conv: str, H: int, W: int, r: int):
    """Conv2d(C, C, r, r) of the tokens viewed as a map (missformer.py:26-27, :88-96)"""
    B, N, C = x.shape
    m = x.permute(0, 2, 1).reshape(B, C, H, W)
    return _q(F.conv2d(m, sd[conv + ".weight"], sd[conv + ".bias"], stride=r).flatten(2).permute(0, 2, 1))


def _mf_mixffn_skip(x, sd: State, prefix: str, H: int, W: int):
    """missformer.py:203-208"""
    f = _mf_linear(x, sd, prefix + ".fc1")
    B, N, C = f.shape
    dw = F.conv2d(f.transpose(1, 2).reshape(B, C, H, W), sd[prefix + ".dwconv.dwconv.weight"],
                  sd[prefix + ".dwconv.dwconv.bias"], padding=1, groups=C).flatten(2).transpose(1, 2)
    a = _q(F.gelu(_q(_layer_norm(_q(dw + f), sd, prefix + ".norm1"))))
    return _mf_linear(a, sd, prefix + ".fc2")


def _mf_mixffn(x, sd: State, prefix: str, H: int, W: int):
    """MixFFN.forward — missformer.py:186-189 (token_mlp='mix')"""
    f = _mf_linear(x, sd, prefix + ".fc1")
    B, N, C = f.shape
    dw = F.conv2d(f.transpose(1, 2).reshape(B, C, H, W), sd[prefix + ".dwconv.dwconv.weight"],
                  sd[prefix + ".dwconv.dwconv.bias"], padding=1, groups=C).flatten(2).transpose(1, 2)
    return _mf_linear(_q(F.gelu(_q(dw))), sd, prefix + ".fc2")


def _mf_block(x, sd: State, prefix: str, H: int, W: int, heads: int, r: int):
    """TransformerBlock (missformer.py:265-268)"""
    n1 = _q(_layer_norm(x, sd, prefix + ".norm1"))
    red = _q(_layer_norm(_mf_reduce(n1, sd, prefix + ".attn.sr", H, W, r), sd, prefix + ".attn.norm")) if r > 1 else n1
    tx = _q(x + _mf_attention(n1, red, sd, prefix + ".attn", heads))
    n2 = _q(_layer_norm(tx, sd, prefix + ".norm2"))
    if prefix + ".mlp.norm1.weight" in sd:         # token_mlp='mix_skip' (the default)
        return _q(tx + _mf_mixffn_skip(n2, sd, prefix + ".mlp", H, W))
    return _q(tx + _mf_mixffn(n2, sd, prefix + ".mlp", H, W))


def _mf_expand(x, sd: State, prefix: str, H: int, W: int, r: int):
    """PatchExpand / FinalPatchExpand_X4 (missformer.py:522-537, :549-564)"""
    y = _mf_linear(x, sd, prefix + ".expand")
    B, _, Ce = y.shape
    c = Ce // (r * r)
    y = y.reshape(B, H, W, r, r, c).permute(0, 1, 3, 2, 4, 5).reshape(B, H * r * W * r, c)
    return _q(_layer_norm(y, sd, prefix + ".norm"))


def missformer_forward(sd: State, x: torch.Tensor, training: bool, image_size: int = 512) -> torch.Tensor:
    """MISSFormer.forward (missformer.py:922-938) for the default B1 / mix_skip configuration"""
    dims, ratios, heads = [64, 128, 320, 512], [8, 4, 2, 1], [1, 2, 5, 8]
    if x.shape[1] == 1:
        x = x.repeat(1, 3, 1, 1)
    B = x.shape[0]
    feats, t = [], x
    for i in range(4):                                              # MiT.forward (:336-368)
        pe = f"backbone.patch_embed{i + 1}"
        k, s, p = (7, 4, 3) if i == 0 else (3, 2, 1)
        m = _q(F.conv2d(t if i == 0 else t_map, sd[pe + ".proj.weight"], sd[pe + ".proj.bias"], stride=s, padding=p))
        H, W = m.shape[2:]
        t = _q(_layer_norm(m.flatten(2).transpose(1, 2), sd, pe + ".norm"))
        j = 0
        while f"backbone.block{i + 1}.{j}.norm1.weight" in sd:
            t = _mf_block(t, sd, f"backbone.block{i + 1}.{j}", H, W, heads[i], ratios[i])
            j += 1
        t = _q(_layer_norm(t, sd, f"backbone.norm{i + 1}"))
        t_map = t.reshape(B, H, W, -1).permute(0, 3, 1, 2)
        feats.append((t, H, W))
    res = [(image_size // s, image_size // s) for s in (4, 8, 16, 32)]
    cuts = [0]
    for h, w in res:
        cuts.append(cuts[-1] + h * w)
    tok = None
    for li in range(1, 5):                                          # BridgeLayer_4.forward (:665-702)
        bp = f"bridge.bridge_layer{li}"
        if tok is None:
            tok = torch.cat([_mf_linear(f, sd, f"{bp}.proj_c{i + 1}") for i, (f, _, _) in enumerate(feats)], -2)
        n1 = _q(_layer_norm(tok, sd, bp + ".norm1"))
        red = []
        for i, ((h, w), r) in enumerate(zip(res, ratios)):          # Scale_reduce (:81-100)
            sl = n1[:, cuts[i]:cuts[i + 1], :]
            red.append(_mf_reduce(sl, sd, f"{bp}.attn.scale_reduce.sr_convs.{i}", h, w, r) if r > 1 else sl)
        red = _q(_layer_norm(torch.cat(red, -2), sd, bp + ".attn.scale_reduce.norm"))
        tx1 = _q(tok + _mf_attention(n1, red, sd, bp + ".attn", 1))
        tx = _q(_layer_norm(tx1, sd, bp + ".norm2"))
        ffn = torch.cat([_mf_mixffn_skip(tx[:, cuts[i]:cuts[i + 1], :], sd, f"{bp}.mixffn{i + 1}", h, w)
                         for i, (h, w) in enumerate(res)], -2)
        tok = _q(tx1 + ffn)
    skips = [_mf_linear(tok[:, cuts[i]:cuts[i + 1], :], sd, f"bridge.proj_back_c{i + 1}") for i in range(4)]   # :801-811
    t = skips[3]
    for k in (3, 2, 1, 0):                                          # SegU_decoder.forward (:602-633)
        dp, (h, w) = f"decoder_{k}", res[k]
        if k < 3:
            t = _mf_linear(torch.cat([t, skips[k]], -1), sd, dp + ".concat_linear")
        t = _mf_block(t, sd, dp + ".layer_former_1", h, w, heads[k], ratios[k])
        t = _mf_block(t, sd, dp + ".layer_former_2", h, w, heads[k], ratios[k])
        t = _mf_expand(t, sd, dp + ".layer_up", h, w, 4 if k == 0 else 2)
    S = res[0][0] * 4
    return F.conv2d(t.reshape(B, S, S, -1).permute(0, 3, 1, 2), sd["decoder_0.last_layer.weight"], sd["decoder_0.last_layer.bias"])


FORWARDS["missformer"] = missformer_forward
FORWARDS["transatt_unet"] = transatt_unet_forward
FORWARDS["unet_transformer"] = unet_transformer_forward
FORWARDS["multiresunet"] = multiresunet_forward
FORWARDS["uctransnet"] = uctransnet_forward


def clone_state(sd: State, requires_grad: bool = False) -> "OrderedDict[str, torch.Tensor]":
    out = OrderedDict()
    for k, v in sd.items():
        t = v.detach().clone().cpu()
        if requires_grad and t.is_floating_point() and not _is_buffer(k):
            t.requires_grad_(True)
        out[k] = t
    return out


def _is_buffer(key: str) -> bool:
    return key.endswith(("running_mean", "running_var", "num_batches_tracked", "attn_mask",
                         "log_relative_position_index", "inv_freq"))


def train_step_reference(model_name: str, sd: State, x: torch.Tensor, mask: torch.Tensor, **fw_kwargs):
    """One forward + BCEWithLogits + backward of the reference step
    (unet_zoo/utils/training_loop.py:112-119, criterion from scripts/train.py:135).
    Returns (logits, loss, {param name: grad}, updated state with new running stats)."""
    st = clone_state(sd, requires_grad=True)
    logits = FORWARDS[model_name](st, x.float().cpu(), True, **fw_kwargs)
    loss = model_loss(logits, mask.float().cpu())
    names = [k for k, v in st.items() if v.requires_grad]
    grads = torch.autograd.grad(loss, [st[k] for k in names], allow_unused=True)
    names, grads = zip(*[(n, g) for n, g in zip(names, grads) if g is not None])  # swin: mlp / norm2 are unused
    if isinstance(logits, dict):
        logits = {k: v.detach() for k, v in logits.items()}
    elif isinstance(logits, (list, tuple)):
        logits = [v.detach() for v in logits]
    else:
        logits = logits.detach()
    return logits, loss.detach(), dict(zip(names, grads)), st


def synthetic_batch(B: int, C: int, H: int, W: int, seed: int = 1):
    """The fixture input protocol (SURVEY.md §8c iii): randn image, rand>0.5 mask, one generator."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=g)
    mask = (torch.rand(B, 1, H, W, generator=g) > 0.5).float()
    return x, mask
