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


# ---------------------------------------------------------------------------------------------
# primitives
# ---------------------------------------------------------------------------------------------
def conv_bn_relu(x: torch.Tensor, sd: State, conv: str, bn: str, training: bool,
                 dilation: int = 1) -> torch.Tensor:
    """Conv2d(k3, padding=dilation) -> BatchNorm2d -> ReLU.
    unet_zoo/models/common_layers.py:28-30 (and :31-33, :47-56; u2net.py:10-17 with dilation).
    Train mode: batch mean / biased variance, running stats updated with momentum 0.1 and the
    unbiased variance (torch.nn.BatchNorm2d defaults, eps 1e-5)."""
    x = F.conv2d(x, sd[conv + ".weight"], sd.get(conv + ".bias"), padding=dilation, dilation=dilation)
    x = F.batch_norm(x, sd[bn + ".running_mean"], sd[bn + ".running_var"], sd[bn + ".weight"],
                     sd[bn + ".bias"], training=training, momentum=0.1, eps=1e-5)
    if training and (bn + ".num_batches_tracked") in sd:
        sd[bn + ".num_batches_tracked"] += 1
    return F.relu(x)


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
    x1 = F.conv_transpose2d(x1, sd[f"{prefix}.up.weight"], sd[f"{prefix}.up.bias"], stride=2)
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
    x = F.conv2d(x, sd[f"{prefix}.0.weight"], sd[f"{prefix}.0.bias"])
    x = F.batch_norm(x, sd[f"{prefix}.1.running_mean"], sd[f"{prefix}.1.running_var"], sd[f"{prefix}.1.weight"],
                     sd[f"{prefix}.1.bias"], training=training, momentum=0.1, eps=1e-5)
    if training and f"{prefix}.1.num_batches_tracked" in sd:
        sd[f"{prefix}.1.num_batches_tracked"] += 1
    return x


def attention_block(g, x, sd: State, prefix: str, training: bool) -> torch.Tensor:
    """AttentionBlock.forward — attention_unet.py:34-40."""
    g1 = _conv1x1_bn(g, sd, f"{prefix}.w_g", training)
    x1 = _conv1x1_bn(x, sd, f"{prefix}.w_x", training)
    psi = torch.sigmoid(_conv1x1_bn(F.relu(g1 + x1), sd, f"{prefix}.psi", training))
    return psi * x


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


FORWARDS = {"unet": unet_forward, "attention_unet": attention_unet_forward}


def clone_state(sd: State, requires_grad: bool = False) -> "OrderedDict[str, torch.Tensor]":
    out = OrderedDict()
    for k, v in sd.items():
        t = v.detach().clone().cpu()
        if requires_grad and t.is_floating_point() and not _is_buffer(k):
            t.requires_grad_(True)
        out[k] = t
    return out


def _is_buffer(key: str) -> bool:
    return key.endswith(("running_mean", "running_var", "num_batches_tracked"))


def train_step_reference(model_name: str, sd: State, x: torch.Tensor, mask: torch.Tensor):
    """One forward + BCEWithLogits + backward of the reference step
    (unet_zoo/utils/training_loop.py:112-119, criterion from scripts/train.py:135).
    Returns (logits, loss, {param name: grad}, updated state with new running stats)."""
    st = clone_state(sd, requires_grad=True)
    logits = FORWARDS[model_name](st, x.float().cpu(), True)
    loss = F.binary_cross_entropy_with_logits(logits, mask.float().cpu())
    names = [k for k, v in st.items() if v.requires_grad]
    grads = torch.autograd.grad(loss, [st[k] for k in names])
    return logits.detach(), loss.detach(), dict(zip(names, grads)), st


def synthetic_batch(B: int, C: int, H: int, W: int, seed: int = 1):
    """The fixture input protocol (SURVEY.md §8c iii): randn image, rand>0.5 mask, one generator."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=g)
    mask = (torch.rand(B, 1, H, W, generator=g) > 0.5).float()
    return x, mask
