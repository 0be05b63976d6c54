"""UNet++ (NestedUNet) on the HIP engine (reference graph: unet_zoo/models/nested_unet.py:4-105).

Every node x_{i,j} is a VGGBlock (two Conv3x3 + BN + ReLU) on ``cat([x_{i,0}, ..., x_{i,j-1}, up(x_{i+1,j-1})])``
with ``up`` = bilinear x2, align_corners=True.  All concat buffers are allocated up front: a node is written
straight into its slot of the FIRST concat that reads it (its later readers get a copy, the slots'
gradients flow back without copies), the upsampling writes its slot directly, the 2x2 pooling of the
backbone column is fused into the producing BN/ReLU pass.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from ..engine import Engine
from ..graph import HipModule
from ..ops import Act


class VGGBlock(nn.Module):
    """conv1-bn1-relu-conv2-bn2-relu (nested_unet.py:4-22); the children only own parameters."""

    def __init__(self, in_channels: int, middle_channels: int, out_channels: int):
        super().__init__()
        self.relu = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(in_channels, middle_channels, 3, padding=1)
        self.bn1 = nn.BatchNorm2d(middle_channels)
        self.conv2 = nn.Conv2d(middle_channels, out_channels, 3, padding=1)
        self.bn2 = nn.BatchNorm2d(out_channels)

    def emit(self, eng: Engine, x: Act, *, out: Optional[Act] = None, pool: bool = False,
             im2col: bool = False) -> Tuple[Act, Optional[Act]]:
        mid, _ = eng.conv_bn_relu(x, self.conv1, self.bn1, im2col=im2col, defer_apply=self.conv2)   # Engine.fold_bn_apply
        return eng.conv_bn_relu(mid, self.conv2, self.bn2, out=out, pool=pool, sole_reader=True)


class NestedUNet(HipModule):
    """Same constructor as the reference (nested_unet.py:24-66): `num_classes` first, five widths 32..512."""

    def __init__(self, num_classes, in_channels=3, deep_supervision=False, **kwargs):
        super().__init__()
        nb = [32, 64, 128, 256, 512]
        self.nb_filter = nb
        self.deep_supervision = deep_supervision
        self.pool = nn.MaxPool2d(2, 2)
        self.up = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)

        # node x_{i,j}: row i (resolution H >> i), column j (number of dense skips it reads).  Registered column by
        # column, rows ascending -- the reference's construction order (nested_unet.py:34-57), which fixes both the
        # state_dict order and the order the default initialisers draw from the RNG.
        for j in range(5):
            for i in range(5 - j):
                if j == 0:
                    cin = in_channels if i == 0 else nb[i - 1]
                else:
                    cin = nb[i] * j + nb[i + 1]
                setattr(self, f"conv{i}_{j}", VGGBlock(cin, nb[i], nb[i]))

        if self.deep_supervision:
            for j in range(1, 5):
                setattr(self, f"final{j}", nn.Conv2d(nb[0], num_classes, kernel_size=1))
        else:
            self.final = nn.Conv2d(nb[0], num_classes, kernel_size=1)

    def wrap_outputs(self, outs):
        # the reference returns a list of four maps under deep supervision, one tensor otherwise (:95-105)
        return list(outs) if self.deep_supervision else outs[0]

    def emit(self, eng: Engine, x: torch.Tensor):
        N, _, H, W = x.shape
        if H % 16 or W % 16:
            raise ValueError(f"NestedUNet needs H, W divisible by 16 (four 2x2 poolings and x2 upsamplings), got {H}x{W}")
        nb = self.nb_filter
        # cats[(i, j)] for node x_{i,j}, j >= 1: slots [x_{i,0}, ..., x_{i,j-1}, up(x_{i+1,j-1})]
        cats: Dict[Tuple[int, int], Tuple[Act, list]] = {}
        for i in range(4):
            for j in range(1, 5 - i):
                cats[(i, j)] = eng.new_cat(N, H >> i, W >> i, [nb[i]] * j + [nb[i + 1]])
        node: Dict[Tuple[int, int], Act] = {}

        def home(i: int, j: int) -> Optional[Act]:
            """slot of x_{i,j} in the first concat that reads it (None: no same-row reader)"""
            return cats[(i, j + 1)][1][j] if (i, j + 1) in cats else None

        def gather(i: int, j: int) -> Act:
            """fill the concat of x_{i,j}: copies of the older same-row nodes, the upsampled lower node"""
            full, parts = cats[(i, j)]
            for k in range(j - 1):                       # x_{i,j-1} was written into its slot by its producer
                eng.copy_into(node[(i, k)], parts[k])
            eng.resize_bilinear(node[(i + 1, j - 1)], parts[j], align_corners=True)
            return full

        # backbone column and the nested nodes in the reference's order (nested_unet.py:74-93)
        cur = eng.input_im2col(x)
        pooled = None
        for i in range(5):
            src = cur if i == 0 else pooled
            node[(i, 0)], pooled = getattr(self, f"conv{i}_0").emit(eng, src, out=home(i, 0), pool=(i < 4), im2col=(i == 0))
            for r in range(i - 1, -1, -1):               # the anti-diagonal that x_{i,0} completes
                j = i - r
                node[(r, j)], _ = getattr(self, f"conv{r}_{j}").emit(eng, gather(r, j), out=home(r, j))
        if self.deep_supervision:
            return tuple(eng.out_conv(node[(0, j)], getattr(self, f"final{j}")) for j in range(1, 5))
        return (eng.out_conv(node[(0, 4)], self.final),)
