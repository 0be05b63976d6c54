"""ResUnet on the HIP engine (reference graph: unet_zoo/models/resunet.py:9-78 with ResidualConv /
UpsampleResUnet of common_layers.py:182-207).

Pre-activation residual blocks: BN -> ReLU -> Conv3x3(stride s) -> BN -> ReLU -> Conv3x3, plus a skip
Conv1x1(stride s) -> BN, summed.  A BatchNorm here normalises a SUM (or the network input's first block), not a
convolution output, so its statistics come from one extra pass (`Engine.bn_act`); the stride-1 middle pair
Conv -> BN -> ReLU is the fused kernel of the other models.  Stride 2: the 3x3 convolution is a nine-tap strided
gather on the LDS-DMA GEMM (`Engine.conv3x3_s2`, `UZ_TAPS_CONV_S2`), its weight gradient the same gather in the
LDS-DMA weight-gradient kernel; only its input gradient takes the dense route (dy between zeros, stride-1
convolution).  The 1x1 skip convolution reads every second pixel (`Engine.subsample2`).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ..engine import Engine
from ..graph import HipModule
from ..ops import Act


class ResidualConv(nn.Module):
    """common_layers.py:182-199 (children own the parameters, `emit` does the arithmetic)."""

    def __init__(self, in_channels, out_channels, stride, padding):
        super().__init__()
        assert padding == 1 and stride in (1, 2)
        self.stride = stride
        self.conv_block = nn.Sequential(
            nn.BatchNorm2d(in_channels),
            nn.ReLU(),
            nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=stride, padding=padding, bias=False),
            nn.BatchNorm2d(out_channels),
            nn.ReLU(),
            nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1, bias=False),
        )
        self.conv_skip = nn.Sequential(
            nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=stride, bias=False),
            nn.BatchNorm2d(out_channels),
        )

    def emit(self, eng: Engine, x: Act) -> Act:
        cb, cs = self.conv_block, self.conv_skip
        h = eng.bn_act(x, cb[0], relu=True)
        if self.stride == 1:
            h, _ = eng.conv_bn_relu(h, cb[2], cb[3])                       # Conv -> BN -> ReLU, fused statistics
            s = x
        else:
            h = eng.bn_act(eng.conv3x3_s2(h, cb[2]), cb[3], relu=True)
            s = eng.subsample2(x)
        h = eng.conv_plain(h, cb[5])
        s = eng.bn_act(eng.conv_plain(s, cs[0]), cs[1], relu=False)
        return eng.add(h, s)


class UpsampleResUnet(nn.Module):
    """common_layers.py:201-207"""

    def __init__(self, in_channels, out_channels, kernel_size, stride):
        super().__init__()
        assert kernel_size == 2 and stride == 2
        self.upsample = nn.ConvTranspose2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride)


class ResUnet(HipModule):
    """Same constructor as the reference (resunet.py:10-50)."""

    def __init__(self, in_channels: int = 3, num_classes: int = 1, filters: list = None):
        super().__init__()
        if filters is None:
            filters = [64, 128, 256, 512]
        if num_classes > 1:
            print(f"Warning: ResUnet output layer is set for 1 class by default. "
                  f"For {num_classes} classes, consider changing the final Conv2d output channel.")
            self.final_conv_out_channels = num_classes
        else:
            self.final_conv_out_channels = 1
        self.filters = f = list(filters)

        def conv3(cin, cout):
            return nn.Conv2d(cin, cout, kernel_size=3, padding=1)

        # registration order = the reference's (resunet.py:26-52): it fixes the state_dict order and the order the
        # default initialisers draw from the RNG
        self.input_layer = nn.Sequential(conv3(in_channels, f[0]), nn.BatchNorm2d(f[0]), nn.ReLU(), conv3(f[0], f[0]))
        self.input_skip = nn.Sequential(conv3(in_channels, f[0]))
        for name, cin, cout in (("residual_conv_1", f[0], f[1]), ("residual_conv_2", f[1], f[2]), ("bridge", f[2], f[3])):
            setattr(self, name, ResidualConv(cin, cout, 2, 1))
        for k, (cin, cout) in enumerate(((f[3], f[2]), (f[2], f[1]), (f[1], f[0])), 1):
            setattr(self, f"upsample_{k}", UpsampleResUnet(cin, cout, 2, 2))
            setattr(self, f"up_residual_conv{k}", ResidualConv(cout + cout, cout, 1, 1))
        self.output_layer = nn.Sequential(nn.Conv2d(f[0], self.final_conv_out_channels, 1, 1))

    def emit(self, eng: Engine, x: torch.Tensor):
        N, _, H, W = x.shape
        if H % 8 or W % 8:
            raise ValueError(f"ResUnet needs H, W divisible by 8 (three stride-2 blocks and x2 upsamplings), got {H}x{W}")
        f = self.filters
        patches = eng.input_im2col(x)
        h, _ = eng.conv_bn_relu(patches, self.input_layer[0], self.input_layer[1], im2col=True)
        x1 = eng.add(eng.conv_plain(h, self.input_layer[3]), eng.conv_plain(patches, self.input_skip[0], im2col=True))
        x2 = self.residual_conv_1.emit(eng, x1)
        x3 = self.residual_conv_2.emit(eng, x2)
        x4 = self.bridge.emit(eng, x3)

        def up_cat(low: Act, skip: Act, up: UpsampleResUnet, c: int) -> Act:
            """torch.cat([upsample(low), skip], 1) (resunet.py:60-61): both written into one buffer"""
            full, (up_slot, skip_slot) = eng.new_cat(N, skip.H, skip.W, (c, c))
            eng.conv_transpose2x2(low, up.upsample, up_slot)
            eng.copy_into(skip, skip_slot)
            return full

        x6 = self.up_residual_conv1.emit(eng, up_cat(x4, x3, self.upsample_1, f[2]))
        x8 = self.up_residual_conv2.emit(eng, up_cat(x6, x2, self.upsample_2, f[1]))
        x10 = self.up_residual_conv3.emit(eng, up_cat(x8, x1, self.upsample_3, f[0]))
        return (eng.out_conv(x10, self.output_layer[0]),)
