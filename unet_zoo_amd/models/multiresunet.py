"""MultiResUNet on the HIP engine (reference graph: unet_zoo/models/multiresunet.py:32-240).

The reference's channel counts (8 / 17 / 26 / 51 ... 142 / 284 / 427 / 853) are not multiples of the 16-byte vector
every kernel of this library addresses, so this model runs in a CHANNEL-PADDED layout: every tensor's channel axis
is rounded up to a multiple of 8, a concat keeps each part's padding (holes in the middle), and a layer's parameters
live in the padded shape with zeros in the padding -- zeros that stay zero (their gradients are exactly zero: a padded
input channel is always 0, a padded output channel only feeds zero weight columns; weight decay and AdamW keep 0 at
0).  ``state_dict()`` / ``load_state_dict()`` present the reference's shapes and names (hooks gather / scatter through
the channel maps), the seed-0 values are drawn by the same initialisers in the same order.

Blocks: ``Conv2d_batchnorm`` (:7-31; BatchNorm2d(affine=False)), ``Multiresblock`` (:32-83: three chained 3x3
convolutions written into one concat buffer, a 1x1 shortcut laid out like that concat, BN, relu(x + shortcut), the
SAME BN module again), ``Respath`` (:85-137), MaxPool2d(2, 2) fused into the block's last BatchNorm pass,
ConvTranspose2d(k2, s2) into its concat slot, and a head that ends in a BatchNorm (:196-197).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from ..engine import Engine
from ..graph import HipModule
from ..ops import Act

PAD = 8


def _rup(c: int) -> int:
    return (c + PAD - 1) // PAD * PAD


class Layout:
    """physical channel position of every logical channel of a tensor"""

    def __init__(self, index: Sequence[int], width: int):
        self.index = list(index)
        self.width = width

    @staticmethod
    def plain(c: int) -> "Layout":
        return Layout(range(c), _rup(c))

    @staticmethod
    def cat(parts: Sequence["Layout"]) -> "Layout":
        idx, off = [], 0
        for p in parts:
            idx += [off + i for i in p.index]
            off += p.width
        return Layout(idx, off)

    @property
    def logical(self) -> int:
        return len(self.index)


class _Padded(nn.Module):
    """parameters / buffers held in padded shape, shown to state_dict in the reference's shape"""

    def _install_hooks(self, names_maps):
        self._maps = names_maps    # name -> tuple of (dim, index list) pairs

        def to_ref(module, sd, prefix, _meta):
            for name, dims in module._maps.items():
                k = prefix + name
                if k in sd:
                    t = sd[k]
                    for dim, idx in dims:
                        t = t.index_select(dim, torch.as_tensor(idx, device=t.device))
                    sd[k] = t
            return sd

        def from_ref(sd, prefix, *_):
            for name, dims in self._maps.items():
                k = prefix + name
                if k in sd:
                    cur = getattr(self, name)
                    t = sd[k]
                    if tuple(t.shape) == tuple(cur.shape):
                        continue                      # already padded (a copy of this module's own tensors)
                    full = torch.zeros(cur.shape, dtype=t.dtype, device=t.device)
                    if name == "running_var":
                        full.fill_(1.0)
                    view = full
                    sl: List = [slice(None)] * full.dim()
                    if len(dims) == 1:
                        sl[dims[0][0]] = torch.as_tensor(dims[0][1])
                        view[tuple(sl)] = t
                    else:                              # two mapped axes: rows then columns
                        (d0, i0), (d1, i1) = dims
                        assert (d0, d1) == (0, 1)
                        full[torch.as_tensor(i0)[:, None], torch.as_tensor(i1)[None, :]] = t
                    sd[k] = full

        self._register_state_dict_hook(to_ref)
        self._register_load_state_dict_pre_hook(from_ref)


class PadConv2d(_Padded):
    """nn.Conv2d(k, padding) whose weight is stored (Cout_p, Cin_p, k, k) with the logical channels at the positions
    of the layouts; quacks like nn.Conv2d for the engine"""

    def __init__(self, lin: Layout, lout: Layout, k: int, padding: int):
        super().__init__()
        ref = nn.Conv2d(lin.logical, lout.logical, kernel_size=(k, k), padding=padding)   # the reference's initial values
        w = torch.zeros(lout.width, lin.width, k, k)
        w[torch.as_tensor(lout.index)[:, None], torch.as_tensor(lin.index)[None, :]] = ref.weight.detach()
        b = torch.zeros(lout.width)
        b[torch.as_tensor(lout.index)] = ref.bias.detach()
        self.weight, self.bias = nn.Parameter(w), nn.Parameter(b)
        self.kernel_size, self.stride, self.padding, self.dilation = (k, k), (1, 1), (padding, padding), (1, 1)
        self.in_channels, self.out_channels = lin.width, lout.width
        self._install_hooks({"weight": ((0, lout.index), (1, lin.index)), "bias": ((0, lout.index),)})


class PadConvTranspose2d(_Padded):
    def __init__(self, lin: Layout, cout: int):
        super().__init__()
        assert cout % PAD == 0
        ref = nn.ConvTranspose2d(lin.logical, cout, kernel_size=(2, 2), stride=(2, 2), padding=0)
        w = torch.zeros(lin.width, cout, 2, 2)
        w[torch.as_tensor(lin.index)] = ref.weight.detach()
        self.weight, self.bias = nn.Parameter(w), nn.Parameter(ref.bias.detach().clone())
        self.kernel_size, self.stride = (2, 2), (2, 2)
        self.in_channels, self.out_channels = lin.width, cout
        self._install_hooks({"weight": ((0, lin.index),)})


class PadBatchNorm2d(_Padded):
    """nn.BatchNorm2d(C, affine=False) over a padded channel axis: gamma = 1, beta = 0 are constants"""

    def __init__(self, layout: Layout):
        super().__init__()
        C = layout.width
        self.num_features, self.eps, self.momentum = C, 1e-5, 0.1
        self.register_buffer("running_mean", torch.zeros(C))
        self.register_buffer("running_var", torch.ones(C))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.register_buffer("weight", torch.ones(C), persistent=False)
        self.register_buffer("bias", torch.zeros(C), persistent=False)
        self._install_hooks({"running_mean": ((0, layout.index),), "running_var": ((0, layout.index),)})


class Conv2d_batchnorm(nn.Module):
    def __init__(self, lin: Layout, lout: Layout, kernel_size=(2, 2), activation='relu', padding=0):
        super().__init__()
        self.activation = activation
        self.conv1 = PadConv2d(lin, lout, kernel_size[0], padding)
        self.batchnorm = PadBatchNorm2d(lout)

    def emit(self, eng: Engine, x: Act, out: Optional[Act] = None) -> Act:
        act, _ = eng.conv_bn_relu(x, self.conv1, self.batchnorm, out=out, relu=(self.activation == 'relu'))
        return act


def mrb_filters(unet_filters: int, alpha: float = 1.67) -> Tuple[int, int, int]:
    W = int(unet_filters * alpha)
    return int(W * 0.167), int(W * 0.333), int(W * 0.5)


class Multiresblock(nn.Module):
    def __init__(self, lin: Layout, corresponding_unet_filters: int, alpha: float = 1.67):
        super().__init__()
        self.corresponding_unet_filters, self.alpha = corresponding_unet_filters, alpha
        self.W = int(corresponding_unet_filters * alpha)
        f3, f5, f7 = mrb_filters(corresponding_unet_filters, alpha)
        la, lb, lc = Layout.plain(f3), Layout.plain(f5), Layout.plain(f7)
        self.parts = (la, lb, lc)
        self.lout = Layout.cat(self.parts)
        self.conv2d_bn_1x1 = Conv2d_batchnorm(lin, self.lout, kernel_size=(1, 1), activation='None', padding=0)
        self.conv2d_bn_3x3 = Conv2d_batchnorm(lin, la, kernel_size=(3, 3), activation='relu', padding=1)
        self.conv2d_bn_5x5 = Conv2d_batchnorm(la, lb, kernel_size=(3, 3), activation='relu', padding=1)
        self.conv2d_bn_7x7 = Conv2d_batchnorm(lb, lc, kernel_size=(3, 3), activation='relu', padding=1)
        self.batch_norm1 = PadBatchNorm2d(self.lout)

    def emit(self, eng: Engine, x: Act, pool: bool = False):
        temp = self.conv2d_bn_1x1.emit(eng, x)
        full, (sa, sb, sc) = eng.new_cat(x.N, x.H, x.W, [p.width for p in self.parts])     # cat([a, b, c], 1)
        a = self.conv2d_bn_3x3.emit(eng, x, out=sa)
        b = self.conv2d_bn_5x5.emit(eng, a, out=sb)
        self.conv2d_bn_7x7.emit(eng, b, out=sc)
        t = eng.bn_act(full, self.batch_norm1, relu=False)
        r = eng.add_relu(t, temp)
        return eng.bn_act(r, self.batch_norm1, relu=False, pool=pool)       # the same module a second time (:81)


class _RespathBlock(nn.Sequential):
    pass


class Respath(nn.Module):
    def __init__(self, lin: Layout, filters: int, respath_length: int):
        super().__init__()
        self.filters, self.respath_length = filters, respath_length
        lf = Layout.plain(filters)
        self.conv2d_bn_1x1_initial = Conv2d_batchnorm(lin, lf, kernel_size=(1, 1), activation='None', padding=0)
        self.conv2d_bn_3x3_initial = Conv2d_batchnorm(lin, lf, kernel_size=(3, 3), activation='relu', padding=1)
        self.batch_norm_initial = PadBatchNorm2d(lf)
        self.blocks = nn.ModuleList()
        for _ in range(respath_length - 1):
            self.blocks.append(nn.Sequential(
                Conv2d_batchnorm(lf, lf, kernel_size=(1, 1), activation='None', padding=0),
                Conv2d_batchnorm(lf, lf, kernel_size=(3, 3), activation='relu', padding=1),
                PadBatchNorm2d(lf)))

    def emit(self, eng: Engine, x: Act, out: Optional[Act] = None) -> Act:
        stages = [(self.conv2d_bn_1x1_initial, self.conv2d_bn_3x3_initial, self.batch_norm_initial)]
        stages += [(b[0], b[1], b[2]) for b in self.blocks]
        for i, (c1, c3, bn) in enumerate(stages):
            shortcut = c1.emit(eng, x)
            r = eng.add_relu(c3.emit(eng, x), shortcut)
            x = eng.bn_act(r, bn, relu=False)
        if out is not None:
            eng.copy_into(x, out)
            return out
        return x


class MultiResUnet(HipModule):
    def __init__(self, in_channels: int, filters: int = 32, num_classes: int = 1, **kwargs):
        super().__init__()
        self.alpha, self.filters, self.nclasses, self.in_channels = 1.67, filters, num_classes, in_channels
        if filters % PAD:
            raise ValueError(f"filters must be a multiple of {PAD}, got {filters}")
        f = filters
        lin = Layout.plain(in_channels)
        self.multiresblock1 = Multiresblock(lin, f)
        self.pool1 = nn.MaxPool2d(2, stride=2)
        self.respath1 = Respath(self.multiresblock1.lout, f, respath_length=4)
        self.multiresblock2 = Multiresblock(self.multiresblock1.lout, f * 2)
        self.pool2 = nn.MaxPool2d(2, 2)
        self.respath2 = Respath(self.multiresblock2.lout, f * 2, respath_length=3)
        self.multiresblock3 = Multiresblock(self.multiresblock2.lout, f * 4)
        self.pool3 = nn.MaxPool2d(2, 2)
        self.respath3 = Respath(self.multiresblock3.lout, f * 4, respath_length=2)
        self.multiresblock4 = Multiresblock(self.multiresblock3.lout, f * 8)
        self.pool4 = nn.MaxPool2d(2, 2)
        self.respath4 = Respath(self.multiresblock4.lout, f * 8, respath_length=1)
        self.multiresblock5 = Multiresblock(self.multiresblock4.lout, f * 16)

        def up(prev: Multiresblock, c: int):
            return PadConvTranspose2d(prev.lout, c), Multiresblock(Layout.plain(2 * c), c)

        self.upsample6, self.multiresblock6 = up(self.multiresblock5, f * 8)
        self.upsample7, self.multiresblock7 = up(self.multiresblock6, f * 4)
        self.upsample8, self.multiresblock8 = up(self.multiresblock7, f * 2)
        self.upsample9, self.multiresblock9 = up(self.multiresblock8, f)
        self.conv_final = Conv2d_batchnorm(self.multiresblock9.lout, Layout.plain(num_classes), kernel_size=(1, 1),
                                           activation='None')

    def emit(self, eng: Engine, x: torch.Tensor):
        N, _, H, W = x.shape
        if H % 16 or W % 16:
            raise ValueError(f"MultiResUnet needs H, W divisible by 16 (four 2x2 poolings and x2 upsamplings), got {H}x{W}")
        f = self.filters
        enc = ((self.multiresblock1, self.respath1, f), (self.multiresblock2, self.respath2, 2 * f),
               (self.multiresblock3, self.respath3, 4 * f), (self.multiresblock4, self.respath4, 8 * f))
        cats = []
        cur = eng.input_nhwc(x, PAD)
        for lvl, (blk, rp, c) in enumerate(enc):
            full, (up_slot, skip_slot) = eng.new_cat(N, H >> lvl, W >> lvl, (c, c))     # cat([upsample(.), respath], 1)
            cats.append((full, up_slot))
            xm, pooled = blk.emit(eng, cur, pool=True)
            rp.emit(eng, xm, out=skip_slot)
            cur = pooled
        cur = self.multiresblock5.emit(eng, cur)
        for lvl, ups, blk in ((3, self.upsample6, self.multiresblock6), (2, self.upsample7, self.multiresblock7),
                              (1, self.upsample8, self.multiresblock8), (0, self.upsample9, self.multiresblock9)):
            full, up_slot = cats[lvl]
            eng.conv_transpose2x2(cur, ups, up_slot)
            cur = blk.emit(eng, full)
        head = self.conv_final.emit(eng, cur)
        return (eng.act_to_logits(head, self.nclasses),)
