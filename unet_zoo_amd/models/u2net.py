"""U^2-Net / U^2-NetP on the HIP engine (reference graph: unet_zoo/models/u2net.py:6-381).

Module and attribute names follow the reference (``stage1.rebnconvin.conv_s1.weight`` ...), and
children are created in the reference's order, so ``state_dict()`` keys, shapes and the seed-0
initialisation interchange.  The modules only own parameters; ``emit`` lowers the graph:

* REBNCONV (u2net.py:6-17)  -> direct/implicit-GEMM 3x3 convolution with the BatchNorm statistics
  in its epilogue, then one BN-apply+ReLU pass that also writes the 2x2 max-pool and, for the last
  convolution of a block, adds the residual (``hx1d + hxin``).
* ``torch.cat((up, skip), 1)`` is never materialised: each decoder convolution reads ONE buffer
  whose halves were written in place by the skip's producer and by the bilinear-resize kernel.
* the six side heads, their resize to full resolution and the 1x1 fuse conv run on NCHW fp32 logit
  planes (``Engine.u2net_heads``).

``MaxPool2d(2, 2, ceil_mode=True)`` (u2net.py:30, 221-229): the fused pool writes ceil(H/2) x ceil(W/2)
with border windows clipped, so any input size works as in the reference.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from ..engine import Engine
from ..graph import HipModule
from ..ops import Act


class REBNCONV(nn.Module):
    """Conv3x3(dilation=d, padding=d) -> BN -> ReLU (reference: u2net.py:6-17)."""

    def __init__(self, in_ch: int = 3, out_ch: int = 3, dirate: int = 1):
        super().__init__()
        self.conv_s1 = nn.Conv2d(in_ch, out_ch, 3, padding=dirate, dilation=dirate)
        self.bn_s1 = nn.BatchNorm2d(out_ch)
        self.relu_s1 = nn.ReLU(inplace=True)

    def emit(self, eng: Engine, x: Act, **kw) -> Tuple[Act, Optional[Act]]:
        return eng.conv_bn_relu(x, self.conv_s1, self.bn_s1, **kw)


def _pool() -> nn.MaxPool2d:
    return nn.MaxPool2d(2, stride=2, ceil_mode=True)


def _half(n: int) -> int:
    return (n + 1) // 2   # output size of MaxPool2d(2, stride=2, ceil_mode=True)


class _RSU(nn.Module):
    """Residual U-block of height L (reference: RSU7/6/5/4, u2net.py:25-188)."""

    height = 0

    def __init__(self, in_ch: int = 3, mid_ch: int = 12, out_ch: int = 3):
        super().__init__()
        L = self.height
        self.rebnconvin = REBNCONV(in_ch, out_ch, dirate=1)
        self.rebnconv1 = REBNCONV(out_ch, mid_ch, dirate=1)
        self.pool1 = _pool()
        for i in range(2, L):
            setattr(self, f"rebnconv{i}", REBNCONV(mid_ch, mid_ch, dirate=1))
            if i < L - 1:
                setattr(self, f"pool{i}", _pool())
        setattr(self, f"rebnconv{L}", REBNCONV(mid_ch, mid_ch, dirate=2))
        for i in range(L - 1, 1, -1):
            setattr(self, f"rebnconv{i}d", REBNCONV(mid_ch * 2, mid_ch, dirate=1))
        self.rebnconv1d = REBNCONV(mid_ch * 2, out_ch, dirate=1)
        self.mid_ch = mid_ch

    def emit(self, eng: Engine, x: Act, *, out: Optional[Act] = None, pool: bool = False,
             im2col: bool = False) -> Tuple[Act, Optional[Act]]:
        L, mid = self.height, self.mid_ch
        hxin, _ = self.rebnconvin.emit(eng, x, im2col=im2col)
        N = hxin.N
        # encoder: level i (1..L-1) runs at (H >> (i-1)); its output is the right half of the
        # concat buffer the decoder convolution `rebnconv{i}d` reads
        cats: List[Tuple[Act, Act]] = []
        cur = hxin
        h, w = hxin.H, hxin.W
        for i in range(1, L):
            full, (up_slot, skip_slot) = eng.new_cat(N, h, w, (mid, mid))
            cats.append((full, up_slot))
            last = i == L - 1
            skip, pooled = getattr(self, f"rebnconv{i}").emit(eng, cur, out=skip_slot, pool=not last, pool_ceil=True)
            cur = skip if last else pooled
            if not last:
                h, w = _half(h), _half(w)
        # bottom: dilation 2 at the coarsest resolution, straight into the left half
        full, up_slot = cats[L - 2]
        getattr(self, f"rebnconv{L}").emit(eng, cur, out=up_slot)
        # decoder
        for i in range(L - 1, 0, -1):
            full, _ = cats[i - 1]
            if i > 1:
                tmp, _ = getattr(self, f"rebnconv{i}d").emit(eng, full)
                eng.resize_bilinear(tmp, cats[i - 2][1])
            else:
                return self.rebnconv1d.emit(eng, full, out=out, pool=pool, residual=hxin, pool_ceil=True)
        raise AssertionError("unreachable")


class RSU7(_RSU):
    height = 7


class RSU6(_RSU):
    height = 6


class RSU5(_RSU):
    height = 5


class RSU4(_RSU):
    height = 4


class RSU4F(nn.Module):
    """Dilated residual block without resampling (reference: u2net.py:191-213)."""

    def __init__(self, in_ch: int = 3, mid_ch: int = 12, out_ch: int = 3):
        super().__init__()
        self.rebnconvin = REBNCONV(in_ch, out_ch, dirate=1)
        self.rebnconv1 = REBNCONV(out_ch, mid_ch, dirate=1)
        self.rebnconv2 = REBNCONV(mid_ch, mid_ch, dirate=2)
        self.rebnconv3 = REBNCONV(mid_ch, mid_ch, dirate=4)
        self.rebnconv4 = REBNCONV(mid_ch, mid_ch, dirate=8)
        self.rebnconv3d = REBNCONV(mid_ch * 2, mid_ch, dirate=4)
        self.rebnconv2d = REBNCONV(mid_ch * 2, mid_ch, dirate=2)
        self.rebnconv1d = REBNCONV(mid_ch * 2, out_ch, dirate=1)
        self.mid_ch = mid_ch

    def emit(self, eng: Engine, x: Act, *, out: Optional[Act] = None, pool: bool = False,
             im2col: bool = False) -> Tuple[Act, Optional[Act]]:
        mid = self.mid_ch
        hxin, _ = self.rebnconvin.emit(eng, x, im2col=im2col)
        N, H, W = hxin.N, hxin.H, hxin.W
        c1, (d2_slot, h1_slot) = eng.new_cat(N, H, W, (mid, mid))   # cat((hx2d, hx1), 1)
        c2, (d3_slot, h2_slot) = eng.new_cat(N, H, W, (mid, mid))   # cat((hx3d, hx2), 1)
        c3, (h4_slot, h3_slot) = eng.new_cat(N, H, W, (mid, mid))   # cat((hx4, hx3), 1)
        hx1, _ = self.rebnconv1.emit(eng, hxin, out=h1_slot)
        hx2, _ = self.rebnconv2.emit(eng, hx1, out=h2_slot)
        hx3, _ = self.rebnconv3.emit(eng, hx2, out=h3_slot)
        self.rebnconv4.emit(eng, hx3, out=h4_slot)
        self.rebnconv3d.emit(eng, c3, out=d3_slot)
        self.rebnconv2d.emit(eng, c2, out=d2_slot)
        return self.rebnconv1d.emit(eng, c1, out=out, pool=pool, residual=hxin, pool_ceil=True)


class _U2NetBase(HipModule):
    """Six encoder + five decoder RSU stages, six side heads, one fuse conv (u2net.py:216-298)."""

    #: (block class, mid_ch, out_ch) of stage1..stage6 and (block, in_ch, mid_ch, out_ch) of stage5d..stage1d
    enc_cfg: Tuple = ()
    dec_cfg: Tuple = ()

    def __init__(self, in_ch: int = 3, out_ch: int = 1):
        super().__init__()
        cin = in_ch
        for i, (blk, mid, cout) in enumerate(self.enc_cfg, start=1):
            setattr(self, f"stage{i}", blk(cin, mid, cout))
            if i < 6:
                setattr(self, f"pool{i}{i + 1}", _pool())
            cin = cout
        for lvl, (blk, cin_d, mid, cout) in zip((5, 4, 3, 2, 1), self.dec_cfg):
            setattr(self, f"stage{lvl}d", blk(cin_d, mid, cout))
        side_in = [self.dec_cfg[4][3], self.dec_cfg[3][3], self.dec_cfg[2][3], self.dec_cfg[1][3],
                   self.dec_cfg[0][3], self.enc_cfg[5][2]]
        for k, c in enumerate(side_in, start=1):
            setattr(self, f"side{k}", nn.Conv2d(c, out_ch, 3, padding=1))
        self.outconv = nn.Conv2d(6 * out_ch, out_ch, 1)

    def emit(self, eng: Engine, x: torch.Tensor):
        N, _, H, W = x.shape
        enc_out = [c[2] for c in self.enc_cfg]
        # decoder stage `lvl`d reads cat((upsampled deeper map, encoder map lvl), 1)
        cats: Dict[int, Tuple[Act, Act]] = {}
        cur = eng.input_im2col(x)
        deepest: Optional[Act] = None
        h, w = H, W
        for i in range(1, 7):
            stage = getattr(self, f"stage{i}")
            if i < 6:
                up_c = enc_out[5] if i == 5 else self.dec_cfg[4 - i][3]   # channels of the map resized into this level
                full, (up_slot, skip_slot) = eng.new_cat(N, h, w, (up_c, enc_out[i - 1]))
                cats[i] = (full, up_slot)
                _, cur = stage.emit(eng, cur, out=skip_slot, pool=True, im2col=(i == 1))
                h, w = _half(h), _half(w)
            else:
                deepest, _ = stage.emit(eng, cur)
        dec: Dict[int, Act] = {6: deepest}
        cur = deepest
        for lvl in (5, 4, 3, 2, 1):
            full, up_slot = cats[lvl]
            eng.resize_bilinear(cur, up_slot)
            cur, _ = getattr(self, f"stage{lvl}d").emit(eng, full)
            dec[lvl] = cur
        feats = [dec[k] for k in range(1, 7)]
        sides = [getattr(self, f"side{k}") for k in range(1, 7)]
        return tuple(eng.u2net_heads(feats, sides, self.outconv))

    def wrap_outputs(self, outs):
        keys = ("main", "side1", "side2", "side3", "side4", "side5", "side6")
        return dict(zip(keys, outs))


class U2NET(_U2NetBase):
    """176 MB U^2-Net (reference: u2net.py:216-298)."""
    enc_cfg = ((RSU7, 32, 64), (RSU6, 32, 128), (RSU5, 64, 256), (RSU4, 128, 512), (RSU4F, 256, 512),
               (RSU4F, 256, 512))
    dec_cfg = ((RSU4F, 1024, 256, 512), (RSU4, 1024, 128, 256), (RSU5, 512, 64, 128), (RSU6, 256, 32, 64),
               (RSU7, 128, 16, 64))


class U2NETP(_U2NetBase):
    """4.7 MB U^2-NetP (reference: u2net.py:301-381)."""
    enc_cfg = ((RSU7, 16, 64), (RSU6, 16, 64), (RSU5, 16, 64), (RSU4, 16, 64), (RSU4F, 16, 64), (RSU4F, 16, 64))
    dec_cfg = ((RSU4F, 128, 16, 64), (RSU4, 128, 16, 64), (RSU5, 128, 16, 64), (RSU6, 128, 16, 64),
               (RSU7, 128, 16, 64))
