"""GPU: the weight gradient whose x operand is read THROUGH the BatchNorm + ReLU in front of the layer (uz_wgrad_xf,
round 5; reference: autograd's weight gradient of the second nn.Conv2d of DoubleConv, common_layers.py:28-33, entered from
loss.backward(), training_loop.py:119).  R holds the raw output of the first convolution; the row-walk kernel's loader
waves form relu(R * scale + shift) inside the LDS ring.

Held against (a) uz_wgrad on the activation uz_bn_relu_apply materialises -- the SAME kernel on the same operands after the
transform: slabs and result equal BIT FOR BIT -- and (b) autograd's weight gradient of F.conv2d.  Every strip width
(16, 32, 64, several strips per row), segment starts inside a workgroup's range, channel tails (a 40-channel x), windows
of NaN-poisoned buffers, shift > 0 everywhere (a transformed zero pad would count), the nearest-upsampled x operand."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from unet_zoo_amd import _lib as L
from unet_zoo_amd import ops
from unet_zoo_amd.ops import Act, act_from_nchw

DEV = "cuda"
dt = torch.bfloat16


def rnd(t):
    return t.to(dt).float()


CASES = [
    # N, H, W, Ci (dy channels), Cj (x channels), window, upsampled
    (16, 64, 64, 64, 64, 0, False),      # one strip per row, KW = 64
    (2, 128, 256, 64, 64, 0, False),     # four strips per row
    (4, 32, 32, 128, 128, 0, False),     # KW = 32, two rows per step, two channel tiles each way
    (8, 16, 16, 256, 192, 0, False),     # KW = 16, four rows per step
    (3, 64, 128, 72, 40, 16, False),     # channel tails on both operands, windows of NaN-poisoned buffers
    (5, 64, 64, 64, 64, 0, False),       # 5 images: segment starts inside workgroup ranges
    (2, 64, 128, 64, 64, 0, True),       # x at half resolution (UpConvBlock)
]


@pytest.mark.parametrize("N,H,W,Ci,Cj,win,ups", CASES)
def test_wgrad_xf_equals_apply_then_wgrad(N, H, W, Ci, Cj, win, ups):
    g = torch.Generator().manual_seed(11)
    Hr, Wr = (H // 2, W // 2) if ups else (H, W)
    x = rnd(torch.randn(N, Cj, Hr, Wr, generator=g))
    dy = rnd(torch.randn(N, Ci, H, W, generator=g))
    scale = (torch.rand(Cj, generator=g) + 0.5) * torch.where(torch.rand(Cj, generator=g) < 0.2, -1.0, 1.0)
    shift = torch.rand(Cj, generator=g) * 0.8 + 0.3
    xa, da = act_from_nchw(x.to(DEV), dt), act_from_nchw(dy.to(DEV), dt)
    if win:
        wide = torch.full((xa.P, Cj + 2 * win), float("nan"), dtype=dt, device=DEV)
        wide[:, win:win + Cj] = xa.buf
        xa = Act(wide, win, Cj, N, Hr, Wr)
        wide2 = torch.full((da.P, Ci + 2 * win), float("nan"), dtype=dt, device=DEV)
        wide2[:, win:win + Ci] = da.buf
        da = Act(wide2, win, Ci, N, H, W)
    sc, sh = scale.to(DEV), shift.to(DEV)
    tm = L.TAPS_CONV_UP2 if ups else L.TAPS_CONV
    assert ops.wgrad_xform_supported(da, xa, 9, taps_mode=tm)
    a = ops.new_act(N, Hr, Wr, Cj, dt, torch.device(DEV), False)
    ops.bn_relu_apply(xa, sc, sh, a)
    d = L.WgradDesc(L.dtype_code(dt), N, H, W, Hr, Wr, Ci, da.ld, Cj, a.ld, 9, tm, 1)
    assert ops.wgrad_kernel_name(d) == "wgrad9_bf16_64x64_rowwalk"
    w0 = ops.wgrad(da, a, (Ci, Cj, 3, 3), ntaps=9, taps_mode=tm)
    w1 = ops.wgrad(da, xa, (Ci, Cj, 3, 3), ntaps=9, taps_mode=tm, xform=(sc, sh))
    torch.cuda.synchronize()
    assert torch.isfinite(w1).all()
    assert torch.equal(w0, w1), f"max diff {(w0 - w1).abs().max().item()} of {w0.abs().max().item()}"
    # autograd on the activation as the stand-alone pass stores it
    act = rnd(torch.relu(torch.addcmul(shift.view(1, -1, 1, 1), x, scale.view(1, -1, 1, 1))))
    xin = F.interpolate(act, scale_factor=2, mode="nearest") if ups else act
    wt = torch.zeros(Ci, Cj, 3, 3, requires_grad=True)
    F.conv2d(xin, wt, padding=1).backward(dy)
    err = (w1.cpu().double() - wt.grad.double()).abs().max() / wt.grad.double().abs().max()
    assert err < 2e-3, err
    # repeatable; R untouched
    w2 = ops.wgrad(da, xa, (Ci, Cj, 3, 3), ntaps=9, taps_mode=tm, xform=(sc, sh))
    assert torch.equal(w1, w2)
    assert torch.equal(xa.dense().cpu(), x)


def test_wgrad_xf_refuses_what_it_cannot_take():
    from ctypes import byref
    lib = L.load()
    d = L.WgradDesc(L.dtype_code(dt), 2, 24, 24, 24, 24, 64, 64, 64, 64, 9, L.TAPS_CONV, 1)    # W = 24: not a strip width
    assert lib.uz_wgrad_xf_supported(byref(d)) == 0
    d1 = L.WgradDesc(L.dtype_code(dt), 2, 64, 64, 64, 64, 64, 64, 64, 64, 1, L.TAPS_CONV, 1)   # one tap
    assert lib.uz_wgrad_xf_supported(byref(d1)) == 0
