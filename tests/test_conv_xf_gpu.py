"""GPU: the 3x3 convolution that reads its input THROUGH the BatchNorm + ReLU in front of it (uz_conv_igemm_xf, round 5):
x is the raw output of the preceding convolution, the kernel forms relu(x * scale + shift) on the halo patch inside LDS,
so the middle tensor of a DoubleConv (reference: unet_zoo/models/common_layers.py:28-33, Conv -> BN -> ReLU -> Conv) is
never materialised.

Held against (a) the two-launch form it replaces -- uz_bn_relu_apply into a tensor, then uz_conv_igemm on the SAME
ping-pong configuration: outputs and BatchNorm partial sums must be equal BIT FOR BIT (same operands after the transform,
same MFMA order) -- and (b) F.conv2d on the bf16-rounded activation.  Cases: several tiles per workgroup and one, ragged
tiles (zero padding must stay zero AFTER the affine map: shift > 0 everywhere makes a wrong halo visible), one and several
channel slabs, a channel window of a NaN-poisoned buffer, the nearest-upsampled input."""
import pytest
import torch
import torch.nn.functional as F
from ctypes import byref

pytestmark = pytest.mark.gpu

from unet_zoo_amd import _lib as L
from unet_zoo_amd import ops
from unet_zoo_amd.ops import Act, act_from_nchw

DEV = "cuda"
dt = torch.bfloat16


def rnd(t):
    return t.to(dt).float()


def relerr(a, b):
    return ((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30)).item()


CASES = [
    # configuration, N, H, W, Cin, Cout, xwin, ups
    ("pp512x64", 16, 64, 64, 64, 64, 0, False),     # 128 tiles, one per workgroup, two slabs
    ("pp512x64", 3, 128, 256, 128, 64, 0, False),   # four slabs
    ("pp512x64", 2, 120, 250, 64, 40, 32, False),   # ragged both ways, channel tail, input window of a NaN-poisoned buffer
    ("pp512x64", 9, 128, 256, 32, 64, 0, False),    # 576 tiles: up to three per workgroup, ONE slab per tile (first == last)
    ("pp512x64", 5, 128, 256, 64, 64, 0, False),    # 320 tiles: some workgroups two, the rest one
    ("pp512x64", 3, 128, 256, 64, 64, 0, True),     # nearest x2 upsampled input
]


@pytest.mark.parametrize("cfg,N,H,W,Cin,Cout,win,ups", CASES)
def test_conv_xf_equals_apply_then_conv(cfg, N, H, W, Cin, Cout, win, ups):
    g = torch.Generator().manual_seed(5)
    Hi, Wi = (H // 2, W // 2) if ups else (H, W)
    x = rnd(torch.randn(N, Cin, Hi, Wi, generator=g))
    w = rnd(torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05)
    b = torch.randn(Cout, generator=g)
    scale = (torch.rand(Cin, generator=g) + 0.5) * torch.where(torch.rand(Cin, generator=g) < 0.2, -1.0, 1.0)
    shift = torch.rand(Cin, generator=g) * 0.8 + 0.3          # > 0: relu(shift) != 0, a transformed zero pad would show
    xa = act_from_nchw(x.to(DEV), dt)
    if win:
        wide = torch.full((xa.P, Cin + 2 * win), float("nan"), dtype=dt, device=DEV)
        wide[:, win:win + Cin] = xa.buf
        xa = Act(wide, win, Cin, N, Hi, Wi)
    wp = ops.pack_weights(w.to(DEV), L.PACK_CONV_FWD, dt)
    tm = L.TAPS_CONV_UP2 if ups else L.TAPS_CONV
    sc, sh = scale.to(DEV), shift.to(DEV)
    assert ops.conv_xform_supported(xa, Cout, Cout + 16, upsample=ups)

    # (a) the two launches it replaces
    a = ops.new_act(N, Hi, Wi, Cin, dt, torch.device(DEV), False)
    ops.bn_relu_apply(xa, sc, sh, a)
    ywide0 = torch.full((N * H * W, Cout + 16), 7.0, dtype=dt, device=DEV)
    y0 = Act(ywide0, 8, Cout, N, H, W)
    d = L.ConvDesc(L.dtype_code(dt), N, H, W, Hi, Wi, Cin, a.ld, Cout, y0.ld, 9, tm, 1, L.STORE_PLAIN, 0, 0, 0)
    assert ops.conv_kernel_name(d) == f"conv3x3_{cfg}_bf16" + ("_up2" if ups else "")
    st0 = ops.conv_igemm(a, wp, b.to(DEV), y0, ntaps=9, want_stats=True, taps_mode=tm)

    # the fused form
    ywide1 = torch.full((N * H * W, Cout + 16), 7.0, dtype=dt, device=DEV)
    y1 = Act(ywide1, 8, Cout, N, H, W)
    st1 = ops.conv_igemm(xa, wp, b.to(DEV), y1, ntaps=9, want_stats=True, taps_mode=tm, xform=(sc, sh))
    torch.cuda.synchronize()
    assert torch.isfinite(y1.dense()).all()
    assert torch.equal(ywide0, ywide1), f"max diff {(ywide0.float() - ywide1.float()).abs().max().item()}"
    assert torch.equal(st0, st1)

    # (b) the operator the reference calls, on the activation as the stand-alone pass would have stored it
    act = rnd(torch.relu(torch.addcmul(shift.view(1, -1, 1, 1), x, scale.view(1, -1, 1, 1))))
    xin = F.interpolate(act, scale_factor=2, mode="nearest") if ups else act
    ref = F.conv2d(xin, w, b, padding=1)
    assert relerr(y1.dense().cpu(), ref) < 2e-2

    # repeatable; the raw input is left as it was
    ywide2 = torch.full((N * H * W, Cout + 16), 7.0, dtype=dt, device=DEV)
    y2 = Act(ywide2, 8, Cout, N, H, W)
    st2 = ops.conv_igemm(xa, wp, b.to(DEV), y2, ntaps=9, want_stats=True, taps_mode=tm, xform=(sc, sh))
    assert torch.equal(ywide1, ywide2) and torch.equal(st1, st2)
    assert torch.equal(xa.dense().cpu(), x)


def test_conv_xf_refuses_what_it_cannot_take():
    lib = L.load()
    # 128 output channels: the 512 x 128 configuration's LDS image leaves no room for the table
    d = L.ConvDesc(L.dtype_code(dt), 4, 128, 128, 128, 128, 128, 128, 128, 128, 9, L.TAPS_CONV, 1, L.STORE_PLAIN, 0, 0, 0)
    if not lib.uz_conv_igemm_xf_supported(byref(d)):
        x = torch.zeros(4 * 128 * 128, 128, dtype=dt, device=DEV)
        w = torch.zeros(128, 9 * 128, dtype=dt, device=DEV)
        v = torch.zeros(128, device=DEV)
        rc = lib.uz_conv_igemm_xf(byref(d), x.data_ptr(), v.data_ptr(), v.data_ptr(), w.data_ptr(), None, x.data_ptr(), None, None)
        assert rc == -2   # UZ_ENOTIMPL
    # fp32 run mode: never
    d32 = L.ConvDesc(L.dtype_code(torch.float32), 4, 64, 64, 64, 64, 64, 64, 64, 64, 9, L.TAPS_CONV, 1, L.STORE_PLAIN, 0, 0, 0)
    assert lib.uz_conv_igemm_xf_supported(byref(d32)) == 0
