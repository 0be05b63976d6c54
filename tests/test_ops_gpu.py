"""GPU: every C-ABI kernel against the plain-PyTorch fp32 CPU restatement of the same op
(torch.nn.functional — what the reference itself calls), on seeded inputs, ragged sizes included.
fp32 run dtype must agree to 2e-5 of the tensor's max (measured ~1e-6; the north-star tolerance is
1e-3 relative); bf16 is checked against the same fp32 math fed bf16-rounded operands."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from unet_zoo_amd import _lib as L
from unet_zoo_amd import ops
from unet_zoo_amd.ops import Act, act_from_nchw

DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16]


def tol(dt):
    return 2e-5 if dt == torch.float32 else 2e-2


def rnd(dt, t):
    """round through the run dtype so both sides see identical operands"""
    return t.to(dt).float()


def relerr(a, b):
    return ((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30)).item()


def bk(dt):
    return 32 if dt == torch.float32 else 64


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("N,H,W,Cin,Cout,dil", [
    (2, 12, 20, 64, 64, 1),      # M = 480: ragged last tile
    (1, 16, 16, 128, 192, 1),    # Nout tail inside a 128-wide tile
    (3, 9, 7, 64, 128, 2),       # odd sizes, dilation 2
    (1, 33, 5, 64, 32, 4),       # dilation larger than W
    (2, 11, 40, 64, 64, 1),      # W >= 32: 8x32 patches, ragged in both directions
    (1, 40, 72, 128, 256, 1),    # 8x32 patches, two channel slabs, two N tiles
])
def test_conv3x3_fwd_bias_stats(dt, N, H, W, Cin, Cout, dil):
    g = torch.Generator().manual_seed(0)
    x = rnd(dt, torch.randn(N, Cin, H, W, generator=g))
    w = rnd(dt, torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05)
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x, w, b, padding=dil, dilation=dil)
    xa = act_from_nchw(x.to(DEV), dt)
    wp = ops.pack_weights(w.to(DEV), L.PACK_CONV_FWD, dt)
    y = ops.new_act(N, H, W, Cout, dt, DEV)
    stats = ops.conv_igemm(xa, wp, b.to(DEV), y, ntaps=9, dil=dil, want_stats=True)
    got = y.dense().cpu()
    assert relerr(got, ref) < tol(dt)
    # statistics are of the STORED (rounded) value
    s = stats.double().sum(0).cpu()
    stored = got.double()
    assert relerr(s[0], stored.sum((0, 2, 3))) < 1e-4 + tol(dt) * 0.1
    assert relerr(s[1], (stored ** 2).sum((0, 2, 3))) < 1e-4


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("N,H,W,Cin,Cout,dil", [(2, 12, 20, 64, 128, 1), (1, 10, 14, 128, 64, 2)])
def test_conv3x3_dgrad_and_wgrad(dt, N, H, W, Cin, Cout, dil):
    g = torch.Generator().manual_seed(1)
    x = rnd(dt, torch.randn(N, Cin, H, W, generator=g)).requires_grad_(True)
    w = rnd(dt, torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05).requires_grad_(True)
    dy = rnd(dt, torch.randn(N, Cout, H, W, generator=g))
    F.conv2d(x, w, None, padding=dil, dilation=dil).backward(dy)
    dya = act_from_nchw(dy.to(DEV), dt)
    xa = act_from_nchw(x.detach().to(DEV), dt)
    wd = ops.pack_weights(w.detach().to(DEV), L.PACK_CONV_DGRAD, dt)
    dx = ops.new_act(N, H, W, Cin, dt, DEV)
    ops.conv_igemm(dya, wd, None, dx, ntaps=9, dil=dil)
    assert relerr(dx.dense().cpu(), x.grad) < tol(dt)
    dw = ops.wgrad(dya, xa, (Cout, Cin, 3, 3), ntaps=9, dil=dil)
    assert relerr(dw.cpu(), w.grad) < tol(dt)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("N,H,W,Cin,Cout", [
    (2, 16, 64, 64, 64),     # 64x64 channel tile, all nine taps per workgroup, one row per K-step
    (1, 8, 16, 128, 128),    # 128x128 tile, W=16: four image rows per K-step
    (1, 4, 32, 256, 128),    # W=32: two rows per K-step, several channel tiles
    (2, 6, 128, 128, 64),    # W=128: two K-steps per row; Cout=64 -> 64-tiles
    (3, 2, 64, 64, 192),     # ragged split count
])
def test_wgrad3x3_lds_dma_kernel_shapes(dt, N, H, W, Cin, Cout):
    """shapes the LDS-DMA 3x3 weight-gradient kernel takes in bf16 (fp32 runs the generic one),
    with both operands living in wider buffers (channel windows)"""
    g = torch.Generator().manual_seed(9)
    x = rnd(dt, torch.randn(N, Cin, H, W, generator=g))
    dy = rnd(dt, torch.randn(N, Cout, H, W, generator=g))
    w = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    F.conv2d(x, w, None, padding=1).backward(dy)
    P = N * H * W
    xw = torch.full((P, Cin + 64), 3.0, dtype=dt, device=DEV)
    xw[:, 64:] = act_from_nchw(x.to(DEV), dt).buf
    dw_ = torch.full((P, 2 * Cout), -2.0, dtype=dt, device=DEV)
    dw_[:, :Cout] = act_from_nchw(dy.to(DEV), dt).buf
    dwg = ops.wgrad(Act(dw_, 0, Cout, N, H, W), Act(xw, 64, Cin, N, H, W), (Cout, Cin, 3, 3), ntaps=9)
    assert relerr(dwg.cpu(), w.grad) < tol(dt)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("N,H,W,Ci,Cj", [(2, 8, 64, 64, 64), (1, 16, 16, 128, 128), (1, 6, 10, 64, 64),
                                         (2, 4, 64, 32, 64), (1, 8, 32, 96, 40),
                                         # token maps of swin_unet_v2 at 224 x 224 (flat walk, ragged last row of 64)
                                         (3, 7, 7, 768, 768), (2, 14, 14, 384, 1152), (2, 56, 56, 96, 288),
                                         (1, 5, 5, 96, 96)])
def test_wgrad_one_tap(dt, N, H, W, Ci, Cj):
    """ntaps = 1 (1x1 convolution / the im2col'd first layer): out[i][j] = sum_p L[p,i] R[p,j]"""
    g = torch.Generator().manual_seed(10)
    a = rnd(dt, torch.randn(N, Ci, H, W, generator=g))
    b = rnd(dt, torch.randn(N, Cj, H, W, generator=g))
    ref = torch.einsum("nihw,njhw->ij", a.double(), b.double()).float()
    got = ops.wgrad(act_from_nchw(a.to(DEV), dt), act_from_nchw(b.to(DEV), dt), (Ci, Cj), ntaps=1)
    assert relerr(got.cpu(), ref) < tol(dt)


@pytest.mark.parametrize("dt", DTYPES)
def test_wgrad_split_k_large_pixel_count(dt):
    """many pixels, few channels: the split-K / atomic path"""
    g = torch.Generator().manual_seed(2)
    N, H, W, C = 2, 64, 96, 64
    x = rnd(dt, torch.randn(N, C, H, W, generator=g))
    dy = rnd(dt, torch.randn(N, C, H, W, generator=g))
    w = torch.zeros(C, C, 3, 3, requires_grad=True)
    F.conv2d(x, w, None, padding=1).backward(dy)
    dw = ops.wgrad(act_from_nchw(dy.to(DEV), dt), act_from_nchw(x.to(DEV), dt), (C, C, 3, 3), ntaps=9)
    assert relerr(dw.cpu(), w.grad) < tol(dt)


@pytest.mark.parametrize("dt", DTYPES)
def test_conv_reads_and_writes_channel_windows(dt):
    """input and output may be channel slices of wider buffers (virtual concat)"""
    g = torch.Generator().manual_seed(3)
    N, H, W, Cin, Cout = 1, 8, 8, 64, 64
    x = rnd(dt, torch.randn(N, Cin, H, W, generator=g))
    w = rnd(dt, torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05)
    ref = F.conv2d(x, w, None, padding=1)
    wide_in = torch.full((N * H * W, 3 * Cin), 7.0, dtype=dt, device=DEV)
    wide_in[:, Cin:2 * Cin] = act_from_nchw(x.to(DEV), dt).buf
    xin = Act(wide_in, Cin, Cin, N, H, W)
    wide_out = torch.full((N * H * W, 2 * Cout), -3.0, dtype=dt, device=DEV)
    yout = Act(wide_out, Cout, Cout, N, H, W)
    ops.conv_igemm(xin, ops.pack_weights(w.to(DEV), L.PACK_CONV_FWD, dt), None, yout, ntaps=9)
    assert relerr(yout.dense().cpu(), ref) < tol(dt)
    assert torch.all(wide_out[:, :Cout].float() == -3.0)  # neighbours untouched


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 6, 10, 128, 64), (1, 4, 4, 256, 128),
                                            # map widths the LDS-DMA weight-gradient kernel takes (gather mode)
                                            (2, 16, 16, 256, 128), (1, 8, 32, 128, 64), (1, 4, 64, 64, 96),
                                            (2, 2, 16, 384, 192)])
def test_conv_transpose2x2_fwd_dgrad_wgrad(dt, N, H, W, Cin, Cout):
    g = torch.Generator().manual_seed(4)
    x = rnd(dt, torch.randn(N, Cin, H, W, generator=g)).requires_grad_(True)
    w = rnd(dt, torch.randn(Cin, Cout, 2, 2, generator=g) * 0.05).requires_grad_(True)
    b = torch.randn(Cout, generator=g).requires_grad_(True)
    dy = rnd(dt, torch.randn(N, Cout, 2 * H, 2 * W, generator=g))
    ref = F.conv_transpose2d(x, w, b, stride=2)
    ref.backward(dy)
    xa = act_from_nchw(x.detach().to(DEV), dt)
    y = ops.new_act(N, 2 * H, 2 * W, Cout, dt, DEV)
    wp = ops.pack_weights(w.detach().to(DEV), L.PACK_CONVT_FWD, dt)
    ops.conv_igemm(xa, wp, b.detach().to(DEV).repeat(4), y, ntaps=1, store_mode=L.STORE_SHUFFLE2X2,
                   nout=4 * Cout, co=Cout)
    assert relerr(y.dense().cpu(), ref.detach()) < tol(dt)
    dya = act_from_nchw(dy.to(DEV), dt)
    dx = ops.new_act(N, H, W, Cin, dt, DEV)
    ops.conv_igemm(dya, ops.pack_weights(w.detach().to(DEV), L.PACK_CONVT_DGRAD, dt), None, dx, ntaps=4,
                   taps_mode=L.TAPS_GATHER2X2)
    assert relerr(dx.dense().cpu(), x.grad) < tol(dt)
    dw = ops.wgrad(xa, dya, (Cin, Cout, 2, 2), ntaps=4, taps_mode=L.TAPS_GATHER2X2)
    assert relerr(dw.cpu(), w.grad) < tol(dt)
    assert relerr(ops.colsum(dya).cpu(), b.grad) < tol(dt)


@pytest.mark.parametrize("dt", DTYPES)
def test_im2col_first_conv(dt):
    g = torch.Generator().manual_seed(5)
    N, C, H, W, Cout = 2, 3, 10, 12, 64
    x = torch.randn(N, C, H, W, generator=g)
    w = rnd(dt, torch.randn(Cout, C, 3, 3, generator=g) * 0.2)
    ref = F.conv2d(rnd(dt, x), w, None, padding=1)
    a = ops.im2col3x3_nchw(x.to(DEV), bk(dt), dt)
    wp = ops.pack_weights(w.to(DEV), L.PACK_IM2COL, dt, bk(dt))
    y = ops.new_act(N, H, W, Cout, dt, DEV)
    ops.conv_igemm(a, wp, None, y, ntaps=1)
    assert relerr(y.dense().cpu(), ref) < tol(dt)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("pool", [False, True])
@pytest.mark.parametrize("N,H,W,C", [(2, 8, 12, 64), (1, 6, 6, 256)])
def test_bn_relu_pool_forward_backward(dt, pool, N, H, W, C):
    """train-mode BatchNorm2d + ReLU (+ MaxPool2d(2,2)) forward, running stats and backward with a
    direct gradient, a second direct gradient and a pooled gradient"""
    g = torch.Generator().manual_seed(6)
    y = rnd(dt, torch.randn(N, C, H, W, generator=g) * 2 + 0.5).requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.randn(C, generator=g) * 0.1).requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    act_ref = F.relu(F.batch_norm(y, rm, rv, gamma, beta, True, 0.1, 1e-5))
    g0 = rnd(dt, torch.randn(N, C, H, W, generator=g))
    g1 = rnd(dt, torch.randn(N, C, H, W, generator=g))
    loss = (act_ref * (g0 + g1)).sum()
    if pool:
        p_ref = F.max_pool2d(act_ref, 2, 2)
        gp = rnd(dt, torch.randn(N, C, H // 2, W // 2, generator=g))
        loss = loss + (p_ref * gp).sum()
    loss.backward()

    ya = act_from_nchw(y.detach().to(DEV), dt)
    # statistics exactly as the conv epilogue would deliver them: one partial row
    yd = ya.buf.double()
    stats = torch.stack([yd.sum(0), (yd ** 2).sum(0)]).float().reshape(1, 2, C)
    rm_d, rv_d = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    vec = ops.bn_finalize(stats, N * H * W, gamma.detach().to(DEV), beta.detach().to(DEV), 1e-5, 0.1, rm_d, rv_d)
    assert relerr(rm_d.cpu(), rm) < 1e-5 and relerr(rv_d.cpu(), rv) < 1e-5
    act = ops.new_act(N, H, W, C, dt, DEV)
    pooled = ops.new_act(N, H // 2, W // 2, C, dt, DEV) if pool else None
    ops.bn_relu_apply(ya, vec[0], vec[1], act, pooled)
    assert relerr(act.dense().cpu(), act_ref.detach()) < tol(dt)
    if pool:
        assert relerr(pooled.dense().cpu(), p_ref.detach()) < tol(dt)

    sums = torch.zeros(2, C, dtype=torch.float64, device=DEV)
    dy = ops.new_act(N, H, W, C, dt, DEV)
    dgb = torch.empty(2, C, device=DEV)
    ops.bn_relu_bwd(ya, vec, act_from_nchw(g0.to(DEV), dt), act_from_nchw(g1.to(DEV), dt),
                    act_from_nchw(gp.to(DEV), dt) if pool else None, sums, dy, dgb[0], dgb[1])
    t = tol(dt) * (5 if dt == torch.bfloat16 else 1)
    assert relerr(dy.dense().cpu(), y.grad) < t
    assert relerr(dgb[0].cpu(), gamma.grad) < t
    assert relerr(dgb[1].cpu(), beta.grad) < t


def test_bn_eval_scale():
    g = torch.Generator().manual_seed(7)
    C = 96
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.1
    vec = ops.bn_eval_scale(gamma.to(DEV), beta.to(DEV), rm.to(DEV), rv.to(DEV), 1e-5)
    x = torch.randn(4, C, 3, 3, generator=g)
    ref = F.batch_norm(x, rm, rv, gamma, beta, False, 0.1, 1e-5)
    got = x * vec[0].cpu().view(1, C, 1, 1) + vec[1].cpu().view(1, C, 1, 1)
    assert relerr(got, ref) < 1e-5


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("K", [1, 3])
def test_outconv_fwd_bwd(dt, K):
    g = torch.Generator().manual_seed(8)
    N, H, W, C = 2, 9, 11, 64
    x = rnd(dt, torch.randn(N, C, H, W, generator=g)).requires_grad_(True)
    w = (torch.randn(K, C, 1, 1, generator=g) * 0.1).requires_grad_(True)
    b = torch.randn(K, generator=g).requires_grad_(True)
    gl = torch.randn(N, K, H, W, generator=g)
    ref = F.conv2d(x, w, b)
    ref.backward(gl)
    xa = act_from_nchw(x.detach().to(DEV), dt)
    wd = w.detach().reshape(K, C).to(DEV)
    out = ops.outconv_fwd(xa, wd, b.detach().to(DEV))
    assert relerr(out.cpu(), ref.detach()) < 1e-5
    dx = ops.new_act(N, H, W, C, dt, DEV)
    dw, db = ops.outconv_bwd(xa, wd, gl.to(DEV), dx)
    assert relerr(dx.dense().cpu(), x.grad) < tol(dt)
    assert relerr(dw.cpu(), w.grad.reshape(K, C)) < 1e-4
    assert relerr(db.cpu(), b.grad) < 1e-4


@pytest.mark.parametrize("N,H,W,Cin,Cout,win,ups", [
    (2, 16, 64, 64, 64, 0, False),      # the DoubleConv 64 -> 64 layer, several tiles per workgroup column
    (1, 50, 70, 64, 64, 0, False),      # ragged in both directions
    (3, 9, 40, 32, 64, 0, False),       # half a K slab (zero-filled by the descriptor range check)
    (2, 24, 33, 16, 32, 0, False),      # Cin 16, Nout 32: tails on both sides
    (1, 32, 32, 64, 128, 0, False),     # two workgroup columns of 64 output channels
    (2, 12, 20, 64, 64, 0, False),      # W < 32: 16x16 patches
    (1, 20, 36, 64, 192, 64, False),    # input is a channel window of a wider (NaN-poisoned) buffer
    (2, 16, 48, 64, 64, 0, True),       # nearest x2 upsampled input (UpConvBlock)
    (32, 32, 32, 64, 64, 0, False),     # more tiles than one wave of workgroups: the deferred epilogue of waves 4-7
])
def test_conv3x3_res64_register_resident_weights(N, H, W, Cin, Cout, win, ups):
    """bf16, Cin <= 64: conv3x3_res64_kernel (weights in registers, wave-local epilogue, waves 4-7 half a tile
    behind) against F.conv2d on the same rounded operands, with bias and the BatchNorm partial sums"""
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(5)
    Hi, Wi = (H // 2, W // 2) if ups else (H, W)
    x = rnd(dt, torch.randn(N, Cin, Hi, Wi, generator=g))
    w = rnd(dt, torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05)
    b = torch.randn(Cout, generator=g)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if ups else x
    ref = F.conv2d(xin, w, b, padding=1)
    xa = act_from_nchw(x.to(DEV), dt)
    if win:
        wide = torch.full((xa.P, Cin + 2 * win), float("nan"), dtype=dt, device=DEV)
        wide[:, win:win + Cin] = xa.buf
        xa = Act(wide, win, Cin, N, Hi, Wi)
    wp = ops.pack_weights(w.to(DEV), L.PACK_CONV_FWD, dt)
    y = ops.new_act(N, H, W, Cout, dt, DEV)
    y.buf.fill_(float("nan"))
    stats = ops.conv_igemm(xa, wp, b.to(DEV), y, ntaps=9, want_stats=True,
                           taps_mode=L.TAPS_CONV_UP2 if ups else L.TAPS_CONV)
    got = y.dense().cpu()
    assert torch.isfinite(got).all()
    assert relerr(got, ref) < tol(dt)
    s = stats.double().sum(0).cpu()
    assert relerr(s[0], got.double().sum((0, 2, 3))) < 3e-3
    assert relerr(s[1], (got.double() ** 2).sum((0, 2, 3))) < 1e-4
    # bitwise repeatable (fixed reduction order, no atomics)
    y2 = ops.new_act(N, H, W, Cout, dt, DEV)
    stats2 = ops.conv_igemm(xa, wp, b.to(DEV), y2, ntaps=9, want_stats=True,
                            taps_mode=L.TAPS_CONV_UP2 if ups else L.TAPS_CONV)
    assert torch.equal(y.buf, y2.buf) and torch.equal(stats, stats2)


@pytest.mark.parametrize("N,H,W,C,Cn", [
    (2, 24, 40, 64, 64),      # LDS-resident 64 -> 64 kernel, ragged 8x32 patches
    (1, 40, 72, 128, 128),    # streaming kernel, two channel slabs
    (3, 16, 16, 256, 128),    # 16x16 patches, 256 -> 128 (the gradient of a 128 -> 256 convolution)
    (2, 9, 20, 192, 72),      # channel tail inside the 128-wide tile, ragged rows
])
def test_bn_backward_reduction_in_the_input_gradient_epilogue(N, H, W, C, Cn):
    """uz_conv_igemm_bnred: the input-gradient convolution of the SECOND half of a DoubleConv (common_layers.py:31-33)
    leaves, beside the gradient g of the middle activation relu(bn(y)), the partial rows of sum(dz) and sum(dz * xhat)
    of that BatchNorm's backward.  The gradient must be bit-identical to the plain kernel's, the finalized sums (and
    with them dgamma, dbeta and dy) must agree with the stand-alone reduction pass to fp32 summation noise."""
    dt = torch.bfloat16
    gen = torch.Generator().manual_seed(11)
    dyb = act_from_nchw(rnd(dt, torch.randn(N, C, H, W, generator=gen)).to(DEV), dt)      # gradient of conv b's output
    wb = torch.randn(Cn, C, 3, 3, generator=gen) * 0.05                                     # dgrad weights: C -> Cn
    wp = ops.pack_weights(wb.to(DEV), L.PACK_CONV_FWD, dt)
    y = act_from_nchw(rnd(dt, torch.randn(N, Cn, H, W, generator=gen) * 2 + 0.3).to(DEV), dt)   # pre-BN output of conv a
    gamma = (torch.rand(Cn, generator=gen) + 0.5).to(DEV)
    beta = (torch.randn(Cn, generator=gen) * 0.2).to(DEV)
    yd = y.dense().double()
    stats = torch.stack([yd.sum((0, 2, 3)), (yd ** 2).sum((0, 2, 3))]).float().reshape(1, 2, Cn)
    vec = ops.bn_finalize(stats, N * H * W, gamma, beta, 1e-5, 0.1, torch.zeros(Cn, device=DEV), torch.ones(Cn, device=DEV))

    g_plain = ops.new_act(N, H, W, Cn, dt, DEV)
    ops.conv_igemm(dyb, wp, None, g_plain, ntaps=9)
    g_fused = ops.new_act(N, H, W, Cn, dt, DEV)
    part = ops.conv_igemm(dyb, wp, None, g_fused, ntaps=9, bnred=(y, vec))
    assert part is not None and part.shape[1:] == (2, Cn)
    assert torch.equal(g_plain.dense(), g_fused.dense())

    def bwd(partials):
        sums = torch.zeros(2, Cn, dtype=torch.float64, device=DEV)
        dx = ops.new_act(N, H, W, Cn, dt, DEV)
        dgb = torch.empty(2, Cn, device=DEV)
        ops.bn_relu_bwd(y, vec, g_fused, None, None, sums, dx, dgb[0], dgb[1], partials=partials)
        return sums.clone(), dx.dense().float(), dgb.clone()

    s0, dx0, dgb0 = bwd(None)
    s1, dx1, dgb1 = bwd(part)
    assert relerr(s1, s0) < 2e-5 and relerr(dgb1, dgb0) < 2e-5
    assert relerr(dx1, dx0) < 1e-2      # bf16 outputs: a last-place flip where the two sums differ in fp32 noise


@pytest.mark.parametrize("N,H,W,Cin,Co", [(2, 16, 24, 128, 64), (1, 8, 8, 1024, 512), (2, 9, 13, 256, 128)])
def test_bn_backward_reduction_in_the_conv_transpose_input_gradient(N, H, W, Cin, Co):
    """the same fusion on the LDS-DMA GEMM: the 2x2-gather input gradient of ConvTranspose2d(k2, s2)
    (common_layers.py:104) produces the gradient of the decoder block / bottleneck output below it"""
    dt = torch.bfloat16
    gen = torch.Generator().manual_seed(12)
    g_up = act_from_nchw(rnd(dt, torch.randn(N, Co, 2 * H, 2 * W, generator=gen)).to(DEV), dt)   # gradient of the ConvT output
    w = torch.randn(Cin, Co, 2, 2, generator=gen) * 0.05
    wp = ops.pack_weights(w.to(DEV), L.PACK_CONVT_DGRAD, dt)
    y = act_from_nchw(rnd(dt, torch.randn(N, Cin, H, W, generator=gen) * 2 + 0.3).to(DEV), dt)
    gamma = (torch.rand(Cin, generator=gen) + 0.5).to(DEV)
    beta = (torch.randn(Cin, generator=gen) * 0.2).to(DEV)
    yd = y.dense().double()
    stats = torch.stack([yd.sum((0, 2, 3)), (yd ** 2).sum((0, 2, 3))]).float().reshape(1, 2, Cin)
    vec = ops.bn_finalize(stats, N * H * W, gamma, beta, 1e-5, 0.1, torch.zeros(Cin, device=DEV), torch.ones(Cin, device=DEV))
    g_plain = ops.new_act(N, H, W, Cin, dt, DEV)
    ops.conv_igemm(g_up, wp, None, g_plain, ntaps=4, taps_mode=L.TAPS_GATHER2X2)
    g_fused = ops.new_act(N, H, W, Cin, dt, DEV)
    part = ops.conv_igemm(g_up, wp, None, g_fused, ntaps=4, taps_mode=L.TAPS_GATHER2X2, bnred=(y, vec))
    assert part is not None and part.shape[1:] == (2, Cin)
    assert torch.equal(g_plain.dense(), g_fused.dense())
    out = []
    for partials in (None, part):
        sums = torch.zeros(2, Cin, dtype=torch.float64, device=DEV)
        dx = ops.new_act(N, H, W, Cin, dt, DEV)
        dgb = torch.empty(2, Cin, device=DEV)
        ops.bn_relu_bwd(y, vec, g_fused, None, None, sums, dx, dgb[0], dgb[1], partials=partials)
        out.append((sums.clone(), dx.dense().float(), dgb.clone()))
    assert relerr(out[1][0], out[0][0]) < 2e-5 and relerr(out[1][2], out[0][2]) < 2e-5
    assert relerr(out[1][1], out[0][1]) < 1e-2


@pytest.mark.parametrize("N,H,W,C,K", [(2, 24, 40, 64, 1), (1, 17, 9, 128, 3)])
def test_bn_backward_reduction_in_the_head_gradient(N, H, W, C, K):
    """uz_outconv_bwd_bnred: the 1x1 head's input gradient (common_layers.py:125) and the BatchNorm-backward sums of the
    block that feeds it in one pass"""
    dt = torch.bfloat16
    gen = torch.Generator().manual_seed(13)
    y = act_from_nchw(rnd(dt, torch.randn(N, C, H, W, generator=gen) * 2 + 0.3).to(DEV), dt)
    gamma = (torch.rand(C, generator=gen) + 0.5).to(DEV)
    beta = (torch.randn(C, generator=gen) * 0.2).to(DEV)
    yd = y.dense().double()
    stats = torch.stack([yd.sum((0, 2, 3)), (yd ** 2).sum((0, 2, 3))]).float().reshape(1, 2, C)
    vec = ops.bn_finalize(stats, N * H * W, gamma, beta, 1e-5, 0.1, torch.zeros(C, device=DEV), torch.ones(C, device=DEV))
    act = ops.new_act(N, H, W, C, dt, DEV)
    ops.bn_relu_apply(y, vec[0], vec[1], act, None)
    w = (torch.randn(K, C, generator=gen) * 0.2).to(DEV)
    g = torch.randn(N, K, H, W, generator=gen).to(DEV)
    dx0 = ops.new_act(N, H, W, C, dt, DEV)
    dw0, db0 = ops.outconv_bwd(act, w, g, dx0)
    dx1 = ops.new_act(N, H, W, C, dt, DEV)
    dw1, db1 = ops.outconv_bwd(act, w, g, dx1, bnred=(y, vec))
    assert torch.equal(dx0.dense(), dx1.dense()) and torch.equal(dw0, dw1) and torch.equal(db0, db1)
    part = dx1.bn_partials
    out = []
    for partials in (None, part):
        sums = torch.zeros(2, C, dtype=torch.float64, device=DEV)
        d = ops.new_act(N, H, W, C, dt, DEV)
        dgb = torch.empty(2, C, device=DEV)
        ops.bn_relu_bwd(y, vec, dx1, None, None, sums, d, dgb[0], dgb[1], partials=partials)
        out.append((sums.clone(), d.dense().float(), dgb.clone()))
    assert relerr(out[1][0], out[0][0]) < 2e-5 and relerr(out[1][2], out[0][2]) < 2e-5
    assert relerr(out[1][1], out[0][1]) < 1e-2


def test_conv3x3_padding_of_a_tensor_beyond_256_mib():
    """zero padding is produced by out-of-range buffer offsets: the sentinel offset must lie beyond the END of the tensor
    for every size the plan accepts (< 2 GiB), not just beyond 256 MiB (found in round 2: the first level of
    attention_unet at B = 16, 512 x 512 is 537 MB)"""
    dt = torch.bfloat16
    N, H, W, Cin, Cout = 1, 1040, 1040, 128, 8          # 277 MB input
    gen = torch.Generator().manual_seed(14)
    x = rnd(dt, torch.randn(N, Cin, H, W, generator=gen))
    w = rnd(dt, torch.randn(Cout, Cin, 3, 3, generator=gen) * 0.05)
    xa = act_from_nchw(x.to(DEV), dt)
    assert xa.buf.numel() * 2 > (1 << 28)
    ya = ops.new_act(N, H, W, Cout, dt, DEV)
    ops.conv_igemm(xa, ops.pack_weights(w.to(DEV), L.PACK_CONV_FWD, dt), None, ya, ntaps=9)
    got = ya.dense().float().cpu()
    # the image border (where the padding enters) and an interior band, against F.conv2d on the same rounded operands
    ref_top = F.conv2d(x[:, :, :3, :], w, padding=(1, 1))[:, :, :2, :]
    assert relerr(got[:, :, :2, :], ref_top) < tol(dt)
    ref_bot = F.conv2d(x[:, :, -3:, :], w, padding=(1, 1))[:, :, -2:, :]
    assert relerr(got[:, :, -2:, :], ref_bot) < tol(dt)
    ref_left = F.conv2d(x[:, :, 500:540, :3], w, padding=(0, 1))[:, :, :, :2]
    assert relerr(got[:, :, 501:539, :2], ref_left) < tol(dt)
    ref_right = F.conv2d(x[:, :, 500:540, -3:], w, padding=(0, 1))[:, :, :, -2:]
    assert relerr(got[:, :, 501:539, -2:], ref_right) < tol(dt)
