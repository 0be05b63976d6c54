"""CPU: the round-5 restatements of oracle/uz_ref.c pinned against the torch operators the reference calls (and, for the
backward entries, against torch autograd of those operators).  tests/test_c_ref_r5_gpu.py then holds the kernels against
them on the same bytes."""
from ctypes import byref

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import c_ref
from unet_zoo_amd import _lib as L
from test_c_ref import DTS, close, nchw, nhwc, pack, rnd


def npdt(dt):
    return np.uint16 if dt == torch.bfloat16 else np.float32


def activation(x, scale, shift, dt):
    """relu(x * scale + shift) as the stand-alone pass stores it: one fp32 fma, one rounding to the tensor type"""
    z = torch.addcmul(shift.view(1, -1, 1, 1), x.float(), scale.view(1, -1, 1, 1))     # fp32 (fma or not: inside the tolerance)
    return torch.relu(z).to(dt)


@pytest.mark.parametrize("dt", DTS)
def test_convolution_and_weight_gradient_read_through_batchnorm_relu(dt):
    """uz_conv_igemm_xf_ref / uz_wgrad_xf_ref: Conv2d(relu(bn(x))) and its weight gradient with x the RAW output of the
    convolution in front (common_layers.py:28-33), the middle tensor rounded to the tensor type as the reference's autocast
    stores it"""
    lib = c_ref.load()
    g = torch.Generator().manual_seed(51)
    N, Ci, Co, H, W = 2, 8, 16, 7, 9
    x, w, b = rnd((N, Ci, H, W), dt, g), rnd((Co, Ci, 3, 3), dt, g, 0.3), torch.randn(Co, generator=g)
    scale = (torch.rand(Ci, generator=g) + 0.5) * torch.where(torch.rand(Ci, generator=g) < 0.3, -1.0, 1.0)
    shift = torch.rand(Ci, generator=g) * 0.8 + 0.3           # relu(shift) > 0: a transformed zero pad would show
    a = activation(x, scale, shift, dt)
    ref = F.conv2d(a.double(), w.double(), b.double(), padding=1)
    y = np.zeros(N * H * W * Co, npdt(dt))
    stats = np.zeros(2 * Co, np.float32)
    d = L.ConvDesc(L.dtype_code(dt), N, H, W, H, W, Ci, Ci, Co, Co, 9, L.TAPS_CONV, 1, L.STORE_PLAIN, 0, 0, 0)
    xh, sch, shh, bh = c_ref.host(nhwc(x)), c_ref.host(scale), c_ref.host(shift), c_ref.host(b)
    wp = pack(w, L.PACK_CONV_FWD, dt)
    assert lib.uz_conv_igemm_xf_ref(byref(d), c_ref.ptr(xh), c_ref.ptr(sch), c_ref.ptr(shh), c_ref.ptr(wp), c_ref.ptr(bh), c_ref.ptr(y),
                                    c_ref.ptr(stats), None) == 0
    yt = c_ref.tensor(y, dt).reshape(N * H * W, Co)
    close(nchw(yt, N, H, W), ref, dt, "conv through bn+relu", f32_tol=1e-5)
    np.testing.assert_allclose(stats[:Co], yt.double().sum(0).numpy(), rtol=1e-5, atol=1e-4)
    assert np.array_equal(xh, c_ref.host(nhwc(x)))            # the raw input is read only

    gy = rnd((N, Co, H, W), dt, g)
    wr = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(a.double(), wr, padding=1).backward(gy.double())
    out = np.zeros(Co * Ci * 9, np.float32)
    dw = L.WgradDesc(L.dtype_code(dt), N, H, W, H, W, Co, Co, Ci, Ci, 9, L.TAPS_CONV, 1)
    Lh = c_ref.host(nhwc(gy))
    assert lib.uz_wgrad_xf_ref(byref(dw), c_ref.ptr(Lh), c_ref.ptr(xh), c_ref.ptr(sch), c_ref.ptr(shh), c_ref.ptr(out), None, None, 0) == 0
    np.testing.assert_allclose(out.reshape(Co, Ci, 3, 3), wr.grad.numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("ceil_mode", [0, 1])
def test_batchnorm_relu_residual_pool_and_the_pool_gradient(dt, ceil_mode):
    """uz_bn_relu_add_apply_ref (u2net.py:74 `hx1d + hxin`, :221 MaxPool2d(2, 2, ceil_mode=True)) and uz_pool_grad_combine_ref
    against torch and its autograd, odd sizes included"""
    lib = c_ref.load()
    g = torch.Generator().manual_seed(52 + ceil_mode)
    dc = L.dtype_code(dt)
    N, C, H, W = 2, 8, 7, 9
    y, res = rnd((N, C, H, W), dt, g), rnd((N, C, H, W), dt, g)
    scale, shift = torch.randn(C, generator=g), torch.randn(C, generator=g)
    Hp, Wp = ((H + 1) // 2, (W + 1) // 2) if ceil_mode else (H // 2, W // 2)
    act, pooled = np.zeros(N * H * W * C, npdt(dt)), np.zeros(N * Hp * Wp * C, npdt(dt))
    yh, rh, sch, shh = c_ref.host(nhwc(y)), c_ref.host(nhwc(res)), c_ref.host(scale), c_ref.host(shift)
    assert lib.uz_bn_relu_add_apply_ref(dc, c_ref.ptr(yh), C, c_ref.ptr(sch), c_ref.ptr(shh), N, H, W, C, c_ref.ptr(rh), C, c_ref.ptr(act), C,
                                        c_ref.ptr(pooled), C, ceil_mode, None) == 0
    ref = torch.relu(y.double() * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)) + res.double()
    at = nchw(c_ref.tensor(act, dt).reshape(N * H * W, C), N, H, W)
    close(at, ref, dt, "bn relu add", f32_tol=1e-5)
    # the pool of the STORED tensor, exactly
    assert torch.equal(nchw(c_ref.tensor(pooled, dt).reshape(N * Hp * Wp, C), N, Hp, Wp).float(),
                       F.max_pool2d(at.float(), 2, 2, ceil_mode=bool(ceil_mode)))
    # without the ReLU (bit 1), without a residual
    act2 = np.zeros(N * H * W * C, npdt(dt))
    assert lib.uz_bn_relu_add_apply_ref(dc, c_ref.ptr(yh), C, c_ref.ptr(sch), c_ref.ptr(shh), N, H, W, C, None, 0, c_ref.ptr(act2), C, None, 0,
                                        2, None) == 0
    close(nchw(c_ref.tensor(act2, dt).reshape(N * H * W, C), N, H, W), y.double() * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1),
          dt, "bn only", f32_tol=1e-5)

    # gradient of act used directly twice and through the pool; ties (relu zeros) go to the first maximum as in ATen
    av = at.double().clone().requires_grad_(True)
    g0, g1, gp = rnd((N, C, H, W), dt, g), rnd((N, C, H, W), dt, g), rnd((N, C, Hp, Wp), dt, g)
    (F.max_pool2d(av, 2, 2, ceil_mode=bool(ceil_mode)) * gp.double()).sum().backward()
    want = av.grad + g0.double() + g1.double()
    out = np.zeros(N * H * W * C, npdt(dt))
    g0h, g1h, gph = c_ref.host(nhwc(g0)), c_ref.host(nhwc(g1)), c_ref.host(nhwc(gp))
    assert lib.uz_pool_grad_combine_ref(dc, N, H, W, C, c_ref.ptr(act), C, c_ref.ptr(g0h), C, c_ref.ptr(g1h), C, c_ref.ptr(gph), C, c_ref.ptr(out), C,
                                        ceil_mode, None) == 0
    close(nchw(c_ref.tensor(out, dt).reshape(N * H * W, C), N, H, W), want, dt, "pool grad combine")


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("align", [0, 1])
def test_bilinear_resize_backward(dt, align):
    """uz_resize_bilinear_bwd_ref / uz_bilinear_bwd_ref against autograd of F.interpolate(mode='bilinear') (u2net.py:19-22;
    nn.Upsample(align_corners=True), nested_unet.py:32), up and down, NHWC rows and NCHW one-channel planes"""
    lib = c_ref.load()
    g = torch.Generator().manual_seed(53)
    dc = L.dtype_code(dt)
    for (N, C, Hi, Wi, Ho, Wo) in [(2, 4, 5, 7, 10, 14), (1, 3, 9, 8, 4, 5), (2, 1, 3, 3, 12, 11)]:
        x = torch.zeros(N, C, Hi, Wi, dtype=torch.float64, requires_grad=True)
        gy = rnd((N, C, Ho, Wo), dt, g)
        F.interpolate(x, size=(Ho, Wo), mode="bilinear", align_corners=bool(align)).backward(gy.double())
        gh = c_ref.host(nhwc(gy))
        dx = np.zeros(N * Hi * Wi * C, npdt(dt))
        assert lib.uz_resize_bilinear_bwd_ref(dc, c_ref.ptr(gh), C, Ho * Wo * C, N, Hi, Wi, C, c_ref.ptr(dx), C, Hi * Wi * C, Ho, Wo, align, None) == 0
        close(nchw(c_ref.tensor(dx, dt).reshape(N * Hi * Wi, C), N, Hi, Wi), x.grad, dt, f"resize bwd {Hi}x{Wi}->{Ho}x{Wo}", f32_tol=1e-5)
        if not align:
            dx2 = np.zeros_like(dx)
            assert lib.uz_bilinear_bwd_ref(dc, c_ref.ptr(gh), C, Ho * Wo * C, N, Hi, Wi, C, c_ref.ptr(dx2), C, Hi * Wi * C, Ho, Wo, None) == 0
            assert np.array_equal(dx, dx2)
        if C == 1:       # the NCHW plane form: ld = 1, image stride = plane size
            gp = c_ref.host(gy.contiguous())
            dx3 = np.zeros_like(dx)
            assert lib.uz_resize_bilinear_bwd_ref(dc, c_ref.ptr(gp), 1, Ho * Wo, N, Hi, Wi, 1, c_ref.ptr(dx3), 1, Hi * Wi, Ho, Wo, align, None) == 0
            assert np.array_equal(dx, dx3)


def test_loss_and_metric():
    """uz_bce_dice_ref against nn.BCEWithLogitsLoss (scripts/train.py:135), its autograd and dice_coefficient written out
    (utils/metrics.py:7-24); the empty-union case returns 1"""
    lib = c_ref.load()
    g = torch.Generator().manual_seed(54)
    n = 3 * 17 * 19
    x = (torch.randn(n, generator=g) * 3).requires_grad_(True)
    t = (torch.rand(n, generator=g) > 0.6).float()
    loss = F.binary_cross_entropy_with_logits(x, t)
    loss.backward()
    pred = (torch.sigmoid(x.detach()) > 0.5).float()
    dice = (2 * (pred * t).sum() + 1e-7) / (pred.sum() + t.sum() + 1e-7)
    xh, th = c_ref.host(x), c_ref.host(t)
    dl, out2 = np.zeros(n, np.float32), np.zeros(2, np.float32)
    assert lib.uz_bce_dice_workspace_bytes_ref(n) > 0
    assert lib.uz_bce_dice_ref(c_ref.ptr(xh), c_ref.ptr(th), n, c_ref.ptr(dl), c_ref.ptr(out2), None, None) == 0
    np.testing.assert_allclose(out2, [loss.item(), dice.item()], rtol=1e-6)
    np.testing.assert_allclose(dl, x.grad.numpy(), rtol=1e-5, atol=1e-9)
    neg, zero = c_ref.host(-torch.ones(n)), c_ref.host(torch.zeros(n))
    assert lib.uz_bce_dice_ref(c_ref.ptr(neg), c_ref.ptr(zero), n, None, c_ref.ptr(out2), None, None) == 0
    assert out2[1] == 1.0


@pytest.mark.parametrize("dt", DTS)
def test_dropout_channel_gate_and_row_sums(dt):
    lib = c_ref.load()
    g = torch.Generator().manual_seed(55)
    dc = L.dtype_code(dt)
    P, C, p = 37, 24, 0.25
    x, u = rnd((P, C), dt, g), torch.rand(P, C, generator=g)
    out = np.zeros(P * C, npdt(dt))
    xh, uh = c_ref.host(x), c_ref.host(u)
    assert lib.uz_dropout_ref(dc, c_ref.ptr(xh), C, c_ref.ptr(uh), p, c_ref.ptr(out), C, P, C, None) == 0
    close(c_ref.tensor(out, dt).reshape(P, C), torch.where(u >= p, x.double() / (1 - p), 0.0), dt, "dropout")

    # CCA's gate (uctransnet.py:417-427): out = relu(x * s[n]); its gradients by autograd
    N, HW = 3, 11
    xg, gg = rnd((N * HW, C), dt, g), rnd((N * HW, C), dt, g)
    s, a = torch.rand(N, C, generator=g) + 0.1, torch.randn(N, C, generator=g)
    xv = xg.double().view(N, HW, C).requires_grad_(True)
    sv = s.double().requires_grad_(True)
    fwd = torch.relu(xv * sv[:, None, :])
    fwd.backward(gg.double().view(N, HW, C))
    xgh, ggh, sh, ah = c_ref.host(xg), c_ref.host(gg), c_ref.host(s), c_ref.host(a)
    o = np.zeros(N * HW * C, npdt(dt))
    assert lib.uz_chanscale_relu_ref(dc, 2, None, 0, c_ref.ptr(xgh), C, c_ref.ptr(sh), None, N, HW, C, c_ref.ptr(o), C, None) == 0
    close(c_ref.tensor(o, dt).reshape(N, HW, C), fwd.detach(), dt, "gate forward")
    assert lib.uz_chanscale_relu_ref(dc, 0, c_ref.ptr(ggh), C, c_ref.ptr(xgh), C, c_ref.ptr(sh), None, N, HW, C, c_ref.ptr(o), C, None) == 0
    if dt == torch.float32:     # the column sums of mode 0 are d(loss)/d(s)
        np.testing.assert_allclose(c_ref.tensor(o, dt).reshape(N, HW, C).double().sum(1).numpy(), sv.grad.numpy(), rtol=1e-5, atol=1e-5)
    assert lib.uz_chanscale_relu_ref(dc, 1, c_ref.ptr(ggh), C, c_ref.ptr(xgh), C, c_ref.ptr(sh), c_ref.ptr(ah), N, HW, C, c_ref.ptr(o), C, None) == 0
    close(c_ref.tensor(o, dt).reshape(N, HW, C), xv.grad + a.double()[:, None, :], dt, "gate input gradient")

    if dt == torch.float32:
        rows, n, n0, ld = 13, 20, 12, 29
        part = torch.randn(rows, ld, generator=g)
        ph = c_ref.host(part)
        o0, o1 = np.zeros(n0, np.float32), np.zeros(n - n0, np.float32)
        assert lib.uz_sum_rows_f32_ld_ref(c_ref.ptr(ph), ld, rows, n, c_ref.ptr(o0), n0, c_ref.ptr(o1), None) == 0
        want = part[:, :n].double().sum(0).float().numpy()
        assert np.array_equal(o0, want[:n0]) and np.array_equal(o1, want[n0:])
        dense = c_ref.host(part[:, :n].contiguous())
        o2 = np.zeros(n, np.float32)
        assert lib.uz_sum_rows_f32_ref(c_ref.ptr(dense), rows, n, c_ref.ptr(o2), n, None, None) == 0
        assert np.array_equal(o2, want)


@pytest.mark.parametrize("dt", DTS)
def test_input_gathers_and_pixel_grid_moves(dt):
    """uz_patchify_ref x weight = Conv2d(kernel = stride = patch) (swin_unet_v2.py:548-556); uz_im2col3x3_nchw_ref x weight =
    Conv2d(k3, p1) (unet.py:31); uz_resample2_ref = slicing"""
    lib = c_ref.load()
    g = torch.Generator().manual_seed(56)
    dc = L.dtype_code(dt)
    N, C, H, W, patch, Co = 2, 3, 12, 8, 4, 10
    x = torch.randn(N, C, H, W, generator=g)
    xr = x.to(dt).double()
    K = patch * patch * C
    Kpad = 64
    rows = np.zeros(N * (H // patch) * (W // patch) * Kpad, npdt(dt))
    xh = c_ref.host(x)
    assert lib.uz_patchify_ref(dc, c_ref.ptr(xh), N, C, H, W, patch, Kpad, c_ref.ptr(rows), None) == 0
    r = c_ref.tensor(rows, dt).reshape(-1, Kpad).double()
    assert (r[:, K:] == 0).all()
    w = torch.randn(Co, C, patch, patch, generator=g).double()
    got = r[:, :K] @ w.permute(0, 2, 3, 1).reshape(Co, K).t()           # k = (kh * patch + kw) * C + c
    ref = F.conv2d(xr, w, stride=patch)
    assert torch.allclose(nchw(got, N, H // patch, W // patch), ref, rtol=1e-9, atol=1e-9)

    Kp3 = 32
    col = np.zeros(N * H * W * Kp3, npdt(dt))
    assert lib.uz_im2col3x3_nchw_ref(dc, c_ref.ptr(xh), N, C, H, W, Kp3, c_ref.ptr(col), None) == 0
    cm = c_ref.tensor(col, dt).reshape(-1, Kp3).double()
    assert (cm[:, 9 * C:] == 0).all()
    w3 = torch.randn(Co, C, 3, 3, generator=g).double()
    got = cm[:, :9 * C] @ w3.permute(0, 2, 3, 1).reshape(Co, 9 * C).t()   # k = t * C + c
    assert torch.allclose(nchw(got, N, H, W), F.conv2d(xr, w3, padding=1), rtol=1e-9, atol=1e-9)

    Hs, Ws, Cc = 7, 9, 8
    src = rnd((N, Cc, Hs, Ws), dt, g)
    sh = c_ref.host(nhwc(src))
    Hd, Wd = (Hs + 1) // 2, (Ws + 1) // 2
    d1 = np.zeros(N * Hd * Wd * Cc, npdt(dt))
    assert lib.uz_resample2_ref(dc, c_ref.ptr(sh), Cc, N, Hs, Ws, Cc, c_ref.ptr(d1), Cc, Hd, Wd, 1, None) == 0
    assert torch.equal(nchw(c_ref.tensor(d1, dt).reshape(-1, Cc), N, Hd, Wd).float(), src[:, :, ::2, ::2].float())
    d2 = np.zeros(N * Hs * Ws * Cc, npdt(dt))
    assert lib.uz_resample2_ref(dc, c_ref.ptr(d1), Cc, N, Hd, Wd, Cc, c_ref.ptr(d2), Cc, Hs, Ws, 2, None) == 0
    want = torch.zeros(N, Cc, Hs, Ws)
    want[:, :, ::2, ::2] = src[:, :, ::2, ::2].float()
    assert torch.equal(nchw(c_ref.tensor(d2, dt).reshape(-1, Cc), N, Hs, Ws).float(), want)
    wide = np.zeros(N * Hs * Ws * (Cc + 4), npdt(dt))                      # mode 0: into a slot of a wider buffer
    assert lib.uz_resample2_ref(dc, c_ref.ptr(sh), Cc, N, Hs, Ws, Cc, wide.ctypes.data + 2 * wide.itemsize, Cc + 4, Hs, Ws, 0, None) == 0
    wt = c_ref.tensor(wide, dt).reshape(-1, Cc + 4)
    assert torch.equal(wt[:, 2:2 + Cc].float(), nhwc(src).float()) and (wt[:, :2] == 0).all() and (wt[:, 2 + Cc:] == 0).all()


def bn_rows(v, gamma, beta, eps=1e-5):
    """(scale, shift, mean, invstd) rows as uz_bn_finalize leaves them, of the columns of v"""
    mean, var = v.mean(0), v.var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + eps)
    scale = gamma * invstd
    return torch.stack([scale, beta - mean * scale, mean, invstd])


@pytest.mark.parametrize("dt", DTS)
def test_attention_gate_backward_restatement(dt):
    """uz_attn_bwd_psi_ref -> uz_attn_bwd_reduce_ref -> uz_attn_bwd_apply_ref against autograd of AttentionBlock.forward
    (attention_unet.py:34-40) with its three BatchNorms in training mode: gradients of the gated tensor (direct part), of the
    raw outputs of W_g / W_x, of the psi convolution's weight and bias"""
    lib = c_ref.load()
    g = torch.Generator().manual_seed(57)
    dc = L.dtype_code(dt)
    P, Fi, C = 90, 12, 20
    g1r, x1r, x, dout = rnd((P, Fi), dt, g), rnd((P, Fi), dt, g), rnd((P, C), dt, g), rnd((P, C), dt, g)
    gam = [torch.rand(n, generator=g).double() + 0.5 for n in (Fi, Fi, 1)]
    bet = [torch.randn(n, generator=g).double() * 0.2 for n in (Fi, Fi, 1)]
    wpsi, bpsi = torch.randn(Fi, generator=g).double() * 0.5, torch.randn(1, generator=g).double()

    tg, tx = g1r.double().requires_grad_(True), x1r.double().requires_grad_(True)
    txx = x.double().requires_grad_(True)
    tw, tb = wpsi.clone().requires_grad_(True), bpsi.clone().requires_grad_(True)
    bn = lambda v, i: F.batch_norm(v, None, None, gam[i], bet[i], training=True, eps=1e-5)
    s = torch.relu(bn(tg, 0) + bn(tx, 1))
    q = s @ tw + tb
    psi = torch.sigmoid(bn(q[:, None], 2))
    (txx * psi).backward(dout.double())

    vg, vx = bn_rows(g1r.double(), gam[0], bet[0]).float(), bn_rows(x1r.double(), gam[1], bet[1]).float()
    qf = q.detach().float()
    vq = bn_rows(qf.double()[:, None], gam[2], bet[2]).float()
    h = c_ref.host
    doh, xh, qh, vqh = h(dout), h(x), h(qf), h(vq)
    dxd, dz, part = np.zeros(P * C, npdt(dt)), np.zeros(P, np.float32), np.zeros(2, np.float32)
    assert lib.uz_attn_bwd_psi_ref(dc, c_ref.ptr(doh), C, c_ref.ptr(xh), C, c_ref.ptr(qh), c_ref.ptr(vqh), P, C, c_ref.ptr(dxd), C, c_ref.ptr(dz),
                                   c_ref.ptr(part), None) == 0
    close(c_ref.tensor(dxd, dt).reshape(P, C), dout.double() * psi.detach(), dt, "dx direct", f32_tol=1e-5)
    a01 = part.astype(np.float64)
    g1h, x1h, wh, vgh, vxh = h(g1r), h(x1r), h(wpsi.float()), h(vg), h(vx)
    red = np.zeros(4 * Fi + 1, np.float32)
    assert lib.uz_attn_bwd_reduce_ref(dc, c_ref.ptr(g1h), Fi, c_ref.ptr(x1h), Fi, c_ref.ptr(qh), c_ref.ptr(dz), c_ref.ptr(wh), c_ref.ptr(vgh),
                                      c_ref.ptr(vxh), c_ref.ptr(vqh), c_ref.ptr(a01), P, Fi, c_ref.ptr(red), None) == 0
    np.testing.assert_allclose(red[3 * Fi:4 * Fi], tw.grad.numpy(), rtol=2e-4, atol=2e-5)       # d w_psi
    np.testing.assert_allclose(red[4 * Fi], tb.grad.item(), rtol=2e-4, atol=2e-5)               # d b_psi (zero through the BatchNorm)
    totals = red.astype(np.float64)
    dg, dx1 = np.zeros(P * Fi, npdt(dt)), np.zeros(P * Fi, npdt(dt))
    assert lib.uz_attn_bwd_apply_ref(dc, c_ref.ptr(g1h), Fi, c_ref.ptr(x1h), Fi, c_ref.ptr(qh), c_ref.ptr(dz), c_ref.ptr(wh), c_ref.ptr(vgh),
                                     c_ref.ptr(vxh), c_ref.ptr(vqh), c_ref.ptr(a01), c_ref.ptr(totals), P, Fi, c_ref.ptr(dg), Fi, c_ref.ptr(dx1),
                                     Fi, None) == 0
    tol = 2e-4
    close(c_ref.tensor(dg, dt).reshape(P, Fi), tg.grad, dt, "d g1raw", f32_tol=tol)
    close(c_ref.tensor(dx1, dt).reshape(P, Fi), tx.grad, dt, "d x1raw", f32_tol=tol)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("blocks", [1, 4])
def test_spatial_reduction_attention_restatement(dt, blocks):
    """uz_sra_fwd_ref / uz_sra_bwd_ref against softmax(q k^T scale) v and its autograd (EfficientSelfAtten, missformer.py:21-39;
    blocks = 4: the bridge's keys stored as four [B][kps] blocks, missformer.py:81-100)"""
    lib = c_ref.load()
    g = torch.Generator().manual_seed(58)
    dc = L.dtype_code(dt)
    B, Nq, heads, D, kps = 2, 10, 2, 8, 3
    NK, HD = kps * blocks, heads * D
    scale = D ** -0.5
    q, kv, go = rnd((B * Nq, HD), dt, g), rnd((B * NK, 2 * HD), dt, g), rnd((B * Nq, HD), dt, g)
    # key j of image b sits at row ((j // kps) * B + b) * kps + j % kps
    idx = torch.tensor([[((j // kps) * B + b) * kps + j % kps for j in range(NK)] for b in range(B)])
    tq = q.double().view(B, Nq, heads, D).transpose(1, 2).requires_grad_(True)
    tkv = kv.double().requires_grad_(True)
    kk = tkv[idx][..., :HD].view(B, NK, heads, D).transpose(1, 2)
    vv = tkv[idx][..., HD:].view(B, NK, heads, D).transpose(1, 2)
    sc = (tq @ kk.transpose(-1, -2)) * scale
    o = torch.softmax(sc, -1) @ vv                                           # (B, heads, Nq, D)
    o_rows = o.transpose(1, 2).reshape(B * Nq, HD)
    d = L.SraDesc(dc, B, Nq, NK, heads, D, kps, HD, 2 * HD, 2 * HD, HD, scale)
    h = c_ref.host
    qh, kvh = h(q), h(kv)
    out, lse = np.zeros(B * Nq * HD, npdt(dt)), np.zeros(B * heads * Nq, np.float32)
    vptr = kvh.ctypes.data + HD * kvh.itemsize
    assert lib.uz_sra_fwd_ref(byref(d), c_ref.ptr(qh), c_ref.ptr(kvh), vptr, c_ref.ptr(out), c_ref.ptr(lse), None) == 0
    close(c_ref.tensor(out, dt).reshape(B * Nq, HD), o_rows.detach(), dt, "sra forward", f32_tol=1e-5)
    np.testing.assert_allclose(lse.reshape(B, heads, Nq) * np.log(2.0), torch.logsumexp(sc.detach(), -1).numpy(), rtol=1e-5, atol=1e-5)   # log2 units

    o_rows.backward(go.double())
    oh, goh = h(o_rows.detach().to(dt)), h(go)
    dq, dkv = np.zeros(B * Nq * HD, npdt(dt)), np.zeros(B * NK * 2 * HD, npdt(dt))
    assert lib.uz_sra_bwd_workspace_bytes_ref(byref(d)) > 0
    assert lib.uz_sra_bwd_ref(byref(d), c_ref.ptr(qh), c_ref.ptr(kvh), vptr, c_ref.ptr(oh), c_ref.ptr(lse), c_ref.ptr(goh), HD, c_ref.ptr(dq), HD,
                              c_ref.ptr(dkv), 2 * HD, None, None) == 0
    tol = 1e-4 if dt == torch.float32 else None
    want_dq = tq.grad.transpose(1, 2).reshape(B * Nq, HD)
    if dt == torch.float32:
        close(c_ref.tensor(dq, dt).reshape(B * Nq, HD), want_dq, dt, "dq", f32_tol=tol)
        close(c_ref.tensor(dkv, dt).reshape(B * NK, 2 * HD), tkv.grad, dt, "dkv", f32_tol=tol)
    else:       # o was rounded to bf16 before delta = dO . O: a bf16 rounding of O moves the result by ~2^-8 of |dO||O|
        for got, want, what in ((dq, want_dq, "dq"), (dkv, tkv.grad, "dkv")):
            got = c_ref.tensor(got, dt).reshape(want.shape).double()
            assert ((got - want).abs().max() / want.abs().max()).item() < 2e-2, what


def test_head_reads_the_last_block_through_batchnorm_relu():
    """uz_outconv_fwd_xf_ref: OutConv on relu(bn(y)) with y the RAW output of the last decoder convolution
    (common_layers.py:31-33, then :125), the activation rounded to bf16 as the stand-alone pass stores it"""
    lib = c_ref.load()
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(57)
    N, C, H, W, K = 2, 24, 5, 7, 3
    y = rnd((N, C, H, W), dt, g)
    w, b = torch.randn(K, C, generator=g) * 0.3, torch.randn(K, generator=g)
    scale = (torch.rand(C, generator=g) + 0.5) * torch.where(torch.rand(C, generator=g) < 0.3, -1.0, 1.0)
    shift = torch.randn(C, generator=g) * 0.5
    a = activation(y, scale, shift, dt)
    ref = F.conv2d(a.double(), w.double().view(K, C, 1, 1), b.double())
    out = np.zeros(N * K * H * W, np.float32)
    yh, sch, shh, wh, bh = c_ref.host(nhwc(y)), c_ref.host(scale), c_ref.host(shift), c_ref.host(w), c_ref.host(b)
    assert lib.uz_outconv_fwd_xf_ref(L.dtype_code(dt), c_ref.ptr(yh), C, N, H * W, C, c_ref.ptr(sch), c_ref.ptr(shh), c_ref.ptr(wh),
                                     c_ref.ptr(bh), K, c_ref.ptr(out), None) == 0
    got = torch.from_numpy(out).reshape(N, K, H, W).double()
    # fma against multiply-then-add before the bf16 rounding: an element may land on the neighbouring bf16 number
    assert ((got - ref).abs().max() / ref.abs().max()).item() < 2e-3
    # and it is the plain head on the stored activation
    plain = np.zeros(N * K * H * W, np.float32)
    ah = c_ref.host(nhwc(a))
    assert lib.uz_outconv_fwd_ref(L.dtype_code(dt), c_ref.ptr(ah), C, N, H * W, C, c_ref.ptr(wh), c_ref.ptr(bh), K, c_ref.ptr(plain), None) == 0
    assert ((torch.from_numpy(plain).double().reshape(N, K, H, W) - got).abs().max() / ref.abs().max()).item() < 2e-3
    assert lib.uz_outconv_fwd_xf_ref(L.dtype_code(torch.float32), c_ref.ptr(yh), C, N, H * W, C, c_ref.ptr(sch), c_ref.ptr(shh),
                                     c_ref.ptr(wh), c_ref.ptr(bh), K, c_ref.ptr(out), None) != 0      # bf16 only, as the kernel
