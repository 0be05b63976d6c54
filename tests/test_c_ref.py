"""CPU: the plain-C restatement of the kernel library (oracle/uz_ref.c, one `<entry>_ref` per entry, same signature,
compiled from the same header) against the torch operators the reference calls -- this pins the restatement; the GPU tests
(tests/test_c_ref_gpu.py) then hold the kernels against it on the same bytes."""
import ctypes
import math
from ctypes import byref

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import c_ref
from unet_zoo_amd import _lib as L

DTS = [torch.float32, torch.bfloat16]


def rnd(shape, dt, g, scale=1.0):
    return (scale * torch.randn(*shape, generator=g)).to(dt)


def nhwc(t):          # (N, C, H, W) -> (N*H*W, C) rows
    N, C, H, W = t.shape
    return t.permute(0, 2, 3, 1).reshape(N * H * W, C).contiguous()


def nchw(rows, N, H, W):
    return rows.reshape(N, H, W, -1).permute(0, 3, 1, 2)


def close(got, ref, dt, what="", f32_tol=1e-6):
    got, ref = got.double(), ref.double()
    tol = f32_tol if dt == torch.float32 else 2.0 ** -8          # one rounding to the tensor type
    err = ((got - ref).abs() / (ref.abs() + 1e-3 * ref.abs().max() + 1e-30)).max().item()
    assert err <= tol, (what, err)


def pack(w, mode, dt, kpad=0):
    lib = c_ref.load()
    d0, d1 = w.shape[0], w.shape[1]
    T = w.numel() // (d0 * d1)
    co, ci = (d0, d1) if mode in (L.PACK_CONV_FWD, L.PACK_CONV_DGRAD, L.PACK_IM2COL) else (d1, d0)
    n = {L.PACK_CONV_FWD: co * T * ci, L.PACK_CONV_DGRAD: ci * T * co, L.PACK_CONVT_FWD: T * co * ci, L.PACK_CONVT_DGRAD: ci * T * co}.get(mode, co * kpad)
    dst = np.zeros(n, dtype=np.uint16 if dt == torch.bfloat16 else np.float32)
    src = c_ref.host(w.float())
    assert lib.uz_pack_weights_ref(L.dtype_code(dt), mode, c_ref.ptr(src), co, ci, T, kpad, c_ref.ptr(dst), None) == 0
    return dst


def conv_ref(dt, x_rows, w_packed, bias, N, H, W, Hin, Win, Cin, Nout, ntaps, mode=L.TAPS_CONV, dil=1, store=L.STORE_PLAIN, co=0,
             out_pixels=None, ldy=None, want_stats=False):
    lib = c_ref.load()
    ldy = ldy or (co if store == L.STORE_SHUFFLE2X2 else Nout)
    P = out_pixels if out_pixels is not None else N * H * W
    y = np.zeros(P * ldy, dtype=np.uint16 if dt == torch.bfloat16 else np.float32)
    d = L.ConvDesc(L.dtype_code(dt), N, H, W, Hin, Win, Cin, Cin, Nout, ldy, ntaps, mode, dil, store, co, 0, 0)
    stats = np.zeros(2 * Nout, dtype=np.float32) if want_stats else None
    xh = c_ref.host(x_rows)
    b = c_ref.host(bias.float()) if bias is not None else None
    assert lib.uz_conv_igemm_ref(byref(d), c_ref.ptr(xh), c_ref.ptr(w_packed), c_ref.ptr(b), c_ref.ptr(y), c_ref.ptr(stats), None) == 0
    return c_ref.tensor(y, dt).reshape(P, ldy), stats


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("dil", [1, 2])
def test_conv3x3_and_its_statistics(dt, dil):
    """nn.Conv2d(k3, padding=dilation) (common_layers.py:28; u2net.py:10) + the sums BatchNorm needs of the stored output"""
    g = torch.Generator().manual_seed(1)
    N, Ci, Co, H, W = 2, 8, 16, 7, 9
    x, w, b = rnd((N, Ci, H, W), dt, g), rnd((Co, Ci, 3, 3), dt, g, 0.3), torch.randn(Co, generator=g)
    y, stats = conv_ref(dt, nhwc(x), pack(w, L.PACK_CONV_FWD, dt), b, N, H, W, H, W, Ci, Co, 9, dil=dil, want_stats=True)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=dil, dilation=dil)
    close(nchw(y, N, H, W), ref, dt, "conv")
    yd = y.double()
    np.testing.assert_allclose(stats[:Co], yd.sum(0).numpy(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(stats[Co:], (yd * yd).sum(0).numpy(), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("dt", DTS)
def test_conv1x1_upsampled_and_strided_convolutions(dt):
    g = torch.Generator().manual_seed(2)
    N, Ci, Co = 2, 8, 8
    x = rnd((N, Ci, 6, 10), dt, g)
    w1 = rnd((Co, Ci, 1, 1), dt, g)
    y, _ = conv_ref(dt, nhwc(x), pack(w1, L.PACK_CONV_FWD, dt), None, N, 6, 10, 6, 10, Ci, Co, 1)
    close(nchw(y, N, 6, 10), F.conv2d(x.double(), w1.double()), dt, "1x1")
    # nn.Upsample(scale_factor=2) + Conv2d k3 (common_layers.py:69-72): the input lives at half resolution
    w3 = rnd((Co, Ci, 3, 3), dt, g, 0.3)
    y, _ = conv_ref(dt, nhwc(x), pack(w3, L.PACK_CONV_FWD, dt), None, N, 12, 20, 6, 10, Ci, Co, 9, mode=L.TAPS_CONV_UP2)
    close(nchw(y, N, 12, 20), F.conv2d(F.interpolate(x.double(), scale_factor=2, mode="nearest"), w3.double(), padding=1), dt, "up2")
    # Conv2d(k3, stride 2, padding 1) (common_layers.py:188), odd input size
    xo = rnd((N, Ci, 7, 9), dt, g)
    y, _ = conv_ref(dt, nhwc(xo), pack(w3, L.PACK_CONV_FWD, dt), None, N, 4, 5, 7, 9, Ci, Co, 9, mode=L.TAPS_CONV_S2)
    close(nchw(y, N, 4, 5), F.conv2d(xo.double(), w3.double(), stride=2, padding=1), dt, "stride 2")


@pytest.mark.parametrize("dt", DTS)
def test_conv_transpose_forward_and_input_gradient(dt):
    """nn.ConvTranspose2d(k2, s2) (common_layers.py:104): pixel-shuffle store forward, 2x2 gather for the input gradient"""
    g = torch.Generator().manual_seed(3)
    N, Ci, Co, H, W = 2, 8, 8, 5, 6
    x = rnd((N, Ci, H, W), dt, g)
    w = rnd((Ci, Co, 2, 2), dt, g, 0.5)
    b = torch.randn(Co, generator=g)
    y, _ = conv_ref(dt, nhwc(x), pack(w, L.PACK_CONVT_FWD, dt), b.repeat(4), N, H, W, H, W, Ci, 4 * Co, 1, store=L.STORE_SHUFFLE2X2, co=Co,
                    out_pixels=N * 2 * H * 2 * W)
    close(nchw(y, N, 2 * H, 2 * W), F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2), dt, "convT fwd")
    gy = rnd((N, Co, 2 * H, 2 * W), dt, g)
    xr = x.double().requires_grad_(True)
    F.conv_transpose2d(xr, w.double(), stride=2).backward(gy.double())
    dx, _ = conv_ref(dt, nhwc(gy), pack(w, L.PACK_CONVT_DGRAD, dt), None, N, H, W, 2 * H, 2 * W, Co, Ci, 4, mode=L.TAPS_GATHER2X2)
    close(nchw(dx, N, H, W), xr.grad, dt, "convT dgrad")
    # and the 3x3 input gradient: the same entry with the flipped-tap packing
    w3 = rnd((Co, Ci, 3, 3), dt, g, 0.3)
    x3 = rnd((N, Ci, H, W), dt, g).double().requires_grad_(True)
    g3 = rnd((N, Co, H, W), dt, g)
    F.conv2d(x3, w3.double(), padding=1).backward(g3.double())
    dx3, _ = conv_ref(dt, nhwc(g3), pack(w3, L.PACK_CONV_DGRAD, dt), None, N, H, W, H, W, Co, Ci, 9)
    close(nchw(dx3, N, H, W), x3.grad, dt, "conv dgrad")


@pytest.mark.parametrize("dt", DTS)
def test_weight_gradients(dt):
    lib = c_ref.load()
    g = torch.Generator().manual_seed(4)
    N, Ci, Co, H, W = 2, 8, 16, 6, 7
    x, gy = rnd((N, Ci, H, W), dt, g), rnd((N, Co, H, W), dt, g)
    for dil in (1, 2):
        wr = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, requires_grad=True)
        F.conv2d(x.double(), wr, padding=dil, dilation=dil).backward(gy.double())
        out = np.zeros(Co * Ci * 9, dtype=np.float32)
        d = L.WgradDesc(L.dtype_code(dt), N, H, W, H, W, Co, Co, Ci, Ci, 9, L.TAPS_CONV, dil)
        Lh, Rh = c_ref.host(nhwc(gy)), c_ref.host(nhwc(x))
        assert lib.uz_wgrad_ref(byref(d), c_ref.ptr(Lh), c_ref.ptr(Rh), c_ref.ptr(out), None, None) == 0
        np.testing.assert_allclose(out.reshape(Co, Ci, 3, 3), wr.grad.numpy(), rtol=1e-5, atol=1e-5)
    # ConvTranspose2d k2 s2: L = x, R = dy on the doubled grid -> (Cin, Cout, 2, 2)
    gt = rnd((N, Co, 2 * H, 2 * W), dt, g)
    wt = torch.zeros(Ci, Co, 2, 2, dtype=torch.float64, requires_grad=True)
    F.conv_transpose2d(x.double(), wt, stride=2).backward(gt.double())
    out = np.zeros(Ci * Co * 4, dtype=np.float32)
    d = L.WgradDesc(L.dtype_code(dt), N, H, W, 2 * H, 2 * W, Ci, Ci, Co, Co, 4, L.TAPS_GATHER2X2, 1)
    Lh, Rh = c_ref.host(nhwc(x)), c_ref.host(nhwc(gt))
    assert lib.uz_wgrad_ref(byref(d), c_ref.ptr(Lh), c_ref.ptr(Rh), c_ref.ptr(out), None, None) == 0
    np.testing.assert_allclose(out.reshape(Ci, Co, 2, 2), wt.grad.numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dt", DTS)
def test_batchnorm_relu_pool_forward_and_backward(dt):
    """nn.BatchNorm2d (train) -> nn.ReLU -> nn.MaxPool2d(2) (common_layers.py:29-33, :90) and autograd's backward of it with a
    gradient on the activation and one on the pooled tensor"""
    lib = c_ref.load()
    g = torch.Generator().manual_seed(5)
    N, C, H, W = 2, 8, 6, 8
    y = rnd((N, C, H, W), dt, g)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rows = nhwc(y)
    yd = rows.double()
    stats = np.concatenate([yd.sum(0).numpy(), (yd * yd).sum(0).numpy()]).astype(np.float32)
    rm, rv = np.zeros(C, np.float32), np.ones(C, np.float32)
    scale, shift, mean, invstd = (np.zeros(C, np.float32) for _ in range(4))
    cnt = float(N * H * W)
    gm, bt = c_ref.host(gamma), c_ref.host(beta)
    assert lib.uz_bn_finalize_ref(c_ref.ptr(stats), 1, C, cnt, c_ref.ptr(gm), c_ref.ptr(bt), 1e-5, 0.1, c_ref.ptr(rm), c_ref.ptr(rv),
                                  c_ref.ptr(scale), c_ref.ptr(shift), c_ref.ptr(mean), c_ref.ptr(invstd), None) == 0
    bn = torch.nn.BatchNorm2d(C).double()
    with torch.no_grad():
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    yr = y.double().requires_grad_(True)
    act_ref = F.relu(bn(yr))
    pool_ref = F.max_pool2d(act_ref, 2)
    np.testing.assert_allclose(rm, bn.running_mean.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rv, bn.running_var.numpy(), rtol=1e-5, atol=1e-6)
    npdt = np.uint16 if dt == torch.bfloat16 else np.float32
    act, pooled = np.zeros(N * H * W * C, npdt), np.zeros(N * (H // 2) * (W // 2) * C, npdt)
    yh = c_ref.host(rows)
    assert lib.uz_bn_relu_apply_ref(L.dtype_code(dt), c_ref.ptr(yh), C, c_ref.ptr(scale), c_ref.ptr(shift), N, H, W, C, c_ref.ptr(act), C,
                                    c_ref.ptr(pooled), C, None) == 0
    # scale * y + shift is evaluated in fp32 from fp32 scale / shift vectors, as the kernels do: cancellation costs a few ulp
    close(nchw(c_ref.tensor(act, dt).reshape(-1, C), N, H, W), act_ref.detach(), dt, "act", f32_tol=2e-5)
    close(nchw(c_ref.tensor(pooled, dt).reshape(-1, C), N, H // 2, W // 2), pool_ref.detach(), dt, "pooled", f32_tol=2e-5)
    if dt == torch.bfloat16:
        return   # the backward comparison needs the masks of the rounded activation; fp32 pins the formulas
    g0, gp = rnd((N, C, H, W), dt, g), rnd((N, C, H // 2, W // 2), dt, g)
    (act_ref * g0.double()).sum().backward(retain_graph=True)
    (pool_ref * gp.double()).sum().backward()
    d = L.BnBwdDesc(L.dtype_code(dt), N, H, W, C, C, C, 0, C, C, 0)
    sums = np.zeros(2 * C, np.float64)
    dgam, dbet = np.zeros(C, np.float32), np.zeros(C, np.float32)
    g0h, gph = c_ref.host(nhwc(g0)), c_ref.host(nhwc(gp))
    args = (byref(d), c_ref.ptr(yh), c_ref.ptr(scale), c_ref.ptr(shift), c_ref.ptr(mean), c_ref.ptr(invstd), c_ref.ptr(g0h), None, c_ref.ptr(gph))
    assert lib.uz_bn_relu_bwd_reduce_ref(*args, None, c_ref.ptr(sums), c_ref.ptr(dgam), c_ref.ptr(dbet), None) == 0
    np.testing.assert_allclose(dgam, bn.weight.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dbet, bn.bias.grad.numpy(), rtol=1e-4, atol=1e-5)
    dy = np.zeros(N * H * W * C, np.float32)
    assert lib.uz_bn_relu_bwd_apply_ref(*args, c_ref.ptr(sums), cnt, c_ref.ptr(dy), None) == 0
    close(nchw(torch.from_numpy(dy).reshape(-1, C), N, H, W), yr.grad, dt, "dy", f32_tol=2e-5)


@pytest.mark.parametrize("dt", DTS)
def test_batched_products_softmax_and_adaptive_pool(dt):
    lib = c_ref.load()
    g = torch.Generator().manual_seed(6)
    B, M, N, K = 3, 10, 16, 24
    x, w, res = rnd((B, M, K), dt, g), rnd((B, N, K), dt, g), rnd((B, M, N), dt, g)
    bias = torch.randn(N, generator=g)
    y = np.zeros(B * M * N, np.uint16 if dt == torch.bfloat16 else np.float32)
    d = L.GemmDesc(L.dtype_code(dt), B, M, N, K, K, K, N, N, M * K, N * K, M * N, M * N)
    xh, wh, rh, bh = c_ref.host(x), c_ref.host(w), c_ref.host(res), c_ref.host(bias)
    assert lib.uz_gemm_nt_ref(byref(d), c_ref.ptr(xh), c_ref.ptr(wh), c_ref.ptr(bh), c_ref.ptr(rh), c_ref.ptr(y), None) == 0
    prod = torch.matmul(x.double(), w.double().transpose(1, 2)) + bias.double()
    close(c_ref.tensor(y, dt).reshape(B, M, N), prod.to(dt).double() + res.double(), dt, "gemm_nt")
    for axis in (0, 1):
        s = rnd((B, 12, 16), dt, g, 2.0)
        sh = c_ref.host(s)
        assert lib.uz_softmax_fwd_ref(L.dtype_code(dt), c_ref.ptr(sh), 16, 12 * 16, B, 12, 16, axis, 0.5, None, None) == 0
        a = c_ref.tensor(sh, dt).reshape(B, 12, 16)
        close(a, torch.softmax(s.double() * 0.5, dim=1 + axis), dt, "softmax")
        da = rnd((B, 12, 16), dt, g)
        dh = c_ref.host(da)
        dot = np.zeros(B * 16, np.float32)
        assert lib.uz_softmax_bwd_ref(L.dtype_code(dt), c_ref.ptr(sh), c_ref.ptr(dh), 16, 12 * 16, B, 12, 16, axis, 0.5, c_ref.ptr(dot), 0, None) == 0
        ad = a.double()
        close(c_ref.tensor(dh, dt).reshape(B, 12, 16), ad * (da.double() - (ad * da.double()).sum(1 + axis, keepdim=True)) * 0.5, dt, "softmax bwd")
    xa = rnd((2, 8, 7, 9), dt, g)
    for (Ho, Wo) in ((3, 4), (14, 18)):
        xr = xa.double().requires_grad_(True)
        ref = F.adaptive_avg_pool2d(xr, (Ho, Wo))
        yo = np.zeros(2 * Ho * Wo * 8, np.uint16 if dt == torch.bfloat16 else np.float32)
        xh = c_ref.host(nhwc(xa))
        assert lib.uz_adaptive_avgpool_fwd_ref(L.dtype_code(dt), c_ref.ptr(xh), 8, 2, 7, 9, 8, c_ref.ptr(yo), 8, Ho, Wo, None) == 0
        close(nchw(c_ref.tensor(yo, dt).reshape(-1, 8), 2, Ho, Wo), ref.detach(), dt, "adaptive pool")
        gy = rnd(tuple(ref.shape), dt, g)
        ref.backward(gy.double())
        dx = np.zeros(2 * 7 * 9 * 8, np.uint16 if dt == torch.bfloat16 else np.float32)
        gh = c_ref.host(nhwc(gy))
        assert lib.uz_adaptive_avgpool_bwd_ref(L.dtype_code(dt), c_ref.ptr(gh), 8, 2, 7, 9, 8, c_ref.ptr(dx), 8, Ho, Wo, 0, None) == 0
        close(nchw(c_ref.tensor(dx, dt).reshape(-1, 8), 2, 7, 9), xr.grad, dt, "adaptive pool bwd")


def test_channel_attention_probabilities_and_their_gradient():
    """softmax(InstanceNorm2d(scores / sqrt(KV))) per (image, head) plane (uctransnet.py:170-178), P / heads laid out
    (C, heads * KV), against autograd"""
    lib = c_ref.load()
    g = torch.Generator().manual_seed(7)
    B, H, C, KV = 2, 4, 8, 24
    scores = torch.randn(B, H, C, KV, generator=g)
    scale = 1.0 / math.sqrt(KV)
    sr = scores.double().requires_grad_(True)
    P = torch.softmax(F.instance_norm(sr * scale), dim=3) / H
    pcat_ref = P.permute(0, 2, 1, 3).reshape(B, C, H * KV)
    pc, pct = np.zeros(B * C * H * KV, np.float32), np.zeros(B * C * H * KV, np.float32)
    sh = c_ref.host(scores)
    assert lib.uz_chanattn_probs_fwd_ref(0, c_ref.ptr(sh), B, H, C, KV, scale, 1e-5, c_ref.ptr(pc), c_ref.ptr(pct), None) == 0
    np.testing.assert_allclose(pc.reshape(B, C, H * KV), pcat_ref.detach().numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(pct.reshape(B, H * KV, C), pcat_ref.detach().transpose(1, 2).numpy(), rtol=1e-5, atol=1e-7)
    dpc = torch.randn(B, C, H * KV, generator=g)
    pcat_ref.backward(dpc.double())
    ds, dst = np.zeros(B * H * C * KV, np.float32), np.zeros(B * H * C * KV, np.float32)
    dh = c_ref.host(dpc)
    assert lib.uz_chanattn_probs_bwd_ref(0, c_ref.ptr(sh), c_ref.ptr(dh), B, H, C, KV, scale, 1e-5, c_ref.ptr(ds), c_ref.ptr(dst), None) == 0
    np.testing.assert_allclose(ds.reshape(B, H, C, KV), sr.grad.numpy(), rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(dst.reshape(B, H, KV, C), sr.grad.transpose(2, 3).numpy(), rtol=1e-4, atol=1e-7)


def test_every_restated_entry_exists_with_the_products_signature():
    lib, prod = c_ref.load(), L.load()
    for name in c_ref.REF_NAMES:
        assert hasattr(lib, name + "_ref") and hasattr(prod, name)
    assert lib.uz_ref_abi_version() == prod.uz_abi_version()


# ---- second batch: element passes of the transformer / residual families ---------------------------------------------
@pytest.mark.parametrize("dt", DTS)
def test_gelu_residual_relu_and_upsample_gradient_restatements(dt):
    lib = c_ref.load()
    g = torch.Generator().manual_seed(8)
    dc = L.dtype_code(dt)
    npdt = np.uint16 if dt == torch.bfloat16 else np.float32
    P, C = 70, 24
    x, gy, b = rnd((P, C), dt, g), rnd((P, C), dt, g), rnd((P, C), dt, g)
    xh, gh, bh = c_ref.host(x), c_ref.host(gy), c_ref.host(b)
    y = np.zeros(P * C, npdt)
    assert lib.uz_gelu_fwd_ref(dc, c_ref.ptr(xh), C, c_ref.ptr(y), C, P, C, None) == 0
    xr = x.double().requires_grad_(True)
    ref = F.gelu(xr)
    close(c_ref.tensor(y, dt).reshape(P, C), ref.detach(), dt, "gelu")
    ref.backward(gy.double())
    dx = np.zeros(P * C, npdt)
    assert lib.uz_gelu_bwd_ref(dc, c_ref.ptr(xh), C, c_ref.ptr(gh), C, c_ref.ptr(dx), C, P, C, None) == 0
    close(c_ref.tensor(dx, dt).reshape(P, C), xr.grad, dt, "gelu bwd")
    out = np.zeros(P * C, npdt)
    assert lib.uz_add_relu_ref(dc, c_ref.ptr(xh), C, c_ref.ptr(bh), C, c_ref.ptr(out), C, P, C, None) == 0
    close(c_ref.tensor(out, dt).reshape(P, C), F.relu(x.double() + b.double()), dt, "add_relu")
    assert lib.uz_relu_bwd_ref(dc, c_ref.ptr(out), C, c_ref.ptr(gh), C, c_ref.ptr(dx), C, P, C, None) == 0
    close(c_ref.tensor(dx, dt).reshape(P, C), gy.double() * (c_ref.tensor(out, dt).reshape(P, C).double() > 0), dt, "relu bwd")
    # nearest x2 upsampling's gradient
    N, H, W = 2, 5, 6
    du = rnd((N, C, 2 * H, 2 * W), dt, g)
    xs = torch.zeros(N, C, H, W, dtype=torch.float64, requires_grad=True)
    F.interpolate(xs, scale_factor=2, mode="nearest").backward(du.double())
    dxs = np.zeros(N * H * W * C, npdt)
    duh = c_ref.host(nhwc(du))
    assert lib.uz_sum2x2_ref(dc, c_ref.ptr(duh), C, N, H, W, C, c_ref.ptr(dxs), C, None) == 0
    close(nchw(c_ref.tensor(dxs, dt).reshape(-1, C), N, H, W), xs.grad, dt, "sum2x2")
    cs = np.zeros(C, np.float32)
    assert lib.uz_colsum_ref(dc, c_ref.ptr(xh), C, P, C, c_ref.ptr(cs), None) == 0
    np.testing.assert_allclose(cs, x.double().sum(0).numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dt", DTS)
def test_bilinear_depthwise_space_to_depth_and_layernorm_restatements(dt):
    lib = c_ref.load()
    g = torch.Generator().manual_seed(9)
    dc = L.dtype_code(dt)
    npdt = np.uint16 if dt == torch.bfloat16 else np.float32
    N, C, Hi, Wi = 2, 8, 5, 7
    x = rnd((N, C, Hi, Wi), dt, g)
    xh = c_ref.host(nhwc(x))
    for (Ho, Wo) in ((10, 14), (8, 9), (3, 4)):
        for ac in (0, 1):
            y = np.zeros(N * Ho * Wo * C, npdt)
            assert lib.uz_resize_bilinear_fwd_ref(dc, c_ref.ptr(xh), C, Hi * Wi * C, N, Hi, Wi, C, c_ref.ptr(y), C, Ho * Wo * C, Ho, Wo, ac, None) == 0
            ref = F.interpolate(x.double(), size=(Ho, Wo), mode="bilinear", align_corners=bool(ac))
            close(nchw(c_ref.tensor(y, dt).reshape(-1, C), N, Ho, Wo), ref, dt, f"bilinear {Ho}x{Wo} ac={ac}")
    # depthwise 3x3 (+ skip) and its input-gradient form (flipped taps)
    w = torch.randn(C, 1, 3, 3, generator=g)
    bias = torch.randn(C, generator=g)
    taps = w.reshape(C, 9).t().contiguous()
    th, bh = c_ref.host(taps), c_ref.host(bias)
    y = np.zeros(N * Hi * Wi * C, npdt)
    assert lib.uz_dwconv3x3_ref(dc, c_ref.ptr(xh), C, c_ref.ptr(th), c_ref.ptr(bh), c_ref.ptr(y), C, N, Hi, Wi, C, 1, None) == 0
    xr = x.double().requires_grad_(True)
    ref = F.conv2d(xr, w.double(), bias.double(), padding=1, groups=C)
    close(nchw(c_ref.tensor(y, dt).reshape(-1, C), N, Hi, Wi), (ref + xr).detach(), dt, "dwconv + skip")
    gy = rnd((N, C, Hi, Wi), dt, g)
    ref.backward(gy.double())
    gh = c_ref.host(nhwc(gy))
    assert lib.uz_dwconv3x3_ref(dc, c_ref.ptr(gh), C, c_ref.ptr(th), None, c_ref.ptr(y), C, N, Hi, Wi, C, 2, None) == 0
    close(nchw(c_ref.tensor(y, dt).reshape(-1, C), N, Hi, Wi), xr.grad, dt, "dwconv input gradient")
    # space to depth: the rows a Conv2d(C, C', r, r) multiplies
    r = 2
    xf = rnd((N, C, 4 * r, 3 * r), dt, g)
    fh = c_ref.host(nhwc(xf))
    cols = np.zeros(N * 4 * 3 * r * r * C, npdt)
    assert lib.uz_space_to_depth_ref(dc, c_ref.ptr(fh), C, c_ref.ptr(cols), r * r * C, N, 4, 3, C, r, 0, None) == 0
    wc = torch.randn(6, C, r, r, generator=g)
    rows = c_ref.tensor(cols, dt).reshape(N * 12, r * r * C).double()
    got = rows @ wc.permute(0, 2, 3, 1).reshape(6, -1).double().t()
    close(nchw(got, N, 4, 3), F.conv2d(xf.double(), wc.double(), stride=r), torch.float32, "space_to_depth rows", f32_tol=1e-9)
    back = np.zeros(N * 4 * r * 3 * r * C, npdt)
    assert lib.uz_space_to_depth_ref(dc, c_ref.ptr(cols), r * r * C, c_ref.ptr(back), C, N, 4, 3, C, r, 1, None) == 0
    assert np.array_equal(back, fh.reshape(-1))
    # LayerNorm with the engine's fused residual / per-image scale, and the GELU form
    P = N * Hi * Wi
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    res = rnd((P, C), dt, g)
    isc = torch.rand(N, generator=g) + 0.5
    d = L.LnDesc(dc, N, Hi, Wi, C, C, C, C, 0, 0, 0, 1, 1e-5, 0)
    stats = np.zeros(2 * P, np.float32)
    yl = np.zeros(P * C, npdt)
    gm, bt, rh, ih = c_ref.host(gamma), c_ref.host(beta), c_ref.host(res), c_ref.host(isc)
    assert lib.uz_layernorm_fwd_ref(byref(d), c_ref.ptr(xh), c_ref.ptr(gm), c_ref.ptr(bt), c_ref.ptr(rh), c_ref.ptr(ih), c_ref.ptr(yl), c_ref.ptr(stats), None) == 0
    xt = nhwc(x).double()
    ln = F.layer_norm(xt, (C,), gamma.double(), beta.double(), 1e-5)
    ref = res.double() + ln * isc.double().repeat_interleave(Hi * Wi)[:, None]
    close(c_ref.tensor(yl, dt).reshape(P, C), ref, dt, "layernorm + res + scale")
    np.testing.assert_allclose(stats.reshape(P, 2)[:, 0], xt.mean(1).numpy(), rtol=1e-5, atol=1e-6)
    d2 = L.LnDesc(dc, N, Hi, Wi, C, C, C, 0, 0, 0, 0, 1, 1e-5, 1)
    assert lib.uz_layernorm_fwd_ref(byref(d2), c_ref.ptr(xh), c_ref.ptr(gm), c_ref.ptr(bt), None, None, c_ref.ptr(yl), c_ref.ptr(stats), None) == 0
    close(c_ref.tensor(yl, dt).reshape(P, C), F.gelu(ln), dt, "gelu(layernorm)")


# ---- the network's first convolution on the fp32 NCHW image (uz_conv3x3_first_*) -------------------------------------------
@pytest.mark.parametrize("N,C,H,W,Cout", [(2, 3, 9, 11, 32), (1, 1, 6, 5, 64)])
def test_first_convolution_restatement(N, C, H, W, Cout):
    lib = c_ref.load()
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(31)
    x, w, b = torch.randn(N, C, H, W, generator=g), torch.randn(Cout, C, 3, 3, generator=g) * 0.3, torch.randn(Cout, generator=g)
    xb, wb = x.to(dt).double(), w.to(dt).double()              # the operands the layer sees in the bf16 run mode
    assert lib.uz_conv3x3_first_supported_ref(L.dtype_code(dt), C, Cout) == 1 and lib.uz_conv3x3_first_supported_ref(0, C, Cout) == 0
    rows = lib.uz_conv3x3_first_rows_ref(N, H, W)
    y, stats = np.zeros(N * H * W * Cout, np.uint16), np.zeros(rows * 2 * Cout, np.float32)
    xh, wh, bh = c_ref.host(x), c_ref.host(w), c_ref.host(b)
    assert lib.uz_conv3x3_first_fwd_ref(L.dtype_code(dt), c_ref.ptr(xh), N, C, H, W, c_ref.ptr(wh), c_ref.ptr(bh), Cout, c_ref.ptr(y),
                                        Cout, c_ref.ptr(stats), None) == 0
    ref = F.conv2d(xb, wb, b.double(), padding=1)
    got = nchw(c_ref.tensor(y, dt).reshape(-1, Cout), N, H, W)
    assert torch.equal(got, ref.to(dt))                          # double sums, one rounding: the correctly rounded result
    st = stats.reshape(rows, 2, Cout).sum(0)
    np.testing.assert_allclose(st[0], got.double().sum((0, 2, 3)).numpy(), rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(st[1], (got.double() ** 2).sum((0, 2, 3)).numpy(), rtol=1e-6, atol=1e-4)
    gy = rnd((N, Cout, H, W), dt, g)
    wr = torch.zeros(Cout, C, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xb, wr, padding=1).backward(gy.double())
    dw = np.zeros(Cout * C * 9, np.float32)
    gh = c_ref.host(nhwc(gy))
    assert lib.uz_conv3x3_first_wgrad_workspace_bytes_ref(N, H, W, Cout) == 0
    assert lib.uz_conv3x3_first_wgrad_ref(L.dtype_code(dt), c_ref.ptr(xh), N, C, H, W, c_ref.ptr(gh), Cout, Cout, c_ref.ptr(dw), None, None) == 0
    np.testing.assert_allclose(dw.reshape(Cout, C, 3, 3), wr.grad.numpy(), rtol=1e-5, atol=1e-5)


# ---- window attention (uz_winattn_*): pinned against the formula of swin_unet_v2.py:134-152 written with torch + autograd ------
def _window_attention_torch(qkv, tau, bias, B, H, W, heads, ws, shift, scale):
    """qkv (P, 3C) double -> out (P, C); roll / window_partition / mask / window_reverse as the reference does them
    (swin_unet_v2.py:30-56, :214-238, :246-262), the core as :134-152"""
    C = qkv.shape[1] // 3
    N = ws * ws
    x = qkv.reshape(B, H, W, 3 * C)
    if shift > 0:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    xw = x.reshape(B, H // ws, ws, W // ws, ws, 3 * C).permute(0, 1, 3, 2, 4, 5).reshape(-1, N, 3, heads, 32)
    q, k, v = xw[:, :, 0].transpose(1, 2), xw[:, :, 1].transpose(1, 2), xw[:, :, 2].transpose(1, 2)   # (nWin, heads, N, 32)
    q = q * scale
    attn = (q @ k.transpose(-2, -1)) / torch.clamp(q.norm(dim=-1, keepdim=True) * k.norm(dim=-1, keepdim=True).transpose(-2, -1), min=1e-6)
    attn = attn / torch.clamp(tau[:, :N, :N], min=0.01).unsqueeze(0) + bias.unsqueeze(0)
    if shift > 0:
        img = torch.zeros(1, H, W, 1, dtype=torch.float64)
        cnt = 0
        for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
                img[:, hs, wsl, :] = cnt
                cnt += 1
        mw = img.reshape(1, H // ws, ws, W // ws, ws, 1).permute(0, 1, 3, 2, 4, 5).reshape(-1, N)
        mask = mw.unsqueeze(1) - mw.unsqueeze(2)
        mask = torch.where(mask != 0, torch.full_like(mask, -100.0), torch.zeros_like(mask))
        nW = mask.shape[0]
        attn = (attn.reshape(B, nW, heads, N, N) + mask.reshape(1, nW, 1, N, N)).reshape(-1, heads, N, N)
    p = torch.softmax(attn, dim=-1)
    o = (p @ v).transpose(1, 2).reshape(-1, N, C)                                                      # (nWin, N, C)
    o = o.reshape(B, H // ws, W // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, C)
    if shift > 0:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    return o.reshape(-1, C), torch.logsumexp(attn, dim=-1)


@pytest.mark.parametrize("shift,Nt", [(0, 16), (2, 20)])
def test_window_attention_restatement(shift, Nt):
    lib = c_ref.load()
    g = torch.Generator().manual_seed(41 + shift)
    B, H, W, heads, ws = 2, 8, 8, 2, 4
    C, N, P = 32 * heads, ws * ws, B * H * W
    scale = 32 ** -0.5
    qkv = torch.randn(P, 3 * C, generator=g)
    qkv[5, :C] = 0.0                                   # a zero query row: the 1e-6 clamp of the norm product is active
    tau = torch.rand(heads, Nt, Nt, generator=g) * 0.5 + 0.005      # some entries under the 0.01 clip
    bias = torch.randn(heads, N, N, generator=g) * 0.3
    dout = torch.randn(P, C, generator=g)
    qd = qkv.double().requires_grad_(True)
    td, bd = tau.double().requires_grad_(True), bias.double().requires_grad_(True)
    o_ref, lse_ref = _window_attention_torch(qd, td, bd, B, H, W, heads, ws, shift, scale)
    (o_ref * dout.double()).sum().backward()
    d = L.WinAttnDesc(0, B, H, W, C, heads, ws, shift, Nt, 3 * C, C, scale)
    nwin = B * (H // ws) * (W // ws)
    out, lse = np.zeros(P * C, np.float32), np.zeros(nwin * heads * N, np.float32)
    qh, th, bh = c_ref.host(qkv), c_ref.host(tau), c_ref.host(bias)
    assert lib.uz_winattn_fwd_ref(byref(d), c_ref.ptr(qh), c_ref.ptr(th), c_ref.ptr(bh), c_ref.ptr(out), c_ref.ptr(lse), None) == 0
    np.testing.assert_allclose(out.reshape(P, C), o_ref.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(lse.reshape(nwin, heads, N), lse_ref.detach().numpy(), rtol=1e-5, atol=1e-6)
    assert lib.uz_winattn_bwd_rows_ref(byref(d)) == 1
    dq, part = np.zeros(P * 3 * C, np.float32), np.zeros(2 * heads * N * N, np.float32)
    dh = c_ref.host(dout)
    assert lib.uz_winattn_bwd_ref(byref(d), c_ref.ptr(qh), c_ref.ptr(th), c_ref.ptr(bh), c_ref.ptr(out), c_ref.ptr(lse), c_ref.ptr(dh), C,
                                  c_ref.ptr(dq), 3 * C, c_ref.ptr(part), None) == 0
    np.testing.assert_allclose(dq.reshape(P, 3 * C), qd.grad.numpy(), rtol=2e-4, atol=2e-5)
    part = part.reshape(2, heads, N, N)
    np.testing.assert_allclose(part[0], bd.grad.numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(part[1], td.grad[:, :N, :N].numpy(), rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("dt", DTS)
def test_layernorm_restatement_merge_and_expand_addressing(dt):
    """uz_layernorm_fwd_ref under UZ_LN_MERGE (PatchMerging's strided gather + cat, swin_unet_v2.py:320-326) and UZ_LN_EXPAND
    (PatchExpand / FinalPatchExpand_X4's rearrange, :358, :382) against torch on the materialised tensors"""
    lib = c_ref.load()
    g = torch.Generator().manual_seed(51)
    dc = L.dtype_code(dt)
    npdt = np.uint16 if dt == torch.bfloat16 else np.float32
    N = 2
    # merge: input (N, 2Ho, 2Wo, Cq) -> LayerNorm over 4 Cq channels of the gathered token
    Ho, Wo, Cq = 3, 5, 8
    x = rnd((N, 2 * Ho, 2 * Wo, Cq), dt, g)
    C = 4 * Cq
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    xd = x.double()
    cat = torch.cat([xd[:, 0::2, 0::2], xd[:, 1::2, 0::2], xd[:, 0::2, 1::2], xd[:, 1::2, 1::2]], -1).reshape(-1, C)
    ref = F.layer_norm(cat, (C,), gamma.double(), beta.double(), 1e-5)
    P = N * Ho * Wo
    y, stats = np.zeros(P * C, npdt), np.zeros(2 * P, np.float32)
    xh, gm, bt = c_ref.host(x.reshape(-1, Cq)), c_ref.host(gamma), c_ref.host(beta)
    d = L.LnDesc(dc, N, Ho, Wo, C, Cq, C, 0, 0, 0, L.LN_MERGE, 1, 1e-5, 0)
    assert lib.uz_layernorm_fwd_ref(byref(d), c_ref.ptr(xh), c_ref.ptr(gm), c_ref.ptr(bt), None, None, c_ref.ptr(y), c_ref.ptr(stats), None) == 0
    close(c_ref.tensor(y, dt).reshape(P, C), ref, dt, "layernorm(merge)")
    np.testing.assert_allclose(stats.reshape(P, 2)[:, 0], cat.mean(1).numpy(), rtol=1e-5, atol=1e-6)
    # expand: input (N, h, w, r r C) -> output grid (h r, w r), LayerNorm over C
    for r in (2, 4):
        h, w, C = 2, 3, 16
        x = rnd((N, h, w, r * r * C), dt, g)
        gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
        t = x.double().reshape(N, h, w, r, r, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, C)      # b (h p1) (w p2) c
        ref = F.layer_norm(t, (C,), gamma.double(), beta.double(), 1e-5)
        P = N * h * r * w * r
        y, stats = np.zeros(P * C, npdt), np.zeros(2 * P, np.float32)
        xh, gm, bt = c_ref.host(x.reshape(-1, r * r * C)), c_ref.host(gamma), c_ref.host(beta)
        d = L.LnDesc(dc, N, h * r, w * r, C, r * r * C, C, 0, 0, 0, L.LN_EXPAND, r, 1e-5, 0)
        assert lib.uz_layernorm_fwd_ref(byref(d), c_ref.ptr(xh), c_ref.ptr(gm), c_ref.ptr(bt), None, None, c_ref.ptr(y), c_ref.ptr(stats), None) == 0
        close(c_ref.tensor(y, dt).reshape(P, C), ref, dt, f"layernorm(expand {r})")


@pytest.mark.parametrize("dt", DTS)
def test_attention_gate_forward_restatement(dt):
    """uz_attn_psi_fwd_ref / uz_attn_gate_fwd_ref against AttentionBlock.forward written with torch (attention_unet.py:34-40:
    psi = sigmoid(bn(conv1x1(relu(bn(W_g g) + bn(W_x x))))), out = x * psi), the BatchNorms as (scale, shift) rows"""
    lib = c_ref.load()
    g = torch.Generator().manual_seed(61)
    dc = L.dtype_code(dt)
    npdt = np.uint16 if dt == torch.bfloat16 else np.float32
    P, Fi, C = 150, 24, 40
    g1, x1, x = rnd((P, Fi), dt, g), rnd((P, Fi), dt, g), rnd((P, C), dt, g)
    vg, vx = torch.randn(4, Fi, generator=g), torch.randn(4, Fi, generator=g)
    wpsi, bpsi = torch.randn(Fi, generator=g), torch.randn(1, generator=g)
    vq = torch.tensor([[0.7], [-0.2], [0.0], [1.0]])
    a = F.relu(g1.double() * vg[0].double() + vg[1].double() + x1.double() * vx[0].double() + vx[1].double())
    q_ref = a @ wpsi.double() + bpsi.double()
    q, part = np.zeros(P, np.float32), np.zeros(2, np.float32)
    gh, xh1, vgh, vxh, wh, bh = c_ref.host(g1), c_ref.host(x1), c_ref.host(vg), c_ref.host(vx), c_ref.host(wpsi), c_ref.host(bpsi)
    assert lib.uz_attn_grid_ref(dc, P, Fi) == 1
    assert lib.uz_attn_psi_fwd_ref(dc, c_ref.ptr(gh), Fi, c_ref.ptr(xh1), Fi, c_ref.ptr(vgh), c_ref.ptr(vxh), c_ref.ptr(wh), c_ref.ptr(bh), P, Fi,
                                   c_ref.ptr(q), c_ref.ptr(part), None) == 0
    np.testing.assert_allclose(q, q_ref.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(part, [q_ref.sum().item(), (q_ref ** 2).sum().item()], rtol=1e-5)
    out = np.zeros(P * C, npdt)
    xh, vqh = c_ref.host(x), c_ref.host(vq)
    assert lib.uz_attn_gate_fwd_ref(dc, c_ref.ptr(xh), C, c_ref.ptr(q), c_ref.ptr(vqh), P, C, c_ref.ptr(out), C, None) == 0
    ref = x.double() * torch.sigmoid(torch.from_numpy(q).double() * 0.7 - 0.2)[:, None]
    close(c_ref.tensor(out, dt).reshape(P, C), ref, dt, "attention gate")
