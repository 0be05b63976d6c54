"""GPU: the kernels of libunetzoo_hip.so against the round-5 restatements of oracle/uz_ref.c (pinned on the CPU by
tests/test_c_ref_r5.py) on the same bytes and the same descriptors, called through the C ABI with device pointers.
Element passes must agree to the bit or to one rounding of the tensor type; reductions to accumulation order."""
from ctypes import byref

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import c_ref
from unet_zoo_amd import _lib as L
from unet_zoo_amd import ops
from unet_zoo_amd.ops import Act
from test_c_ref_gpu import DTS, agree, rnd

DEV = "cuda"


def npdt(dt):
    return np.uint16 if dt == torch.bfloat16 else np.float32


def dev(t):
    return t.to(DEV).contiguous()


def ok(rc, what=""):
    assert rc == 0, (what, rc, L.last_error() if hasattr(L, "last_error") else "")


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(8, 64, 128, 64, 64), (5, 120, 125, 32, 64)])     # >= 128 tiles of 16 x 32 pixels: the plan the XF form has
def test_convolution_and_weight_gradient_through_batchnorm_relu_against_the_c_restatement(N, H, W, Cin, Cout):
    """uz_conv_igemm_xf / uz_wgrad_xf (bf16 only: the fp32 run mode has no such kernel) against their restatements"""
    dt = torch.bfloat16
    lib, ref = L.load(), c_ref.load()
    g = torch.Generator().manual_seed(81)
    x = rnd((N * H * W, Cin), dt, g)
    w = rnd((Cout, Cin, 3, 3), torch.float32, g, 0.1)
    bias = torch.randn(Cout, generator=g)
    scale = (torch.rand(Cin, generator=g) + 0.5) * torch.where(torch.rand(Cin, generator=g) < 0.2, -1.0, 1.0)
    shift = torch.rand(Cin, generator=g) * 0.8 + 0.3
    d = L.ConvDesc(L.dtype_code(dt), N, H, W, H, W, Cin, Cin, Cout, Cout, 9, L.TAPS_CONV, 1, L.STORE_PLAIN, 0, 0, 0)
    if not lib.uz_conv_igemm_xf_supported(byref(d)):
        pytest.fail("no input-transform plan for this shape")
    xa = Act(dev(x), 0, Cin, N, H, W)
    wp = ops.pack_weights(dev(w), L.PACK_CONV_FWD, dt)
    ya = ops.new_act(N, H, W, Cout, dt, DEV)
    st = ops.conv_igemm(xa, wp, dev(bias), ya, ntaps=9, want_stats=True, xform=(dev(scale), dev(shift)))
    yr, sr = np.zeros(N * H * W * Cout, np.uint16), np.zeros(2 * Cout, np.float32)
    xh, wh, sch, shh, bh = c_ref.host(x), c_ref.host(wp), c_ref.host(scale), c_ref.host(shift), c_ref.host(bias)
    assert ref.uz_conv_igemm_xf_ref(byref(d), c_ref.ptr(xh), c_ref.ptr(sch), c_ref.ptr(shh), c_ref.ptr(wh), c_ref.ptr(bh), c_ref.ptr(yr),
                                    c_ref.ptr(sr), None) == 0
    agree(ya.buf, c_ref.tensor(yr, dt).reshape(-1, Cout), dt, "conv_xf")
    yd = ya.buf.double().cpu()
    assert torch.allclose(st.double().sum(0).cpu()[0], yd.sum(0), rtol=1e-4, atol=1e-2)

    gy = rnd((N * H * W, Cout), dt, g)
    ga = Act(dev(gy), 0, Cout, N, H, W)
    if W % 32:          # the row-walk weight gradient (and with it its XF form) takes whole 32-pixel strips only
        assert not ops.wgrad_xform_supported(ga, xa, 9)
        return
    assert ops.wgrad_xform_supported(ga, xa, 9)
    dw = torch.empty(Cout, Cin, 3, 3, device=DEV)
    ops.wgrad(ga, xa, (Cout, Cin, 3, 3), ntaps=9, out=dw, xform=(dev(scale), dev(shift)))
    out = np.zeros(Cout * Cin * 9, np.float32)
    dd = L.WgradDesc(L.dtype_code(dt), N, H, W, H, W, Cout, Cout, Cin, Cin, 9, L.TAPS_CONV, 1)
    gh = c_ref.host(gy)
    assert ref.uz_wgrad_xf_ref(byref(dd), c_ref.ptr(gh), c_ref.ptr(xh), c_ref.ptr(sch), c_ref.ptr(shh), c_ref.ptr(out), None, None, 0) == 0
    want = torch.from_numpy(out).reshape(Cout, Cin, 3, 3).double()
    err = ((dw.double().cpu() - want).abs().max() / want.abs().max()).item()
    assert err < 1e-4, err          # same bf16 operands, fp32 accumulation order only


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("ceil_mode", [0, 1])
def test_residual_apply_pool_and_pool_gradient_kernels_against_the_c_restatement(dt, ceil_mode):
    lib, ref = L.load(), c_ref.load()
    g = torch.Generator().manual_seed(82 + ceil_mode)
    dc = L.dtype_code(dt)
    N, H, W, C = 2, 23, 37, 64
    P = N * H * W
    Hp, Wp = ((H + 1) // 2, (W + 1) // 2) if ceil_mode else (H // 2, W // 2)
    y, res = rnd((P, C), dt, g), rnd((P, C), dt, g)
    scale, shift = torch.randn(C, generator=g), torch.randn(C, generator=g)
    yd, rd, sd, hd = dev(y), dev(res), dev(scale), dev(shift)
    act, pooled = torch.zeros(P, C, dtype=dt, device=DEV), torch.zeros(N * Hp * Wp, C, dtype=dt, device=DEV)
    ok(lib.uz_bn_relu_add_apply(dc, yd.data_ptr(), C, sd.data_ptr(), hd.data_ptr(), N, H, W, C, rd.data_ptr(), C, act.data_ptr(), C,
                                pooled.data_ptr(), C, ceil_mode, None))
    ar, pr = np.zeros(P * C, npdt(dt)), np.zeros(N * Hp * Wp * C, npdt(dt))
    yh, rh, sch, shh = c_ref.host(y), c_ref.host(res), c_ref.host(scale), c_ref.host(shift)
    assert ref.uz_bn_relu_add_apply_ref(dc, c_ref.ptr(yh), C, c_ref.ptr(sch), c_ref.ptr(shh), N, H, W, C, c_ref.ptr(rh), C, c_ref.ptr(ar), C,
                                        c_ref.ptr(pr), C, ceil_mode, None) == 0
    torch.cuda.synchronize()
    agree(act, c_ref.tensor(ar, dt).reshape(P, C), dt, "bn_relu_add_apply")
    agree(pooled, c_ref.tensor(pr, dt).reshape(-1, C), dt, "pooled")

    # the gradient router on the kernel's own stored act (ties of the relu's zeros included)
    g0, g1, gp = rnd((P, C), dt, g), rnd((P, C), dt, g), rnd((N * Hp * Wp, C), dt, g)
    out = torch.zeros(P, C, dtype=dt, device=DEV)
    g0d, g1d, gpd = dev(g0), dev(g1), dev(gp)
    ok(lib.uz_pool_grad_combine(dc, N, H, W, C, act.data_ptr(), C, g0d.data_ptr(), C, g1d.data_ptr(), C, gpd.data_ptr(), C, out.data_ptr(), C,
                                ceil_mode, None))
    ah = c_ref.host(act)
    orf = np.zeros(P * C, npdt(dt))
    g0h, g1h, gph = c_ref.host(g0), c_ref.host(g1), c_ref.host(gp)
    assert ref.uz_pool_grad_combine_ref(dc, N, H, W, C, c_ref.ptr(ah), C, c_ref.ptr(g0h), C, c_ref.ptr(g1h), C, c_ref.ptr(gph), C, c_ref.ptr(orf), C,
                                        ceil_mode, None) == 0
    agree(out, c_ref.tensor(orf, dt).reshape(P, C), dt, "pool_grad_combine")


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("align", [0, 1])
def test_bilinear_resize_backward_kernels_against_the_c_restatement(dt, align):
    """both kernels behind uz_resize_bilinear_bwd: the 16-byte vector form (NHWC, C a multiple of the vector) and the wave
    form (one-channel NCHW planes), enlarging and reducing"""
    lib, ref = L.load(), c_ref.load()
    g = torch.Generator().manual_seed(83)
    dc = L.dtype_code(dt)
    for (N, C, Hi, Wi, Ho, Wo) in [(2, 64, 9, 13, 18, 26), (2, 32, 20, 24, 7, 9), (3, 1, 11, 7, 44, 28), (2, 1, 40, 36, 10, 9)]:
        gy = rnd((N * Ho * Wo, C), dt, g)
        gd = dev(gy)
        dx = torch.zeros(N * Hi * Wi, C, dtype=dt, device=DEV)
        ok(lib.uz_resize_bilinear_bwd(dc, gd.data_ptr(), C, Ho * Wo * C, N, Hi, Wi, C, dx.data_ptr(), C, Hi * Wi * C, Ho, Wo, align, None))
        gh = c_ref.host(gy)
        dr = np.zeros(N * Hi * Wi * C, npdt(dt))
        assert ref.uz_resize_bilinear_bwd_ref(dc, c_ref.ptr(gh), C, Ho * Wo * C, N, Hi, Wi, C, c_ref.ptr(dr), C, Hi * Wi * C, Ho, Wo, align, None) == 0
        agree(dx, c_ref.tensor(dr, dt).reshape(-1, C), dt, f"resize bwd {C}ch {Hi}x{Wi}<-{Ho}x{Wo}")     # fp32: the kernel forms the weights in fp32
        if not align:
            dx2 = torch.zeros_like(dx)
            ok(lib.uz_bilinear_bwd(dc, gd.data_ptr(), C, Ho * Wo * C, N, Hi, Wi, C, dx2.data_ptr(), C, Hi * Wi * C, Ho, Wo, None))
            assert torch.equal(dx, dx2)


def test_loss_kernel_against_the_c_restatement():
    lib, ref = L.load(), c_ref.load()
    g = torch.Generator().manual_seed(84)
    for n in (16 * 256 * 256, 12345):
        x, t = torch.randn(n, generator=g) * 3, (torch.rand(n, generator=g) > 0.6).float()
        xd, td = dev(x), dev(t)
        dl, out2 = torch.zeros(n, device=DEV), torch.zeros(2, device=DEV)
        ws = torch.zeros(lib.uz_bce_dice_workspace_bytes(n) // 4 + 1, device=DEV)
        ok(lib.uz_bce_dice(xd.data_ptr(), td.data_ptr(), n, dl.data_ptr(), out2.data_ptr(), ws.data_ptr(), None))
        xh, th = c_ref.host(x), c_ref.host(t)
        dr, o2 = np.zeros(n, np.float32), np.zeros(2, np.float32)
        assert ref.uz_bce_dice_ref(c_ref.ptr(xh), c_ref.ptr(th), n, c_ref.ptr(dr), c_ref.ptr(o2), None, None) == 0
        assert np.allclose(out2.cpu().numpy(), o2, rtol=2e-6)
        assert np.allclose(dl.cpu().numpy(), dr, rtol=1e-5, atol=3e-7 / n)     # (sigmoid - t) / n: fp32 sigmoid next to t = 1


@pytest.mark.parametrize("dt", DTS)
def test_dropout_gate_and_row_sum_kernels_against_the_c_restatement(dt):
    lib, ref = L.load(), c_ref.load()
    g = torch.Generator().manual_seed(85)
    dc = L.dtype_code(dt)
    P, C, p = 777, 96, 0.1
    x, u = rnd((P, C), dt, g), torch.rand(P, C, generator=g)
    xd, ud = dev(x), dev(u)
    out = torch.zeros(P, C, dtype=dt, device=DEV)
    ok(lib.uz_dropout(dc, xd.data_ptr(), C, ud.data_ptr(), p, out.data_ptr(), C, P, C, None))
    xh, uh = c_ref.host(x), c_ref.host(u)
    orf = np.zeros(P * C, npdt(dt))
    assert ref.uz_dropout_ref(dc, c_ref.ptr(xh), C, c_ref.ptr(uh), p, c_ref.ptr(orf), C, P, C, None) == 0
    agree(out, c_ref.tensor(orf, dt).reshape(P, C), dt, "dropout")

    N, HW = 3, 259
    xg, gg = rnd((N * HW, C), dt, g), rnd((N * HW, C), dt, g)
    s, a = torch.rand(N, C, generator=g) + 0.1, torch.randn(N, C, generator=g)
    xgd, ggd, sd, ad = dev(xg), dev(gg), dev(s), dev(a)
    xgh, ggh, sh, ah = c_ref.host(xg), c_ref.host(gg), c_ref.host(s), c_ref.host(a)
    for mode in (2, 0, 1):
        o = torch.zeros(N * HW, C, dtype=dt, device=DEV)
        ok(lib.uz_chanscale_relu(dc, mode, ggd.data_ptr() if mode != 2 else None, C, xgd.data_ptr(), C, sd.data_ptr(),
                                 ad.data_ptr() if mode == 1 else None, N, HW, C, o.data_ptr(), C, None))
        orf = np.zeros(N * HW * C, npdt(dt))
        assert ref.uz_chanscale_relu_ref(dc, mode, c_ref.ptr(ggh) if mode != 2 else None, C, c_ref.ptr(xgh), C, c_ref.ptr(sh),
                                         c_ref.ptr(ah) if mode == 1 else None, N, HW, C, c_ref.ptr(orf), C, None) == 0
        agree(o, c_ref.tensor(orf, dt).reshape(-1, C), dt, f"chanscale_relu mode {mode}")

    if dt == torch.float32:
        rows, n, n0, ld = 300, 200, 128, 333
        part = torch.randn(rows, ld, generator=g)
        pd = dev(part)
        o0, o1 = torch.zeros(n0, device=DEV), torch.zeros(n - n0, device=DEV)
        ok(lib.uz_sum_rows_f32_ld(pd.data_ptr(), ld, rows, n, o0.data_ptr(), n0, o1.data_ptr(), None))
        ph = c_ref.host(part)
        r0, r1 = np.zeros(n0, np.float32), np.zeros(n - n0, np.float32)
        assert ref.uz_sum_rows_f32_ld_ref(c_ref.ptr(ph), ld, rows, n, c_ref.ptr(r0), n0, c_ref.ptr(r1), None) == 0
        assert np.array_equal(o0.cpu().numpy(), r0) and np.array_equal(o1.cpu().numpy(), r1)      # double accumulation: exact
        dense = part[:, :n].contiguous()
        dd = dev(dense)
        o2 = torch.zeros(n, device=DEV)
        ok(lib.uz_sum_rows_f32(dd.data_ptr(), rows, n, o2.data_ptr(), n, None, None))
        assert np.array_equal(o2.cpu().numpy(), np.concatenate([r0, r1]))


@pytest.mark.parametrize("dt", DTS)
def test_input_gather_and_pixel_grid_kernels_against_the_c_restatement(dt):
    """pure data movement: bit for bit"""
    lib, ref = L.load(), c_ref.load()
    g = torch.Generator().manual_seed(86)
    dc = L.dtype_code(dt)
    N, C, H, W, patch, Kpad = 2, 3, 64, 96, 4, 64
    x = torch.randn(N, C, H, W, generator=g)
    xd = dev(x)
    xh = c_ref.host(x)
    rows = N * (H // patch) * (W // patch)
    out = torch.full((rows, Kpad), 7.0, dtype=dt, device=DEV)
    ok(lib.uz_patchify(dc, xd.data_ptr(), N, C, H, W, patch, Kpad, out.data_ptr(), None))
    orf = np.zeros(rows * Kpad, npdt(dt))
    assert ref.uz_patchify_ref(dc, c_ref.ptr(xh), N, C, H, W, patch, Kpad, c_ref.ptr(orf), None) == 0
    assert torch.equal(out.cpu(), c_ref.tensor(orf, dt).reshape(rows, Kpad))

    Hc, Wc, Kp3 = 21, 50, 32
    xs = torch.randn(N, C, Hc, Wc, generator=g)
    xsd, xsh = dev(xs), c_ref.host(xs)
    col = torch.full((N * Hc * Wc, Kp3), 7.0, dtype=dt, device=DEV)
    ok(lib.uz_im2col3x3_nchw(dc, xsd.data_ptr(), N, C, Hc, Wc, Kp3, col.data_ptr(), None))
    crf = np.zeros(N * Hc * Wc * Kp3, npdt(dt))
    assert ref.uz_im2col3x3_nchw_ref(dc, c_ref.ptr(xsh), N, C, Hc, Wc, Kp3, c_ref.ptr(crf), None) == 0
    assert torch.equal(col.cpu(), c_ref.tensor(crf, dt).reshape(-1, Kp3))

    Hs, Ws, Cc = 17, 29, 32
    Hd, Wd = (Hs + 1) // 2, (Ws + 1) // 2
    src = rnd((N * Hs * Ws, Cc), dt, g)
    sd, sh = dev(src), c_ref.host(src)
    for mode, (ha, wa, hb, wb) in ((0, (Hs, Ws, Hs, Ws)), (1, (Hs, Ws, Hd, Wd)), (2, (Hd, Wd, Hs, Ws))):
        s_rows, d_rows = N * ha * wa, N * hb * wb
        o = torch.full((d_rows, Cc), 7.0, dtype=dt, device=DEV)
        ok(lib.uz_resample2(dc, sd.data_ptr(), Cc, N, ha, wa, Cc, o.data_ptr(), Cc, hb, wb, mode, None))
        r = np.zeros(d_rows * Cc, npdt(dt))
        assert ref.uz_resample2_ref(dc, c_ref.ptr(sh), Cc, N, ha, wa, Cc, c_ref.ptr(r), Cc, hb, wb, mode, None) == 0
        assert torch.equal(o.cpu(), c_ref.tensor(r, dt).reshape(d_rows, Cc)), mode
        assert s_rows <= src.shape[0]


@pytest.mark.parametrize("dt", DTS)
def test_attention_gate_backward_kernels_against_the_c_restatement(dt):
    """uz_attn_bwd_psi -> uz_attn_bwd_reduce -> uz_attn_bwd_apply (attention_unet.py:34-40 under autograd), each stage on the
    inputs the kernel chain itself produced"""
    lib, ref = L.load(), c_ref.load()
    g = torch.Generator().manual_seed(87)
    dc = L.dtype_code(dt)
    N, H, W, Fi, C = 2, 24, 40, 32, 64
    P = N * H * W
    g1, x1, x, dout = rnd((P, Fi), dt, g), rnd((P, Fi), dt, g), rnd((P, C), dt, g), rnd((P, C), dt, g)
    q = torch.randn(P, generator=g)
    wpsi = torch.randn(Fi, generator=g) * 0.5

    def rows(v):
        m, var = v.double().mean(0), v.double().var(0, unbiased=False)
        inv = 1 / torch.sqrt(var + 1e-5)
        gam = torch.rand(v.shape[1], generator=g).double() + 0.5
        return torch.stack([gam * inv, 0.1 - m * gam * inv, m, inv]).float()
    vg, vx, vq = rows(g1), rows(x1), rows(q[:, None])
    doa, xa = Act(dev(dout), 0, C, N, H, W), Act(dev(x), 0, C, N, H, W)
    dxd = ops.new_act(N, H, W, C, dt, DEV)
    dz, a01 = ops.attn_bwd_psi(doa, xa, dev(q), dev(vq), dxd)
    h = c_ref.host
    doh, xh, qh, vqh = h(dout), h(x), h(q), h(vq)
    dxr, dzr, pr = np.zeros(P * C, npdt(dt)), np.zeros(P, np.float32), np.zeros(2, np.float32)
    assert ref.uz_attn_bwd_psi_ref(dc, c_ref.ptr(doh), C, c_ref.ptr(xh), C, c_ref.ptr(qh), c_ref.ptr(vqh), P, C, c_ref.ptr(dxr), C, c_ref.ptr(dzr),
                                   c_ref.ptr(pr), None) == 0
    agree(dxd.buf, c_ref.tensor(dxr, dt).reshape(P, C), dt, "dx direct")
    assert np.allclose(dz.cpu().numpy(), dzr, rtol=1e-4, atol=1e-5)
    assert np.allclose(a01.cpu().numpy(), pr, rtol=1e-3, atol=1e-3)

    ga, x1a = Act(dev(g1), 0, Fi, N, H, W), Act(dev(x1), 0, Fi, N, H, W)
    dg, dx1 = ops.new_act(N, H, W, Fi, dt, DEV), ops.new_act(N, H, W, Fi, dt, DEV)
    tot = ops.attn_bwd_branches(ga, x1a, dev(q), dz, dev(wpsi), dev(vg), dev(vx), dev(vq), a01, dg, dx1)
    dzk, a01k = dz.cpu().numpy().copy(), a01.cpu().numpy().astype(np.float64).copy()
    g1h, x1h, wh, vgh, vxh = h(g1), h(x1), h(wpsi), h(vg), h(vx)
    red = np.zeros(4 * Fi + 1, np.float32)
    assert ref.uz_attn_bwd_reduce_ref(dc, c_ref.ptr(g1h), Fi, c_ref.ptr(x1h), Fi, c_ref.ptr(qh), c_ref.ptr(dzk), c_ref.ptr(wh), c_ref.ptr(vgh),
                                      c_ref.ptr(vxh), c_ref.ptr(vqh), c_ref.ptr(a01k), P, Fi, c_ref.ptr(red), None) == 0
    tk = tot.cpu().numpy()
    assert np.allclose(tk, red, rtol=2e-3, atol=2e-3 * np.abs(red).max()), np.abs(tk - red).max()
    totk = tk.astype(np.float64).copy()
    dgr, dxr1 = np.zeros(P * Fi, npdt(dt)), np.zeros(P * Fi, npdt(dt))
    assert ref.uz_attn_bwd_apply_ref(dc, c_ref.ptr(g1h), Fi, c_ref.ptr(x1h), Fi, c_ref.ptr(qh), c_ref.ptr(dzk), c_ref.ptr(wh), c_ref.ptr(vgh),
                                     c_ref.ptr(vxh), c_ref.ptr(vqh), c_ref.ptr(a01k), c_ref.ptr(totk), P, Fi, c_ref.ptr(dgr), Fi,
                                     c_ref.ptr(dxr1), Fi, None) == 0
    agree(dg.buf, c_ref.tensor(dgr, dt).reshape(P, Fi), dt, "d g1raw")
    agree(dx1.buf, c_ref.tensor(dxr1, dt).reshape(P, Fi), dt, "d x1raw")


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("B,Nq,heads,kps,blocks", [(2, 200, 2, 49, 1), (2, 96, 1, 25, 4)])
def test_spatial_reduction_attention_kernels_against_the_c_restatement(dt, B, Nq, heads, kps, blocks):
    ref = c_ref.load()
    g = torch.Generator().manual_seed(88)
    dc = L.dtype_code(dt)
    D = 64
    NK, HD = kps * blocks, heads * D
    scale = D ** -0.5
    q, kv, go = rnd((B * Nq, HD), dt, g), rnd((B * NK, 2 * HD), dt, g), rnd((B * Nq, HD), dt, g)
    qa, kva, goa = Act(dev(q), 0, HD, B, 1, Nq), Act(dev(kv), 0, 2 * HD, B, 1, NK), Act(dev(go), 0, HD, B, 1, Nq)
    out = ops.new_act(B, 1, Nq, HD, dt, DEV)
    lse = ops.sra_fwd(qa, kva, out, B, heads, kps, scale)
    d = L.SraDesc(dc, B, Nq, NK, heads, D, kps, HD, 2 * HD, 2 * HD, HD, scale)
    h = c_ref.host
    qh, kvh = h(q), h(kv)
    orf, lr = np.zeros(B * Nq * HD, npdt(dt)), np.zeros(B * heads * Nq, np.float32)
    vptr = kvh.ctypes.data + HD * kvh.itemsize
    assert ref.uz_sra_fwd_ref(byref(d), c_ref.ptr(qh), c_ref.ptr(kvh), vptr, c_ref.ptr(orf), c_ref.ptr(lr), None) == 0
    want = c_ref.tensor(orf, dt).reshape(-1, HD)
    if dt == torch.float32:
        agree(out.buf, want, dt, "sra forward")
    else:       # the MFMA kernel rounds the probabilities to bf16 for P V (every flash kernel does); the restatement keeps them
        err = ((out.buf.double().cpu() - want.double()).abs().max() / want.double().abs().max()).item()      # in double
        assert err < 1e-2, err
    assert np.allclose(lse.cpu().numpy(), lr, rtol=1e-3, atol=1e-3)          # log2 units of the scaled scores, both

    dq, dkv = ops.new_act(B, 1, Nq, HD, dt, DEV), ops.new_act(B, 1, NK, 2 * HD, dt, DEV)
    ops.sra_bwd(qa, kva, out, lse, goa, dq, dkv, B, heads, kps, scale)
    oh, goh = h(out.buf), h(go)
    lk = lse.cpu().numpy().copy()
    dqr, dkvr = np.zeros(B * Nq * HD, npdt(dt)), np.zeros(B * NK * 2 * HD, npdt(dt))
    assert ref.uz_sra_bwd_ref(byref(d), c_ref.ptr(qh), c_ref.ptr(kvh), vptr, c_ref.ptr(oh), c_ref.ptr(lk), c_ref.ptr(goh), HD, c_ref.ptr(dqr), HD,
                              c_ref.ptr(dkvr), 2 * HD, None, None) == 0
    for got, want, what in ((dq.buf, dqr, "dq"), (dkv.buf, dkvr, "dkv")):
        want = c_ref.tensor(want, dt).reshape(got.shape).double()
        err = ((got.double().cpu() - want).abs().max() / want.abs().max()).item()
        assert err < (1e-4 if dt == torch.float32 else 2e-2), (what, err)


def test_batched_row_sums_equal_the_single_launches_bit_for_bit():
    """uz_sum_rows_f32_batched: every buffer summed with the arithmetic uz_sum_rows_f32 uses for it alone -- the column form
    (LayerNorm dgamma | dbeta rows) and both wide forms (window attention's d(bias) | d(tau) rows: 16 and 4 row groups) --
    so a parameter gradient does not depend on which launch formed it; also against the C restatement (double sums)"""
    ref = c_ref.load()
    g = torch.Generator().manual_seed(89)
    shapes = [(147, 24576, 12288), (64, 49152, 24576), (32, 98304, 49152), (16, 196608, 98304),   # swin_unet_v2's four levels
              (512, 192, 96), (300, 200, 128), (40, 16384, 16384), (256, 20000, 20000), (257, 16384, 8192)]
    items, singles = [], []
    for rows, n, n0 in shapes:
        part = dev(torch.randn(rows, n, generator=g))
        o0, o1 = torch.zeros(n0, device=DEV), (torch.zeros(n - n0, device=DEV) if n0 < n else None)
        s0, s1 = torch.zeros(n0, device=DEV), (torch.zeros(n - n0, device=DEV) if n0 < n else None)
        ops.sum_rows_f32(part, rows, s0, s1)
        items.append((part, rows, o0, o1))
        singles.append((s0, s1))
    ops.sum_rows_f32_batched(items)
    torch.cuda.synchronize()
    for (part, rows, o0, o1), (s0, s1), (_, n, n0) in zip(items, singles, shapes):
        assert torch.equal(o0, s0), (rows, n)
        if o1 is not None:
            assert torch.equal(o1, s1), (rows, n)
        ph = c_ref.host(part)
        r0, r1 = np.zeros(n0, np.float32), np.zeros(max(n - n0, 1), np.float32)
        assert ref.uz_sum_rows_f32_ref(c_ref.ptr(ph), rows, n, c_ref.ptr(r0), n0, c_ref.ptr(r1) if n0 < n else None, None) == 0
        assert np.array_equal(o0.cpu().numpy(), r0)
        if o1 is not None:
            assert np.array_equal(o1.cpu().numpy(), r1[:n - n0])


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("ws,shift,Nt", [(8, 4, 64), (8, 0, 64), (7, 3, 49)])
def test_window_attention_backward_follows_the_norm_clamp_of_the_reference(dt, ws, shift, Nt):
    """swin_unet_v2.py:137-139 divides q k^T by max(|q||k|, 1e-6): a pair under the clamp keeps u / 1e-6 and has no projection
    term in its gradient.  Rounds 1-4: the bf16 kernel was exact for zero rows only.  Here image 0 has every key at 1e-2 of
    its size and three queries at 1e-5 -- norm products of ~5e-7, NOT zero -- beside ordinary rows and a zero row; dqkv
    (whose largest entries are exactly those rows': the clamp's 1e6 factor) against the restatement, which follows the
    clamp pair by pair (pinned against torch autograd of the reference's formula, tests/test_c_ref.py)."""
    lib = c_ref.load()
    g = torch.Generator().manual_seed(7 * ws + shift)
    B, heads = 2, 3
    H = W = 2 * ws
    C, N, P = 32 * heads, ws * ws, B * H * W
    qkv = rnd((P, 3 * C), torch.float32, g)
    qkv[:H * W, C:2 * C] *= 1e-2                       # image 0: small keys
    for t in (3, 17, 40, H * W - 1):
        qkv[t, :C] *= 1e-5                             # ... and a few tiny (non-zero) queries: |scale q||k| ~ 5e-7
    qkv[H * W + 5, :C] = 0                             # image 1: a zero query row
    qkv = qkv.to(dt)
    nq = qkv[3, :32].float().norm() * 32 ** -0.5
    nk = qkv[4, C:C + 32].float().norm()
    assert 0 < nq * nk < 1e-6
    tau = torch.rand(heads, Nt, Nt, generator=g) * 0.5 + 0.05
    bias = torch.randn(heads, N, N, generator=g) * 0.3
    dout = rnd((P, C), dt, g)
    qa, da = Act(qkv.to(DEV), 0, 3 * C, B, H, W), Act(dout.to(DEV), 0, C, B, H, W)
    out, dq = ops.new_act(B, H, W, C, dt, DEV), ops.new_act(B, H, W, 3 * C, dt, DEV)
    lse = ops.winattn_fwd(qa, tau.to(DEV), bias.to(DEV), out, heads, ws, shift)
    ops.winattn_bwd(qa, tau.to(DEV), bias.to(DEV), out, lse, da, dq, heads, ws, shift)
    d = L.WinAttnDesc(L.dtype_code(dt), B, H, W, C, heads, ws, shift, Nt, 3 * C, C, 32 ** -0.5)
    nwin = B * (H // ws) * (W // ws)
    o_r, l_r = np.zeros(P * C, npdt(dt)), np.zeros(nwin * heads * N, np.float32)
    qh, th, bh, dh = c_ref.host(qkv), c_ref.host(tau), c_ref.host(bias), c_ref.host(dout)
    assert lib.uz_winattn_fwd_ref(byref(d), c_ref.ptr(qh), c_ref.ptr(th), c_ref.ptr(bh), c_ref.ptr(o_r), c_ref.ptr(l_r), None) == 0
    dq_r, part = np.zeros(P * 3 * C, npdt(dt)), np.zeros(2 * heads * N * N, np.float32)
    assert lib.uz_winattn_bwd_ref(byref(d), c_ref.ptr(qh), c_ref.ptr(th), c_ref.ptr(bh), c_ref.ptr(o_r), c_ref.ptr(l_r), c_ref.ptr(dh), C,
                                  c_ref.ptr(dq_r), 3 * C, c_ref.ptr(part), None) == 0
    got, want = dq.buf.double().cpu(), c_ref.tensor(dq_r, dt).reshape(P, 3 * C).double()
    tol = 1e-4 if dt == torch.float32 else 3e-2
    # the tiny queries' gradient rows (d q of image 0's rows 3, 17, 40, last) carry the 1e6 factor: compare them by themselves,
    # then everything else by itself (a global maximum would let either hide behind the other)
    tiny = torch.zeros(P, dtype=torch.bool)
    tiny[[3, 17, 40, H * W - 1]] = True
    for name, rows, cols in (("dq of the clamped queries", tiny, slice(0, C)), ("dk of image 0", torch.arange(P) < H * W, slice(C, 2 * C)),
                             ("the rest", ~tiny, slice(0, 3 * C))):
        a, b = got[rows][:, cols], want[rows][:, cols]
        err = ((a - b).abs().max() / b.abs().max()).item()
        assert err <= tol, (name, err, b.abs().max().item())


@pytest.mark.parametrize("Co,Ci,bias_res", [(96, 192, False), (40, 64, True), (24, 32, False), (64, 64, True)])
def test_pixel_shuffle_store_with_a_sub_pixel_boundary_inside_a_tile(Co, Ci, bias_res):
    """uz_conv_igemm with UZ_STORE_SHUFFLE2X2 where Co is a multiple of 8 but not of the kernel's 64 / 128-channel tile
    (PatchExpand with Co = 96, swin_unet_v2.py:343-351: it ran on the round-1 kernel until the LDS-DMA GEMM learned to decide
    the sub-pixel per 16-byte chunk): the kernel family is asserted, the result held against the restatement; with bias and
    the residual form (uz_conv_igemm_res) where given"""
    ref = c_ref.load()
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(Co)
    N, H, W = 2, 24, 40
    x = rnd((N * H * W, Ci), dt, g)
    wt = rnd((Ci, Co, 2, 2), torch.float32, g, 0.2)
    bias = torch.randn(Co, generator=g) if bias_res else None
    xa = Act(dev(x), 0, Ci, N, H, W)
    wpt = ops.pack_weights(dev(wt), L.PACK_CONVT_FWD, dt)
    d = L.ConvDesc(L.dtype_code(dt), N, H, W, H, W, Ci, Ci, 4 * Co, Co, 1, L.TAPS_CONV, 1, L.STORE_SHUFFLE2X2, Co, 0, 0)
    assert ops.conv_kernel_name(d).startswith("gemm_dma"), ops.conv_kernel_name(d)
    yt = ops.new_act(N, 2 * H, 2 * W, Co, dt, DEV)
    b4 = bias.repeat(4) if bias is not None else None          # the GEMM's bias: one value per (sub-pixel, channel) column
    ops.conv_igemm(xa, wpt, dev(b4) if b4 is not None else None, yt, ntaps=1, store_mode=L.STORE_SHUFFLE2X2, nout=4 * Co, co=Co)
    yr = np.zeros(N * 4 * H * W * Co, np.uint16)
    xh, wh = c_ref.host(x), c_ref.host(wpt)
    bh = c_ref.host(b4) if b4 is not None else None
    assert ref.uz_conv_igemm_ref(byref(d), c_ref.ptr(xh), c_ref.ptr(wh), c_ref.ptr(bh), c_ref.ptr(yr), None, None) == 0
    agree(yt.buf, c_ref.tensor(yr, dt).reshape(-1, Co), dt, f"convT forward, Co = {Co}")


@pytest.mark.parametrize("dt", DTS)
def test_batched_weight_packing_of_transposed_convolutions_equals_the_single_calls(dt):
    """uz_pack_weights_batched: the ConvTranspose2d(k2, s2) fast paths (row permutation for the input-gradient layout, LDS
    transpose with permuted destination rows for the forward layout) against uz_pack_weights' element-wise gather and the C
    restatement, bit for bit; odd channel counts and tails of the 64 x 64 tiles included; mixed with other items in one launch"""
    import ctypes
    lib, ref = L.load(), c_ref.load()
    g = torch.Generator().manual_seed(90)
    shapes = [(1024, 512), (128, 64), (96, 40), (8, 8), (72, 100)]
    ws = [dev(torch.randn(ci, co, 2, 2, generator=g)) for ci, co in shapes]
    lin = dev(torch.randn(48, 80, generator=g))                # a one-tap item between them (its own fast path)
    items, outs, begin = [], [], 0
    for w, (ci, co) in zip(ws, shapes):
        for mode, shape in ((L.PACK_CONVT_FWD, (4 * co, ci)), (L.PACK_CONVT_DGRAD, (ci, 4 * co))):
            dst = torch.full(shape, 7.0, dtype=dt, device=DEV)
            items.append(L.PackItem(w.data_ptr(), dst.data_ptr(), begin, mode, co, ci, 4, 0, 0))
            outs.append((w, mode, dst))
            begin += dst.numel()
    dl = torch.full((48, 80), 7.0, dtype=dt, device=DEV)
    items.insert(3, L.PackItem(lin.data_ptr(), dl.data_ptr(), 0, L.PACK_CONV_FWD, 48, 80, 1, 0, 0))
    begin = 0
    arr = (L.PackItem * len(items))()
    for i, it in enumerate(items):                             # `begin` is the prefix sum over the items in launch order
        it.begin = begin
        arr[i] = it
        begin += (48 * 80) if i == 3 else outs[i if i < 3 else i - 1][2].numel()
    tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(DEV)
    ok(lib.uz_pack_weights_batched(L.dtype_code(dt), tab.data_ptr(), len(items), begin, None))
    torch.cuda.synchronize()
    for w, mode, dst in outs:
        single = ops.pack_weights(w, mode, dt)
        assert torch.equal(dst, single), (tuple(w.shape), mode)
        ci, co = w.shape[0], w.shape[1]
        r = np.zeros(dst.numel(), npdt(dt))
        wh = c_ref.host(w)
        assert ref.uz_pack_weights_ref(L.dtype_code(dt), mode, c_ref.ptr(wh), co, ci, 4, 0, c_ref.ptr(r), None) == 0
        assert torch.equal(dst.cpu(), c_ref.tensor(r, dt).reshape(dst.shape))
    assert torch.equal(dl, ops.pack_weights(lin.view(48, 80, 1, 1), L.PACK_CONV_FWD, dt).view(48, 80))


@pytest.mark.parametrize("N,H,W,C,K", [(2, 32, 48, 64, 1), (3, 17, 19, 40, 3), (1, 8, 8, 512, 8)])
def test_head_through_batchnorm_relu_against_the_c_restatement_and_the_two_pass_form(N, H, W, C, K):
    """uz_outconv_fwd_xf against its restatement and, bit for bit, against uz_bn_relu_apply + uz_outconv_fwd; then
    uz_outconv_bwd_bnred with x = NULL (the activation formed from the raw tensor) bit for bit against the same call with
    the stored activation: dx, dW, db and the BatchNorm-backward partial rows"""
    dt = torch.bfloat16
    lib, ref = L.load(), c_ref.load()
    g = torch.Generator().manual_seed(83)
    y = rnd((N * H * W, C), dt, g)
    w, b = torch.randn(K, C, generator=g) * 0.3, torch.randn(K, generator=g)
    scale = (torch.rand(C, generator=g) + 0.5) * torch.where(torch.rand(C, generator=g) < 0.3, -1.0, 1.0)
    shift = torch.randn(C, generator=g) * 0.5
    mean, invstd = torch.randn(C, generator=g) * 0.2, torch.rand(C, generator=g) + 0.5
    ya = Act(dev(y), 0, C, N, H, W)
    sc, sh = dev(scale), dev(shift)
    got = ops.outconv_fwd(ya, dev(w), dev(b), xform=(sc, sh))
    out = np.zeros(N * K * H * W, np.float32)
    yh, sch, shh, wh, bh = c_ref.host(y), c_ref.host(scale), c_ref.host(shift), c_ref.host(w), c_ref.host(b)
    assert ref.uz_outconv_fwd_xf_ref(L.dtype_code(dt), c_ref.ptr(yh), C, N, H * W, C, c_ref.ptr(sch), c_ref.ptr(shh), c_ref.ptr(wh),
                                     c_ref.ptr(bh), K, c_ref.ptr(out), None) == 0
    want = torch.from_numpy(out).reshape(N, K, H, W)
    assert ((got.cpu().double() - want.double()).abs().max() / want.abs().max()).item() < 1e-5     # fp32 sum order over C terms
    act = ops.new_act(N, H, W, C, dt, DEV)
    ops.bn_relu_apply(ya, sc, sh, act, None, None, False)
    assert torch.equal(got, ops.outconv_fwd(act, dev(w), dev(b)))

    gl = dev(torch.randn(N, K, H, W, generator=g))
    vec = (sc, sh, dev(mean), dev(invstd))
    res = []
    for lazy in (False, True):
        dx = ops.new_act(N, H, W, C, dt, DEV)
        dw, db = ops.outconv_bwd(act, dev(w), gl, dx, bnred=(ya, vec), lazy=lazy)
        res.append((dx.buf.clone(), dw.clone(), db.clone(), dx.bn_partials.clone()))
    for a, c in zip(*res):
        assert torch.equal(a, c)
