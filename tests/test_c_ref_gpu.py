"""GPU: the kernels of libunetzoo_hip.so against their plain-C restatement (`<entry>_ref`, oracle/uz_ref.c, pinned on the CPU
by tests/test_c_ref.py) on the same bytes and the same descriptors.  bf16 results may differ by one rounding of the last
bit where the kernel's fp32 accumulation order lands on the other side of a tie; fp32 by accumulation order."""
import math
from ctypes import byref

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import c_ref
from unet_zoo_amd import _lib as L
from unet_zoo_amd import ops
from unet_zoo_amd.ops import Act

DEV = "cuda"
DTS = [torch.float32, torch.bfloat16]


def rnd(shape, dt, g, scale=1.0):
    return (scale * torch.randn(*shape, generator=g)).to(dt)


def agree(got: torch.Tensor, ref: torch.Tensor, dt, what, f32_tol=1e-4):     # fp32: accumulation order over K <= 1152 terms
    got, ref = got.detach().cpu().double(), ref.double()
    den = ref.abs() + 1e-2 * ref.abs().max() + 1e-30
    err = ((got - ref).abs() / den).max().item()
    assert err <= (f32_tol if dt == torch.float32 else 2.0 ** -7), (what, err)
    if dt == torch.bfloat16:      # and nearly all elements are the same bf16 number
        same = (got == ref).double().mean().item()
        assert same > 0.97, (what, same)


def ref_conv(dt, x_rows, w_packed, bias, desc_args, P_out, ldy, Nout):
    lib = c_ref.load()
    y = np.zeros(P_out * ldy, dtype=np.uint16 if dt == torch.bfloat16 else np.float32)
    d = L.ConvDesc(*desc_args)
    stats = np.zeros(2 * Nout, dtype=np.float32)
    xh, wh = c_ref.host(x_rows), c_ref.host(w_packed)
    bh = c_ref.host(bias) if bias is not None else None
    assert lib.uz_conv_igemm_ref(byref(d), c_ref.ptr(xh), c_ref.ptr(wh), c_ref.ptr(bh), c_ref.ptr(y), c_ref.ptr(stats), None) == 0
    return c_ref.tensor(y, dt).reshape(P_out, ldy), stats


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("N,H,W,Ci,Co,dil", [(2, 32, 32, 64, 64, 1), (1, 16, 32, 128, 128, 1), (2, 16, 16, 64, 64, 2), (1, 9, 11, 32, 40, 1)])
def test_conv3x3_kernels_against_the_c_restatement(dt, N, H, W, Ci, Co, dil):
    g = torch.Generator().manual_seed(H + Ci)
    x = rnd((N * H * W, Ci), dt, g)
    w = rnd((Co, Ci, 3, 3), torch.float32, g, 0.1)
    bias = torch.randn(Co, generator=g)
    wp = ops.pack_weights(w.to(DEV), L.PACK_CONV_FWD, dt)
    xa = Act(x.to(DEV), 0, Ci, N, H, W)
    ya = ops.new_act(N, H, W, Co, dt, DEV)
    stats = ops.conv_igemm(xa, wp, bias.to(DEV), ya, ntaps=9, dil=dil, want_stats=True)
    yr, sr = ref_conv(dt, x, wp.cpu(), bias, (L.dtype_code(dt), N, H, W, H, W, Ci, Ci, Co, Co, 9, L.TAPS_CONV, dil, 0, 0, 0, 0), N * H * W, Co, Co)
    agree(ya.buf, yr, dt, ops.conv_kernel_name(L.ConvDesc(L.dtype_code(dt), N, H, W, H, W, Ci, Ci, Co, Co, 9, L.TAPS_CONV, dil, 0, 0, 0, 0)))
    # statistics of the kernel's own stored output (bit-level differences of y are excluded that way)
    yd = ya.buf.double().cpu()
    s = stats.double().sum(0).cpu()
    assert torch.allclose(s[0], yd.sum(0), rtol=1e-4, atol=1e-2) and torch.allclose(s[1], (yd * yd).sum(0), rtol=1e-4, atol=1e-2)
    assert np.allclose(sr[:Co], yr.double().sum(0).numpy(), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("dt", DTS)
def test_pointwise_upsampled_and_transposed_convolutions_against_the_c_restatement(dt):
    g = torch.Generator().manual_seed(12)
    N, H, W, Ci, Co = 2, 8, 12, 64, 64
    x = rnd((N * H * W, Ci), dt, g)
    xa = Act(x.to(DEV), 0, Ci, N, H, W)
    dc = L.dtype_code(dt)
    # 1x1
    w1 = rnd((Co, Ci, 1, 1), torch.float32, g, 0.2)
    wp = ops.pack_weights(w1.to(DEV), L.PACK_CONV_FWD, dt)
    ya = ops.new_act(N, H, W, Co, dt, DEV)
    ops.conv_igemm(xa, wp, None, ya, ntaps=1)
    yr, _ = ref_conv(dt, x, wp.cpu(), None, (dc, N, H, W, H, W, Ci, Ci, Co, Co, 1, L.TAPS_CONV, 1, 0, 0, 0, 0), N * H * W, Co, Co)
    agree(ya.buf, yr, dt, "1x1")
    # 3x3 on the nearest x2 upsampling (common_layers.py:69-72)
    w3 = rnd((Co, Ci, 3, 3), torch.float32, g, 0.1)
    wp3 = ops.pack_weights(w3.to(DEV), L.PACK_CONV_FWD, dt)
    yu = ops.new_act(N, 2 * H, 2 * W, Co, dt, DEV)
    ops.conv_igemm(xa, wp3, None, yu, ntaps=9, taps_mode=L.TAPS_CONV_UP2)
    yr, _ = ref_conv(dt, x, wp3.cpu(), None, (dc, N, 2 * H, 2 * W, H, W, Ci, Ci, Co, Co, 9, L.TAPS_CONV_UP2, 1, 0, 0, 0, 0), N * 4 * H * W, Co, Co)
    agree(yu.buf, yr, dt, "up2")
    # ConvTranspose2d k2 s2 forward: pixel-shuffle store (common_layers.py:104)
    wt = rnd((Ci, Co, 2, 2), torch.float32, g, 0.2)
    wpt = ops.pack_weights(wt.to(DEV), L.PACK_CONVT_FWD, dt)
    yt = ops.new_act(N, 2 * H, 2 * W, Co, dt, DEV)
    ops.conv_igemm(xa, wpt, None, yt, ntaps=1, store_mode=L.STORE_SHUFFLE2X2, nout=4 * Co, co=Co)
    yr, _ = ref_conv(dt, x, wpt.cpu(), None, (dc, N, H, W, H, W, Ci, Ci, 4 * Co, Co, 1, L.TAPS_CONV, 1, L.STORE_SHUFFLE2X2, Co, 0, 0), N * 4 * H * W, Co, 4 * Co)
    agree(yt.buf, yr, dt, "convT fwd")


@pytest.mark.parametrize("dt", DTS)
def test_weight_gradient_kernels_against_the_c_restatement(dt):
    lib = c_ref.load()
    g = torch.Generator().manual_seed(13)
    N, H, W, Ci, Co = 2, 16, 16, 64, 128
    x, gy = rnd((N * H * W, Ci), dt, g), rnd((N * H * W, Co), dt, g)
    out = ops.wgrad(Act(gy.to(DEV), 0, Co, N, H, W), Act(x.to(DEV), 0, Ci, N, H, W), (Co, Ci, 3, 3), ntaps=9)
    ref = np.zeros(Co * Ci * 9, np.float32)
    d = L.WgradDesc(L.dtype_code(dt), N, H, W, H, W, Co, Co, Ci, Ci, 9, L.TAPS_CONV, 1)
    Lh, Rh = c_ref.host(gy), c_ref.host(x)
    assert lib.uz_wgrad_ref(byref(d), c_ref.ptr(Lh), c_ref.ptr(Rh), c_ref.ptr(ref), None, None) == 0
    r = torch.from_numpy(ref).reshape(Co, Ci, 3, 3).double()
    err = ((out.cpu().double() - r).abs().max() / r.abs().max()).item()
    assert err < 1e-5, err


@pytest.mark.parametrize("N,H,W,Cx,Cdy,mode", [
    (2, 8, 64, 64, 64, "conv"),        # row-walk kernel, 64 x 64 tile (uz_wgrad9.hip)
    (1, 8, 128, 128, 128, "conv"),     # row-walk kernel, 128 x 64 tile, two strips per row
    (2, 16, 16, 72, 136, "conv"),      # W = 16, channel tails
    (1, 16, 32, 64, 64, "up2"),        # x through nearest x2 upsampling (x lives at 8 x 16)
    (2, 8, 16, 128, 64, "convt"),      # ConvTranspose2d k2 s2: four taps per workgroup (uz_wgrad_g4.hip)
    (1, 4, 64, 64, 40, "convt"),
])
def test_round4_weight_gradient_kernels_against_the_c_restatement(N, H, W, Cx, Cdy, mode):
    """the row-walk nine-tap kernel and the four-tap 2 x 2 gather kernel (bf16) against uz_wgrad_ref on the same bytes"""
    dt = torch.bfloat16
    lib = c_ref.load()
    g = torch.Generator().manual_seed(23)
    if mode == "convt":     # L = x on the coarse grid, R = the output gradient on the fine grid
        Lt, Rt = rnd((N * H * W, Cx), dt, g), rnd((N * 4 * H * W, Cdy), dt, g)
        La, Ra = Act(Lt.to(DEV), 0, Cx, N, H, W), Act(Rt.to(DEV), 0, Cdy, N, 2 * H, 2 * W)
        shape, nt, tm = (Cx, Cdy, 2, 2), 4, L.TAPS_GATHER2X2
        d = L.WgradDesc(L.dtype_code(dt), N, H, W, 2 * H, 2 * W, Cx, Cx, Cdy, Cdy, 4, tm, 1)
    else:
        hx, wx = (H // 2, W // 2) if mode == "up2" else (H, W)
        Lt, Rt = rnd((N * H * W, Cdy), dt, g), rnd((N * hx * wx, Cx), dt, g)
        La, Ra = Act(Lt.to(DEV), 0, Cdy, N, H, W), Act(Rt.to(DEV), 0, Cx, N, hx, wx)
        tm = L.TAPS_CONV_UP2 if mode == "up2" else L.TAPS_CONV
        shape, nt = (Cdy, Cx, 3, 3), 9
        d = L.WgradDesc(L.dtype_code(dt), N, H, W, hx, wx, Cdy, Cdy, Cx, Cx, 9, tm, 1)
    name = ops.wgrad_kernel_name(d)
    assert name.startswith("wgrad_g4_") if mode == "convt" else name.startswith("wgrad9_"), name
    out = ops.wgrad(La, Ra, shape, ntaps=nt, taps_mode=tm)
    ref = np.zeros(out.numel(), np.float32)
    Lh, Rh = c_ref.host(Lt), c_ref.host(Rt)
    assert lib.uz_wgrad_ref(byref(d), c_ref.ptr(Lh), c_ref.ptr(Rh), c_ref.ptr(ref), None, None) == 0
    r = torch.from_numpy(ref).reshape(shape).double()
    err = ((out.cpu().double() - r).abs().max() / r.abs().max()).item()
    assert err < 1e-5, (name, err)


@pytest.mark.parametrize("N,C,H,W,Cout", [(2, 3, 16, 64, 64), (1, 2, 21, 50, 32)])
def test_first_convolution_kernels_against_the_c_restatement(N, C, H, W, Cout):
    """uz_conv3x3_first_fwd / _wgrad (uz_conv_first.hip) against their restatements on the same bytes"""
    dt = torch.bfloat16
    lib = c_ref.load()
    g = torch.Generator().manual_seed(29)
    x, w, b = torch.randn(N, C, H, W, generator=g), torch.randn(Cout, C, 3, 3, generator=g) * 0.3, torch.randn(Cout, generator=g)
    y = ops.new_act(N, H, W, Cout, dt, DEV)
    stats = ops.conv_first_fwd(x.to(DEV), w.to(DEV), b.to(DEV), y, True)
    yr, sr = np.zeros(N * H * W * Cout, np.uint16), np.zeros(2 * Cout, np.float32)
    xh, wh, bh = c_ref.host(x), c_ref.host(w), c_ref.host(b)
    assert lib.uz_conv3x3_first_fwd_ref(L.dtype_code(dt), c_ref.ptr(xh), N, C, H, W, c_ref.ptr(wh), c_ref.ptr(bh), Cout, c_ref.ptr(yr),
                                        Cout, c_ref.ptr(sr), None) == 0
    agree(y.buf, c_ref.tensor(yr, dt).reshape(-1, Cout), dt, "first conv")
    s = stats.double().sum(0).cpu().numpy()
    np.testing.assert_allclose(s.reshape(-1), sr.astype(np.float64), rtol=2e-3, atol=0.5)   # sums of the stored (once-rounded) values
    gy = rnd((N * H * W, Cout), dt, g)
    dw = ops.conv_first_wgrad(x.to(DEV), Act(gy.to(DEV), 0, Cout, N, H, W))
    dr = np.zeros(Cout * C * 9, np.float32)
    gh = c_ref.host(gy)
    assert lib.uz_conv3x3_first_wgrad_ref(L.dtype_code(dt), c_ref.ptr(xh), N, C, H, W, c_ref.ptr(gh), Cout, Cout, c_ref.ptr(dr), None, None) == 0
    r = torch.from_numpy(dr).reshape(Cout, C, 3, 3).double()
    assert ((dw.cpu().double() - r).abs().max() / r.abs().max()).item() < 1e-5


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("ws,shift,Nt", [(8, 4, 64), (4, 0, 16), (7, 3, 49)])
def test_window_attention_kernels_against_the_c_restatement(dt, ws, shift, Nt):
    """uz_winattn_fwd / uz_winattn_bwd (fp32: the VALU kernels; bf16: the matrix-core kernels of round 4) against the
    restatement pinned to torch autograd on the CPU (tests/test_c_ref.py): outputs, row log-sum-exp, dqkv, d(bias), d(tau);
    a zero query row (the 1e-6 clamp of the norm product), tau entries under the 0.01 clip, the shifted-window mask"""
    lib = c_ref.load()
    g = torch.Generator().manual_seed(100 * ws + shift)
    B, heads = 2, 3
    H = W = 2 * ws
    C, N, P = 32 * heads, ws * ws, B * H * W
    qkv = rnd((P, 3 * C), dt, g)
    qkv[5, :C] = 0
    tau = torch.rand(heads, Nt, Nt, generator=g) * 0.5 + 0.005
    bias = torch.randn(heads, N, N, generator=g) * 0.3
    dout = rnd((P, C), dt, g)
    qa, da = Act(qkv.to(DEV), 0, 3 * C, B, H, W), Act(dout.to(DEV), 0, C, B, H, W)
    out, dq = ops.new_act(B, H, W, C, dt, DEV), ops.new_act(B, H, W, 3 * C, dt, DEV)
    lse = ops.winattn_fwd(qa, tau.to(DEV), bias.to(DEV), out, heads, ws, shift)
    dbias, dtau = ops.winattn_bwd(qa, tau.to(DEV), bias.to(DEV), out, lse, da, dq, heads, ws, shift)
    npdt = np.uint16 if dt == torch.bfloat16 else np.float32
    d = L.WinAttnDesc(L.dtype_code(dt), B, H, W, C, heads, ws, shift, Nt, 3 * C, C, 32 ** -0.5)
    nwin = B * (H // ws) * (W // ws)
    o_r, l_r = np.zeros(P * C, npdt), np.zeros(nwin * heads * N, np.float32)
    qh, th, bh, dh = c_ref.host(qkv), c_ref.host(tau), c_ref.host(bias), c_ref.host(dout)
    assert lib.uz_winattn_fwd_ref(byref(d), c_ref.ptr(qh), c_ref.ptr(th), c_ref.ptr(bh), c_ref.ptr(o_r), c_ref.ptr(l_r), None) == 0
    dq_r, part = np.zeros(P * 3 * C, npdt), np.zeros(2 * heads * N * N, np.float32)
    assert lib.uz_winattn_bwd_ref(byref(d), c_ref.ptr(qh), c_ref.ptr(th), c_ref.ptr(bh), c_ref.ptr(o_r), c_ref.ptr(l_r), c_ref.ptr(dh), C,
                                  c_ref.ptr(dq_r), 3 * C, c_ref.ptr(part), None) == 0
    tol = 1e-4 if dt == torch.float32 else 3e-2      # bf16: P and W enter the second products rounded to bf16

    def near(got, ref, what):
        got, ref = got.detach().cpu().double(), ref.double()
        err = ((got - ref).abs().max() / ref.abs().max()).item()
        assert err <= tol, (what, err)
    near(out.buf, c_ref.tensor(o_r, dt).reshape(P, C), "out")
    near(lse.reshape(-1), torch.from_numpy(l_r), "lse") if dt == torch.float32 else None
    near(dq.buf, c_ref.tensor(dq_r, dt).reshape(P, 3 * C), "dqkv")
    part = torch.from_numpy(part).reshape(2, heads, N, N)
    near(dbias, part[0], "dbias")
    near(dtau, part[1], "dtau")


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("mode,r,N,Ho,Wo,C", [("merge", 1, 2, 14, 14, 384), ("merge", 1, 1, 7, 9, 192), ("expand", 2, 2, 16, 16, 96),
                                              ("expand", 4, 1, 32, 24, 96)])
def test_layernorm_addressing_modes_against_the_c_restatement(dt, mode, r, N, Ho, Wo, C):
    """uz_layernorm_fwd with PatchMerging's gather (UZ_LN_MERGE) and PatchExpand's rearrange (UZ_LN_EXPAND, r = 2 / 4) as
    addressing modes, against the restatement pinned to torch on the CPU (swin_unet_v2.py:320-326, :358, :382)"""
    lib = c_ref.load()
    g = torch.Generator().manual_seed(Ho + C + r)
    dc = L.dtype_code(dt)
    npdt = np.uint16 if dt == torch.bfloat16 else np.float32
    if mode == "merge":
        xin = rnd((N * 2 * Ho * 2 * Wo, C // 4), dt, g)
        xa = Act(xin.to(DEV), 0, C // 4, N, 2 * Ho, 2 * Wo)
        m, ldx = L.LN_MERGE, C // 4
    else:
        xin = rnd((N * (Ho // r) * (Wo // r), r * r * C), dt, g)
        xa = Act(xin.to(DEV), 0, r * r * C, N, Ho // r, Wo // r)
        m, ldx = L.LN_EXPAND, r * r * C
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    y = ops.new_act(N, Ho, Wo, C, dt, DEV)
    stats = ops.layernorm_fwd(xa, gamma.to(DEV), beta.to(DEV), y, mode=m, r=r, eps=1e-5)
    P = N * Ho * Wo
    ref, rstats = np.zeros(P * C, npdt), np.zeros(2 * P, np.float32)
    xh, gm, bt = c_ref.host(xin), c_ref.host(gamma), c_ref.host(beta)
    d = L.LnDesc(dc, N, Ho, Wo, C, ldx, C, 0, 0, 0, m, r, 1e-5, 0)
    assert lib.uz_layernorm_fwd_ref(byref(d), c_ref.ptr(xh), c_ref.ptr(gm), c_ref.ptr(bt), None, None, c_ref.ptr(ref), c_ref.ptr(rstats), None) == 0
    agree(y.buf, c_ref.tensor(ref, dt).reshape(P, C), dt, f"layernorm({mode} {r})")
    assert np.allclose(stats.cpu().numpy().reshape(-1), rstats, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("dt", DTS)
def test_attention_gate_forward_kernels_against_the_c_restatement(dt):
    """uz_attn_psi_fwd / uz_attn_gate_fwd (AttentionBlock.forward without its 1x1 convolutions, attention_unet.py:34-40)
    against their restatements: q, the statistics rows of bn_q, the gated output"""
    lib = c_ref.load()
    g = torch.Generator().manual_seed(71)
    dc = L.dtype_code(dt)
    npdt = np.uint16 if dt == torch.bfloat16 else np.float32
    N, H, W, Fi, C = 2, 24, 40, 32, 64
    P = N * H * W
    g1, x1, x = rnd((P, Fi), dt, g), rnd((P, Fi), dt, g), rnd((P, C), dt, g)
    vg, vx = torch.randn(4, Fi, generator=g), torch.randn(4, Fi, generator=g)
    wpsi, bpsi = torch.randn(Fi, generator=g), torch.randn(1, generator=g)
    vq = torch.tensor([[0.7], [-0.2], [0.0], [1.0]])
    ga, xa1, xa = Act(g1.to(DEV), 0, Fi, N, H, W), Act(x1.to(DEV), 0, Fi, N, H, W), Act(x.to(DEV), 0, C, N, H, W)
    q, part = ops.attn_psi_fwd(ga, xa1, vg.to(DEV), vx.to(DEV), wpsi.to(DEV), bpsi.to(DEV))
    out = ops.new_act(N, H, W, C, dt, DEV)
    ops.attn_gate_fwd(xa, q, vq.to(DEV), out)
    qr, pr = np.zeros(P, np.float32), np.zeros(2, np.float32)
    gh, xh1, vgh, vxh, wh, bh = c_ref.host(g1), c_ref.host(x1), c_ref.host(vg), c_ref.host(vx), c_ref.host(wpsi), c_ref.host(bpsi)
    assert lib.uz_attn_psi_fwd_ref(dc, c_ref.ptr(gh), Fi, c_ref.ptr(xh1), Fi, c_ref.ptr(vgh), c_ref.ptr(vxh), c_ref.ptr(wh), c_ref.ptr(bh), P, Fi,
                                   c_ref.ptr(qr), c_ref.ptr(pr), None) == 0
    assert np.allclose(q.cpu().numpy(), qr, rtol=1e-4, atol=1e-4)
    assert np.allclose(part.double().sum(0).reshape(-1).cpu().numpy(), pr, rtol=1e-4)
    orf = np.zeros(P * C, npdt)
    xh, vqh = c_ref.host(x), c_ref.host(vq)
    qk = q.cpu().numpy().copy()      # the kernel's own q, so that the gate is compared on the same input
    assert lib.uz_attn_gate_fwd_ref(dc, c_ref.ptr(xh), C, c_ref.ptr(qk), c_ref.ptr(vqh), P, C, c_ref.ptr(orf), C, None) == 0
    agree(out.buf, c_ref.tensor(orf, dt).reshape(P, C), dt, "attention gate")


@pytest.mark.parametrize("dt", DTS)
def test_batchnorm_relu_pool_kernels_against_the_c_restatement(dt):
    lib = c_ref.load()
    g = torch.Generator().manual_seed(14)
    N, H, W, C = 2, 16, 24, 64
    y = rnd((N * H * W, C), dt, g)
    scale, shift = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    ya = Act(y.to(DEV), 0, C, N, H, W)
    act, pooled = ops.new_act(N, H, W, C, dt, DEV), ops.new_act(N, H // 2, W // 2, C, dt, DEV)
    ops.bn_relu_apply(ya, scale.to(DEV), shift.to(DEV), act, pooled)
    npdt = np.uint16 if dt == torch.bfloat16 else np.float32
    a_r, p_r = np.zeros(N * H * W * C, npdt), np.zeros(N * H * W * C // 4, npdt)
    yh, sc, sh = c_ref.host(y), c_ref.host(scale), c_ref.host(shift)
    assert lib.uz_bn_relu_apply_ref(L.dtype_code(dt), c_ref.ptr(yh), C, c_ref.ptr(sc), c_ref.ptr(sh), N, H, W, C, c_ref.ptr(a_r), C,
                                    c_ref.ptr(p_r), C, None) == 0
    agree(act.buf, c_ref.tensor(a_r, dt).reshape(-1, C), dt, "bn_relu_apply")
    agree(pooled.buf, c_ref.tensor(p_r, dt).reshape(-1, C), dt, "pooled")


@pytest.mark.parametrize("dt", DTS)
def test_attention_building_blocks_against_the_c_restatement(dt):
    lib = c_ref.load()
    g = torch.Generator().manual_seed(15)
    dc = L.dtype_code(dt)
    npdt = np.uint16 if dt == torch.bfloat16 else np.float32
    B, M, N, K = 3, 200, 136, 72
    x, w, res = rnd((B, M, K), dt, g), rnd((B, N, K), dt, g), rnd((B, M, N), dt, g)
    bias = torch.randn(N, generator=g)
    xd, wd, rd = x.to(DEV), w.to(DEV), res.to(DEV)
    y = torch.empty(B, M, N, dtype=dt, device=DEV)
    ops.gemm_nt(dt, B, M, N, K, xd.data_ptr(), K, M * K, wd.data_ptr(), K, N * K, y.data_ptr(), N, M * N, bias=bias.to(DEV),
                res_ptr=rd.data_ptr(), ldres=N, resb=M * N)
    yr = np.zeros(B * M * N, npdt)
    d = L.GemmDesc(dc, B, M, N, K, K, K, N, N, M * K, N * K, M * N, M * N)
    xh, wh, rh, bh = c_ref.host(x), c_ref.host(w), c_ref.host(res), c_ref.host(bias)
    assert lib.uz_gemm_nt_ref(byref(d), c_ref.ptr(xh), c_ref.ptr(wh), c_ref.ptr(bh), c_ref.ptr(rh), c_ref.ptr(yr), None) == 0
    agree(y, c_ref.tensor(yr, dt).reshape(B, M, N), dt, "gemm_nt")
    for axis, (R, C) in ((0, (640, 136)), (1, (77, 256))):
        s = rnd((B, R, C), dt, g, 2.0)
        sd = s.clone().to(DEV)
        ops.softmax_fwd(sd, axis, 0.3)
        sh = c_ref.host(s)
        assert lib.uz_softmax_fwd_ref(dc, c_ref.ptr(sh), C, R * C, B, R, C, axis, 0.3, None, None) == 0
        agree(sd, c_ref.tensor(sh, dt).reshape(B, R, C), dt, f"softmax axis {axis}", f32_tol=1e-5)
        da = rnd((B, R, C), dt, g)
        dd = da.clone().to(DEV)
        ops.softmax_bwd(sd, dd, axis, 0.3)
        ah, dh = c_ref.host(sd), c_ref.host(da)
        dot = np.zeros(B * C, np.float32)
        assert lib.uz_softmax_bwd_ref(dc, c_ref.ptr(ah), c_ref.ptr(dh), C, R * C, B, R, C, axis, 0.3, c_ref.ptr(dot), 0, None) == 0
        agree(dd, c_ref.tensor(dh, dt).reshape(B, R, C), dt, f"softmax bwd axis {axis}", f32_tol=1e-4)
    # UCTransNet's score planes
    Bc, Hh, Cc, KV = 2, 4, 32, 240
    scores = torch.randn(Bc, Hh, Cc, KV, generator=g)
    pc, pct = ops.chanattn_probs_fwd(scores.to(DEV), 1.0 / math.sqrt(KV), 1e-5, dt)
    pr, ptr_ = np.zeros(Bc * Cc * Hh * KV, npdt), np.zeros(Bc * Cc * Hh * KV, npdt)
    sh = c_ref.host(scores)
    assert lib.uz_chanattn_probs_fwd_ref(dc, c_ref.ptr(sh), Bc, Hh, Cc, KV, 1.0 / math.sqrt(KV), 1e-5, c_ref.ptr(pr), c_ref.ptr(ptr_), None) == 0
    agree(pc, c_ref.tensor(pr, dt).reshape(Bc, Cc, Hh * KV), dt, "chanattn probs", f32_tol=1e-4)
    agree(pct, c_ref.tensor(ptr_, dt).reshape(Bc, Hh * KV, Cc), dt, "chanattn probs^T", f32_tol=1e-4)
    dpc = torch.randn(Bc, Cc, Hh * KV, generator=g)
    ds, dst = ops.chanattn_probs_bwd(scores.to(DEV), dpc.to(DEV), 1.0 / math.sqrt(KV), 1e-5, dt)
    dr, dtr = np.zeros(Bc * Cc * Hh * KV, npdt), np.zeros(Bc * Cc * Hh * KV, npdt)
    dh = c_ref.host(dpc)
    assert lib.uz_chanattn_probs_bwd_ref(dc, c_ref.ptr(sh), c_ref.ptr(dh), Bc, Hh, Cc, KV, 1.0 / math.sqrt(KV), 1e-5, c_ref.ptr(dr), c_ref.ptr(dtr), None) == 0
    agree(ds, c_ref.tensor(dr, dt).reshape(Bc, Hh, Cc, KV), dt, "chanattn dS", f32_tol=2e-4)
    agree(dst, c_ref.tensor(dtr, dt).reshape(Bc, Hh, KV, Cc), dt, "chanattn dS^T", f32_tol=2e-4)


@pytest.mark.parametrize("dt", DTS)
def test_head_level_batches_against_the_c_restatement(dt):
    """the second batch level of uz_gemm_nt / uz_wgrad_batched2: UCTransNet's heads side by side in the channels of one
    token map (uctransnet.py:140-168)"""
    lib = c_ref.load()
    g = torch.Generator().manual_seed(21)
    dc = L.dtype_code(dt)
    npdt = np.uint16 if dt == torch.bfloat16 else np.float32
    B, n, H, C, KV = 3, 64, 4, 16, 40
    Q, K = rnd((B * n, H * C), dt, g), rnd((B * n, H * KV), dt, g)
    scores = ops.wgrad_heads(Act(Q.to(DEV), 0, H * C, B, 8, 8), Act(K.to(DEV), 0, H * KV, B, 8, 8), H)
    ref = np.zeros(B * H * C * KV, np.float32)
    d = L.WgradDesc(dc, 1, 1, n, 1, n, C, H * C, KV, H * KV, 1, L.TAPS_CONV, 1)
    Qh, Kh = c_ref.host(Q), c_ref.host(K)
    assert lib.uz_wgrad_batched2_ref(byref(d), B, H, c_ref.ptr(Qh), n * H * C, C, c_ref.ptr(Kh), n * H * KV, KV, c_ref.ptr(ref), C * KV,
                                     None, None) == 0
    r = torch.from_numpy(ref).reshape(B, H, C, KV).double()
    assert ((scores.cpu().double() - r).abs().max() / r.abs().max()).item() < 1e-5
    # dQ_h = K_h dS_h^T: x = K (head h at column h * KV), w = dS (B, H, C, KV), y = dQ (head h at column h * C)
    ds = rnd((B, H, C, KV), dt, g)
    dQ = torch.zeros(B * n, H * C, dtype=dt, device=DEV)
    Kd, dsd = K.to(DEV), ds.to(DEV)
    ops.gemm_nt(dt, B, n, C, KV, Kd.data_ptr(), H * KV, n * H * KV, dsd.data_ptr(), KV, H * C * KV, dQ.data_ptr(), H * C, n * H * C,
                batch2=H, xb2=KV, wb2=C * KV, yb2=C)
    yr = np.zeros(B * n * H * C, npdt)
    gd = L.GemmDesc(dc, B, n, C, KV, H * KV, KV, H * C, 0, n * H * KV, H * C * KV, n * H * C, 0, H, KV, C * KV, C, 0)
    dsh = c_ref.host(ds)
    assert lib.uz_gemm_nt_ref(byref(gd), c_ref.ptr(Kh), c_ref.ptr(dsh), None, None, c_ref.ptr(yr), None) == 0
    agree(dQ, c_ref.tensor(yr, dt).reshape(B * n, H * C), dt, "gemm_nt heads")


@pytest.mark.parametrize("dt", DTS)
def test_element_kernels_of_the_transformer_families_against_the_c_restatement(dt):
    """GELU, relu(a + b), nearest-upsampling gradient, bilinear resize (both corner conventions), depthwise 3x3, space to
    depth, column sums, LayerNorm with residual / per-image scale / GELU"""
    lib = c_ref.load()
    g = torch.Generator().manual_seed(31)
    dc = L.dtype_code(dt)
    npdt = np.uint16 if dt == torch.bfloat16 else np.float32
    N, H, W, C = 2, 12, 20, 64
    P = N * H * W

    def act(t, h=H, w=W, c=C):
        return Act(t.to(DEV), 0, c, N, h, w)

    def new(h=H, w=W, c=C):
        return ops.new_act(N, h, w, c, dt, DEV)

    x, gy, b = rnd((P, C), dt, g), rnd((P, C), dt, g), rnd((P, C), dt, g)
    xh, gh, bh = c_ref.host(x), c_ref.host(gy), c_ref.host(b)
    ref = np.zeros(P * C, npdt)
    # GELU and its gradient
    y = new()
    ops.gelu_fwd(act(x), y)
    assert lib.uz_gelu_fwd_ref(dc, c_ref.ptr(xh), C, c_ref.ptr(ref), C, P, C, None) == 0
    agree(y.buf, c_ref.tensor(ref, dt).reshape(P, C), dt, "gelu")
    ops.gelu_bwd(act(x), act(gy), y)
    assert lib.uz_gelu_bwd_ref(dc, c_ref.ptr(xh), C, c_ref.ptr(gh), C, c_ref.ptr(ref), C, P, C, None) == 0
    agree(y.buf, c_ref.tensor(ref, dt).reshape(P, C), dt, "gelu bwd")
    # relu(a + b)
    ops.add_relu(act(x), act(b), y)
    assert lib.uz_add_relu_ref(dc, c_ref.ptr(xh), C, c_ref.ptr(bh), C, c_ref.ptr(ref), C, P, C, None) == 0
    agree(y.buf, c_ref.tensor(ref, dt).reshape(P, C), dt, "add_relu")
    # nearest x2 upsampling's gradient
    du = rnd((N * 4 * H * W, C), dt, g)
    dx = new()
    ops.sum2x2(act(du, 2 * H, 2 * W), dx)
    duh = c_ref.host(du)
    assert lib.uz_sum2x2_ref(dc, c_ref.ptr(duh), C, N, H, W, C, c_ref.ptr(ref), C, None) == 0
    agree(dx.buf, c_ref.tensor(ref, dt).reshape(P, C), dt, "sum2x2")
    # bilinear resize
    for (Ho, Wo, ac) in ((24, 40, True), (17, 23, False), (6, 10, False)):
        o = new(Ho, Wo)
        ops.bilinear_fwd(act(x), o, align_corners=ac)
        r2 = np.zeros(N * Ho * Wo * C, npdt)
        assert lib.uz_resize_bilinear_fwd_ref(dc, c_ref.ptr(xh), C, H * W * C, N, H, W, C, c_ref.ptr(r2), C, Ho * Wo * C, Ho, Wo, int(ac), None) == 0
        agree(o.buf, c_ref.tensor(r2, dt).reshape(-1, C), dt, f"bilinear {Ho}x{Wo} {ac}")
    # depthwise 3x3 with the skip
    taps, bias = torch.randn(9, C, generator=g) * 0.3, torch.randn(C, generator=g)
    ops.dwconv3x3(act(x), taps.to(DEV), bias.to(DEV), y, skip=True)
    th, bsh = c_ref.host(taps), c_ref.host(bias)
    assert lib.uz_dwconv3x3_ref(dc, c_ref.ptr(xh), C, c_ref.ptr(th), c_ref.ptr(bsh), c_ref.ptr(ref), C, N, H, W, C, 1, None) == 0
    agree(y.buf, c_ref.tensor(ref, dt).reshape(P, C), dt, "dwconv3x3")
    # space to depth
    cols = new(H // 2, W // 2, 4 * C)
    ops.space_to_depth(act(x), cols, 2)
    rc = np.zeros(P * C, npdt)
    assert lib.uz_space_to_depth_ref(dc, c_ref.ptr(xh), C, c_ref.ptr(rc), 4 * C, N, H // 2, W // 2, C, 2, 0, None) == 0
    assert torch.equal(cols.buf.cpu(), c_ref.tensor(rc, dt).reshape(-1, 4 * C))
    # column sums
    cs = ops.colsum(act(x))
    rs = np.zeros(C, np.float32)
    assert lib.uz_colsum_ref(dc, c_ref.ptr(xh), C, P, C, c_ref.ptr(rs), None) == 0
    assert np.allclose(cs.cpu().numpy(), rs, rtol=1e-5, atol=1e-3)
    # LayerNorm + residual + per-image scale, and GELU(LayerNorm)
    gamma, beta, isc = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g), torch.rand(N, generator=g) + 0.5
    gm, bt, ih = c_ref.host(gamma), c_ref.host(beta), c_ref.host(isc)
    stats = ops.layernorm_fwd(act(x), gamma.to(DEV), beta.to(DEV), y, eps=1e-5, res=act(b), image_scale=isc.to(DEV))
    d = L.LnDesc(dc, N, H, W, C, C, C, C, 0, 0, 0, 1, 1e-5, 0)
    rstats = np.zeros(2 * P, np.float32)
    assert lib.uz_layernorm_fwd_ref(byref(d), c_ref.ptr(xh), c_ref.ptr(gm), c_ref.ptr(bt), c_ref.ptr(bh), c_ref.ptr(ih), c_ref.ptr(ref), c_ref.ptr(rstats), None) == 0
    agree(y.buf, c_ref.tensor(ref, dt).reshape(P, C), dt, "layernorm")
    assert np.allclose(stats.cpu().numpy().reshape(-1), rstats, rtol=1e-4, atol=1e-5)
    ops.layernorm_fwd(act(x), gamma.to(DEV), beta.to(DEV), y, eps=1e-5, gelu=True)
    d2 = L.LnDesc(dc, N, H, W, C, C, C, 0, 0, 0, 0, 1, 1e-5, 1)
    assert lib.uz_layernorm_fwd_ref(byref(d2), c_ref.ptr(xh), c_ref.ptr(gm), c_ref.ptr(bt), None, None, c_ref.ptr(ref), c_ref.ptr(rstats), None) == 0
    agree(y.buf, c_ref.tensor(ref, dt).reshape(P, C), dt, "gelu(layernorm)")
