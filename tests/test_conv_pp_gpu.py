"""GPU: the third-generation ("ping-pong") direct 3x3 convolution (csrc/uz_conv3x3_pp.hip) through the C ABI against
F.conv2d on the same bf16-rounded operands (reference: nn.Conv2d k3 p1, unet_zoo/models/common_layers.py:28,31,47,52,71
and its input gradient).  The library's own plan decides which kernel a descriptor gets; every case below asserts
through uz_conv_igemm_kernel_name() that it IS the ping-pong kernel, so the test cannot pass on another generation.
Covered: bias + BatchNorm partial sums of the stored values, ragged 16 x 32 tiles in both directions, a channel tail
inside the 128-wide tile, 1 ... 8 channel slabs of 32, several tiles per workgroup (the cross-tile prefetch and its
counted waits) and a single tile per workgroup (the drain path), channel windows of wider NaN-poisoned buffers, the
nearest-upsampled input of UpConvBlock (common_layers.py:69-72), the fused BatchNorm-backward reduction, bitwise
repeatability."""
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


def kernel_of(N, H, W, Cin, ldx, Cout, ldy, up=False):
    d = L.ConvDesc(L.dtype_code(dt), N, H, W, H // 2 if up else H, W // 2 if up else W, Cin, ldx, Cout, ldy, 9,
                   L.TAPS_CONV_UP2 if up else L.TAPS_CONV, 1, L.STORE_PLAIN, 0, 0, 0)
    return ops.conv_kernel_name(d)


CASES = [
    # configuration, N, H, W, Cin, Cout, xwin, ups
    ("pp512", 3, 128, 256, 64, 128, 0, False),     # 192 tiles on 256 CUs: one tile per workgroup (drain path), two slabs
    ("pp512", 7, 128, 256, 32, 128, 0, False),     # 448 tiles: two tiles for some workgroups, ONE slab per tile (first == last)
    ("pp512", 2, 120, 250, 96, 200, 0, False),     # ragged rows and columns, three slabs, channel tail in the second N tile
    ("pp512", 6, 112, 224, 128, 136, 32, False),   # input = channel window of a wider NaN-poisoned buffer; 8-channel tail tile
    ("pp512", 3, 128, 256, 64, 128, 0, True),      # nearest x2 upsampled input
    ("pp512", 4, 64, 64, 256, 1024, 0, False),     # eight slabs, eight N tiles (the unet 512 -> 1024 input-gradient shape class)
    ("pp512x64", 3, 128, 256, 128, 64, 0, False),  # 64 output channels: eight waves along the pixels, phases of one tap row
    ("pp512x64", 2, 120, 250, 64, 40, 32, False),  # ragged, channel tail, window; one tile per workgroup
    ("pp512x64", 9, 128, 256, 32, 64, 0, False),   # 576 tiles: up to three per workgroup, one slab per tile
    ("pp256", 6, 40, 72, 128, 256, 0, False),      # 8 x 32 patches, ragged both ways, two N tiles
    ("pp128w16", 2, 40, 72, 128, 256, 0, False),   # the same maps, 60 tiles: 8 x 16 patches (ragged both ways) so that 120 CUs work, not 60
    ("pp128w16", 1, 32, 32, 64, 128, 0, False),    # eight half tiles
    ("pp256", 16, 32, 32, 256, 512, 0, False),     # unet level 4 at batch 16: one tile per workgroup, eight slabs
    ("pp128w16", 2, 48, 64, 32, 128, 0, True),     # upsampled input, one slab
    ("pp256", 12, 48, 64, 32, 128, 0, True),       # upsampled input, one slab, 288 tiles
    ("pp256", 40, 32, 32, 96, 512, 64, False),     # 160 tiles x 4 N tiles: the grid is capped at 64 workgroups per N tile, 2-3 tiles each
    ("pp128w16", 4, 16, 16, 256, 384, 0, False),   # half a 16 x 16 map per tile
    ("pp128w16", 3, 12, 14, 128, 136, 0, False),   # ragged 8 x 16 patches, channel tail
    ("pp256w16", 70, 12, 14, 128, 136, 0, False),  # ragged 16 x 16 patches, channel tail (140 tiles)
    ("pp256w16", 70, 16, 16, 64, 512, 0, False),   # 70 x 4 tiles: the grid is capped at 64 workgroups per N tile
    # split-K (few tiles, many channel slabs: the slabs of a tile are dealt to several workgroups, fp32 partial tiles + a
    # fixed-order reduce pass with bias and statistics)
    ("pp256w16", 16, 16, 16, 1024, 512, 0, False),  # unet's 1024 -> 512 at 16 x 16, B = 16: 64 tiles, 32 slabs -> 4 ranges of 8
    ("pp128w16", 16, 16, 16, 512, 1024, 0, False),  # 128 whole-map tiles: more than a quarter of the CUs -> not split, 256 half tiles
    ("pp256", 2, 32, 32, 512, 256, 32, False),      # 16 tiles -> 4 ranges of 4 slabs; input window of a NaN-poisoned buffer
    ("pp256w16", 2, 12, 14, 544, 136, 0, False),    # ragged patches, channel tail, 17 slabs -> ranges of 5 + 5 + 5 + 2
    ("pp256", 1, 32, 64, 1024, 128, 0, True),       # upsampled input, 32 slabs -> 8 ranges
]
SPLITK = {(16, 16, 16, 1024, 512): 4, (2, 32, 32, 512, 256): 4, (2, 12, 14, 544, 136): 4, (1, 32, 64, 1024, 128): 8}


@pytest.mark.parametrize("cfg,N,H,W,Cin,Cout,win,ups", CASES)
def test_pp_conv3x3_fwd_bias_stats(cfg, N, H, W, Cin, Cout, win, ups):
    g = torch.Generator().manual_seed(21)
    Hi, Wi = (H // 2, W // 2) if ups else (H, W)
    x = rnd(torch.randn(N, Cin, Hi, Wi, generator=g))
    w = rnd(torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05)
    b = torch.randn(Cout, generator=g)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if ups else x
    ref = F.conv2d(xin, w, b, padding=1)
    xa = act_from_nchw(x.to(DEV), dt)
    if win:
        wide = torch.full((xa.P, Cin + 2 * win), float("nan"), dtype=dt, device=DEV)
        wide[:, win:win + Cin] = xa.buf
        xa = Act(wide, win, Cin, N, Hi, Wi)
    wp = ops.pack_weights(w.to(DEV), L.PACK_CONV_FWD, dt)
    # the output is a channel window of a wider buffer too: the neighbours must stay untouched
    ywide = torch.full((N * H * W, Cout + 16), 7.0, dtype=dt, device=DEV)
    y = Act(ywide, 8, Cout, N, H, W)
    assert kernel_of(N, H, W, Cin, xa.ld, Cout, y.ld, ups) == f"conv3x3_{cfg}_bf16" + ("_up2" if ups else "")
    d = L.ConvDesc(L.dtype_code(dt), N, H, W, Hi, Wi, Cin, xa.ld, Cout, y.ld, 9, L.TAPS_CONV_UP2 if ups else L.TAPS_CONV, 1,
                   L.STORE_PLAIN, 0, 0, 0)
    split = SPLITK.get((N, H, W, Cin, Cout), 1)
    assert ops.conv_kernel_name(d, with_workspace=True).endswith("_splitk") == (split > 1)
    assert L.load().uz_conv_igemm_workspace_bytes(byref(d)) == (split * N * H * W * Cout * 4 if split > 1 else 0)
    stats = ops.conv_igemm(xa, wp, b.to(DEV), y, ntaps=9, want_stats=True,
                           taps_mode=L.TAPS_CONV_UP2 if ups else L.TAPS_CONV)
    got = y.dense().cpu()
    assert torch.isfinite(got).all()
    assert relerr(got, ref) < 2e-2
    assert (ywide[:, :8] == 7.0).all() and (ywide[:, 8 + Cout:] == 7.0).all()
    # statistics are those of the STORED (rounded) values
    s = stats.double().sum(0).cpu()
    assert relerr(s[0], got.double().sum((0, 2, 3))) < 3e-3
    assert relerr(s[1], (got.double() ** 2).sum((0, 2, 3))) < 1e-4
    # bitwise repeatable (fixed reduction order, no atomics)
    ywide2 = torch.full((N * H * W, Cout + 16), 7.0, dtype=dt, device=DEV)
    y2 = Act(ywide2, 8, Cout, N, H, W)
    stats2 = ops.conv_igemm(xa, wp, b.to(DEV), y2, ntaps=9, want_stats=True,
                            taps_mode=L.TAPS_CONV_UP2 if ups else L.TAPS_CONV)
    assert torch.equal(ywide, ywide2) and torch.equal(stats, stats2)
    if split > 1:
        # the same descriptor WITHOUT a workspace is the unsplit launch: same sum in another order -- at most one bf16
        # rounding apart, nearly everywhere the identical number
        y3 = ops.new_act(N, H, W, Cout, dt, DEV)
        d3 = L.ConvDesc(L.dtype_code(dt), N, H, W, Hi, Wi, Cin, xa.ld, Cout, y3.ld, 9, L.TAPS_CONV_UP2 if ups else L.TAPS_CONV, 1,
                        L.STORE_PLAIN, 0, 0, 0)
        L.check(L.load().uz_conv_igemm(byref(d3), xa.ptr(), wp.data_ptr(), b.to(DEV).data_ptr(), y3.ptr(), None, L.stream_ptr()),
                "uz_conv_igemm")
        a3, a1 = y3.dense().cpu(), got
        assert relerr(a1, a3) < 2.0 ** -7 and (a1 == a3).float().mean() > 0.97


def test_pp_split_k_exact_on_small_integers():
    """the split-K form on integer-valued operands (exact in fp32 whatever the order): equal to the fp32 reference"""
    g = torch.Generator().manual_seed(24)
    N, H, W, Cin, Cout = 16, 16, 16, 1024, 512
    x = torch.randint(-2, 3, (N, Cin, H, W), generator=g).float()
    w = (torch.randint(-1, 2, (Cout, Cin, 3, 3), generator=g) * (torch.rand(Cout, Cin, 3, 3, generator=g) < 0.02)).float()
    b = torch.randint(-3, 4, (Cout,), generator=g).float()
    ref = F.conv2d(x, w, b, padding=1)
    assert ref.abs().max() < 256
    xa = act_from_nchw(x.to(DEV), dt)
    wp = ops.pack_weights(w.to(DEV), L.PACK_CONV_FWD, dt)
    y = ops.new_act(N, H, W, Cout, dt, DEV)
    d = L.ConvDesc(L.dtype_code(dt), N, H, W, H, W, Cin, xa.ld, Cout, y.ld, 9, L.TAPS_CONV, 1, L.STORE_PLAIN, 0, 0, 0)
    assert ops.conv_kernel_name(d, with_workspace=True) == "conv3x3_pp256w16_bf16_splitk"
    stats = ops.conv_igemm(xa, wp, b.to(DEV), y, ntaps=9, want_stats=True)
    assert torch.equal(y.dense().cpu(), ref)
    assert torch.equal(stats.double().sum(0)[0].cpu(), ref.double().sum((0, 2, 3)))


def test_pp_conv3x3_exact_on_small_integers():
    """integer-valued operands whose products and sums are exact in bf16 x bf16 -> fp32: every output must equal the
    fp32 reference EXACTLY (a wrong tap, a wrong swizzle or a missed halo row cannot hide inside a tolerance)"""
    g = torch.Generator().manual_seed(22)
    N, H, W, Cin, Cout = 3, 128, 256, 64, 128
    x = torch.randint(-2, 3, (N, Cin, H, W), generator=g).float()
    w = torch.randint(-1, 2, (Cout, Cin, 3, 3), generator=g).float()
    ref = F.conv2d(x, w, None, padding=1)
    assert ref.abs().max() < 256     # exactly representable in bf16
    xa = act_from_nchw(x.to(DEV), dt)
    wp = ops.pack_weights(w.to(DEV), L.PACK_CONV_FWD, dt)
    y = ops.new_act(N, H, W, Cout, dt, DEV)
    assert kernel_of(N, H, W, Cin, xa.ld, Cout, y.ld).startswith("conv3x3_pp")
    ops.conv_igemm(xa, wp, None, y, ntaps=9)
    assert torch.equal(y.dense().cpu(), ref)


@pytest.mark.parametrize("N,H,W,C,Cn", [
    (3, 128, 256, 64, 128),     # pp512, one tile per workgroup
    (6, 112, 224, 128, 136),    # pp512, two tiles for some, ragged, channel tail
    (3, 128, 256, 64, 64),      # pp512x64
    (36, 32, 32, 128, 128),     # pp256
    (70, 16, 16, 64, 256),      # pp256w16
    (16, 32, 32, 128, 128),     # pp128w16 on 32-wide maps
    (5, 16, 16, 256, 256),      # pp128w16 on 16-wide maps
])
def test_pp_bn_backward_reduction_in_the_input_gradient_epilogue(N, H, W, C, Cn):
    """uz_conv_igemm_bnred on the ping-pong kernel: the gradient must be bit-identical to the plain launch, the
    finalized BatchNorm-backward sums must agree with the stand-alone reduction pass (uz_bn_relu_bwd_reduce)"""
    gen = torch.Generator().manual_seed(23)
    dyb = act_from_nchw(rnd(torch.randn(N, C, H, W, generator=gen)).to(DEV), dt)
    wb = torch.randn(Cn, C, 3, 3, generator=gen) * 0.05
    wp = ops.pack_weights(wb.to(DEV), L.PACK_CONV_FWD, dt)
    y = act_from_nchw(rnd(torch.randn(N, Cn, H, W, generator=gen) * 2 + 0.3).to(DEV), dt)
    gamma = (torch.rand(Cn, generator=gen) + 0.5).to(DEV)
    beta = (torch.randn(Cn, generator=gen) * 0.2).to(DEV)
    yd = y.dense().double()
    stats = torch.stack([yd.sum((0, 2, 3)), (yd ** 2).sum((0, 2, 3))]).float().reshape(1, 2, Cn)
    vec = ops.bn_finalize(stats, N * H * W, gamma, beta, 1e-5, 0.1, torch.zeros(Cn, device=DEV), torch.ones(Cn, device=DEV))
    g_plain = ops.new_act(N, H, W, Cn, dt, DEV)
    assert kernel_of(N, H, W, C, dyb.ld, Cn, g_plain.ld).startswith("conv3x3_pp")
    # the plain launch WITHOUT a workspace: the unsplit kernel, whose summation order the fused epilogue shares (with a
    # workspace the 16 x 16 case takes the split-K form: same sums, another order)
    dpl = L.ConvDesc(L.dtype_code(dt), N, H, W, H, W, C, dyb.ld, Cn, g_plain.ld, 9, L.TAPS_CONV, 1, L.STORE_PLAIN, 0, 0, 0)
    L.check(L.load().uz_conv_igemm(byref(dpl), dyb.ptr(), wp.data_ptr(), None, g_plain.ptr(), None, L.stream_ptr()), "uz_conv_igemm")
    g_fused = ops.new_act(N, H, W, Cn, dt, DEV)
    part = ops.conv_igemm(dyb, wp, None, g_fused, ntaps=9, bnred=(y, vec))
    assert part is not None and part.shape[1:] == (2, Cn)
    assert torch.equal(g_plain.dense(), g_fused.dense())
    out = []
    for partials in (None, part):
        sums = torch.zeros(2, Cn, dtype=torch.float64, device=DEV)
        dx = ops.new_act(N, H, W, Cn, dt, DEV)
        dgb = torch.empty(2, Cn, device=DEV)
        ops.bn_relu_bwd(y, vec, g_fused, None, None, sums, dx, dgb[0], dgb[1], partials=partials)
        out.append((sums.clone(), dx.dense().float(), dgb.clone()))
    assert relerr(out[1][0], out[0][0]) < 2e-5 and relerr(out[1][2], out[0][2]) < 2e-5
    assert relerr(out[1][1], out[0][1]) < 1e-2


def test_pp_plan_picks_a_configuration_per_problem():
    """uz_pp_plan(): 512-pixel tiles where they fill the chip, 256-pixel tiles on the small maps, 16 x 16 patches on
    16-wide maps, the 64-channel configuration for 64 output channels; everything else stays on the other kernels"""
    assert kernel_of(16, 128, 128, 128, 128, 128, 128) == "conv3x3_pp512_bf16"        # unet level 2 at batch 16
    assert kernel_of(16, 32, 32, 512, 512, 512, 512) == "conv3x3_pp256_bf16"          # level 4: 128 tiles of 512 pixels would idle half the CUs
    assert kernel_of(16, 16, 16, 1024, 1024, 1024, 1024) == "conv3x3_pp128w16_bf16"   # level 5: 128 whole maps would idle half the CUs
    assert kernel_of(32, 16, 16, 1024, 1024, 1024, 1024) == "conv3x3_pp256w16_bf16"   # twice the batch: one map per tile
    assert kernel_of(16, 32, 32, 512, 512, 256, 256) == "conv3x3_pp128w16_bf16"       # the input gradient of level 4's first convolution
    assert kernel_of(16, 256, 256, 128, 128, 64, 64) == "conv3x3_pp512x64_bf16"       # decoder level 1
    assert not kernel_of(1, 32, 32, 64, 64, 64, 64).startswith("conv3x3_pp")          # 64 channels, two tiles
    assert not kernel_of(2, 64, 64, 16, 16, 128, 128).startswith("conv3x3_pp")        # 16 input channels (u2net)
    d = L.ConvDesc(L.dtype_code(torch.float32), 16, 128, 128, 128, 128, 128, 128, 128, 128, 9, L.TAPS_CONV, 1,
                   L.STORE_PLAIN, 0, 0, 0)
    assert not ops.conv_kernel_name(d).startswith("conv3x3_pp")                       # the fp32 parity mode
