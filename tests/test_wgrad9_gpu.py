"""GPU: the row-walk nine-tap weight-gradient kernel (unet_zoo_amd/csrc/uz_wgrad9.hip) through the C ABI, asserted BY NAME
through uz_wgrad_kernel_name(), against autograd's weight gradient of F.conv2d (what loss.backward() computes for
unet_zoo/models/common_layers.py:28,31) on bf16-rounded operands: 2e-2 of the tensor's max on random data, EXACT on small
integers (every product and every partial sum is then representable, so any wrong / missing / doubled pixel shows).
Covers each strip width (W = 16, 32, 64, several strips per row), several channel tiles, channel tails, operands that are
windows of wider NaN-poisoned buffers, the nearest-x2-upsampled x operand, segment starts inside a workgroup's range,
image borders and bitwise repeatability."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from unet_zoo_amd import _lib as L
from unet_zoo_amd import ops
from unet_zoo_amd.ops import Act, act_from_nchw

DEV = "cuda"
dt = torch.bfloat16


def relerr(a, b):
    return ((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30)).item()


def ref_wgrad(x, dy, up=False):
    w = torch.zeros(dy.shape[1], x.shape[1], 3, 3, requires_grad=True)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    F.conv2d(xin, w, None, padding=1).backward(dy)
    return w.grad


def kname(dya, xa, mode=L.TAPS_CONV):
    d = L.WgradDesc(L.dtype_code(dt), dya.N, dya.H, dya.W, xa.H, xa.W, dya.C, dya.ld, xa.C, xa.ld, 9, mode, 1)
    return ops.wgrad_kernel_name(d)


SHAPES = [
    # N, H, W, Cin, Cout
    (2, 16, 64, 64, 64),      # 64 x 64 tile, one strip, one row per step
    (1, 8, 128, 128, 128),    # 128 x 64 tile, two strips per row: real halo columns between the strips
    (1, 6, 256, 32, 64),      # four strips; x channels < 64 (tail of the x tile)
    (2, 16, 32, 128, 128),    # W = 32: two rows per step, two x tiles
    (2, 16, 16, 256, 128),    # W = 16: four rows per step, four images' worth of segment starts
    (3, 16, 16, 64, 320),     # three dy tiles of 128 (the last one a 64-channel tail), odd image count
    (1, 64, 64, 72, 40),      # channel tails on both operands (multiples of 8 only)
    (5, 8, 64, 64, 64),       # segments (8 steps) shorter than a workgroup's range: starts in mid-range
    (1, 128, 64, 192, 64),    # x tiles 3 x dy tile 1, long walk
]


@pytest.mark.parametrize("N,H,W,Cin,Cout", SHAPES)
def test_rowwalk_against_autograd(N, H, W, Cin, Cout):
    g = torch.Generator().manual_seed(H * 1000 + W + Cin)
    x = torch.randn(N, Cin, H, W, generator=g).to(dt).float()
    dy = torch.randn(N, Cout, H, W, generator=g).to(dt).float()
    ref = ref_wgrad(x, dy)
    dya, xa = act_from_nchw(dy.to(DEV), dt), act_from_nchw(x.to(DEV), dt)
    assert kname(dya, xa) == "wgrad9_bf16_64x64_rowwalk"
    got = ops.wgrad(dya, xa, (Cout, Cin, 3, 3), ntaps=9)
    assert relerr(got.cpu(), ref) < 2e-2
    again = ops.wgrad(dya, xa, (Cout, Cin, 3, 3), ntaps=9)
    assert torch.equal(got, again), "two launches on the same bytes differ"


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 16, 64, 64, 64), (1, 8, 128, 128, 128), (2, 16, 32, 128, 128),
                                            (2, 16, 16, 128, 128), (1, 32, 128, 40, 72)])
def test_rowwalk_exact_on_integers(N, H, W, Cin, Cout):
    """operands in {-2..2}: |sum| < 2^24, so fp32 accumulation in any order is exact and the result must EQUAL autograd's"""
    g = torch.Generator().manual_seed(7)
    x = torch.randint(-2, 3, (N, Cin, H, W), generator=g).float()
    dy = torch.randint(-2, 3, (N, Cout, H, W), generator=g).float()
    ref = ref_wgrad(x, dy)
    dya, xa = act_from_nchw(dy.to(DEV), dt), act_from_nchw(x.to(DEV), dt)
    assert kname(dya, xa).startswith("wgrad9_")
    got = ops.wgrad(dya, xa, (Cout, Cin, 3, 3), ntaps=9).cpu()
    bad = (got != ref).nonzero()
    assert bad.numel() == 0, f"{bad.shape[0]} elements differ, first (co, ci, ty, tx) = {bad[0].tolist()}: {got[tuple(bad[0])]} vs {ref[tuple(bad[0])]}"


def test_rowwalk_border_taps_see_zero_padding():
    """an all-ones problem counts the pixels every tap sees: H*W for the centre, (H-1)*W / H*(W-1) / (H-1)*(W-1) for the
    edge and corner taps -- the zero padding of rows -1, H and columns -1, W, per image and per strip"""
    N, H, W, C = 3, 16, 128, 64
    x = torch.ones(N, C, H, W)
    dy = torch.ones(N, C, H, W)
    got = ops.wgrad(act_from_nchw(dy.to(DEV), dt), act_from_nchw(x.to(DEV), dt), (C, C, 3, 3), ntaps=9).cpu()
    rows = torch.tensor([H - 1, H, H - 1]).float()
    cols = torch.tensor([W - 1, W, W - 1]).float()
    want = (N * rows[:, None] * cols[None, :]).expand(C, C, 3, 3)
    assert torch.equal(got, want)


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 16, 64, 64, 64), (1, 16, 32, 96, 192)])
def test_rowwalk_channel_windows_of_poisoned_buffers(N, H, W, Cin, Cout):
    """both operands are channel windows of wider buffers whose other channels are NaN (concat slots)"""
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, Cin, H, W, generator=g).to(dt).float()
    dy = torch.randn(N, Cout, H, W, generator=g).to(dt).float()
    ref = ref_wgrad(x, dy)
    P = N * H * W
    xw = torch.full((P, Cin + 96), float("nan"), dtype=dt, device=DEV)
    xw[:, 64:64 + Cin] = act_from_nchw(x.to(DEV), dt).buf
    dw_ = torch.full((P, Cout + 40), float("nan"), dtype=dt, device=DEV)
    dw_[:, 8:8 + Cout] = act_from_nchw(dy.to(DEV), dt).buf
    dya, xa = Act(dw_, 8, Cout, N, H, W), Act(xw, 64, Cin, N, H, W)
    assert kname(dya, xa).startswith("wgrad9_")
    got = ops.wgrad(dya, xa, (Cout, Cin, 3, 3), ntaps=9)
    assert torch.isfinite(got).all()
    assert relerr(got.cpu(), ref) < 2e-2


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 16, 64, 64, 64), (1, 32, 32, 128, 128), (1, 16, 128, 64, 192)])
def test_rowwalk_upsampled_x(N, H, W, Cin, Cout):
    """UpConvBlock (attention_unet.py:16-29): x lives at (H/2, W/2) and is read through nearest x2 upsampling"""
    g = torch.Generator().manual_seed(13)
    x = torch.randint(-2, 3, (N, Cin, H // 2, W // 2), generator=g).float()
    dy = torch.randint(-2, 3, (N, Cout, H, W), generator=g).float()
    ref = ref_wgrad(x, dy, up=True)
    dya, xa = act_from_nchw(dy.to(DEV), dt), act_from_nchw(x.to(DEV), dt)
    assert kname(dya, xa, L.TAPS_CONV_UP2).startswith("wgrad9_")
    got = ops.wgrad(dya, xa, (Cout, Cin, 3, 3), ntaps=9, taps_mode=L.TAPS_CONV_UP2).cpu()
    assert torch.equal(got, ref)


def test_rowwalk_takes_the_layers_the_first_form_left_to_round3():
    """dy twice as wide as x on a small map; 64 dy channels against 128 x channels (the 128 x 64 tile on eight waves lost
    there; 64 x 64 tiles with loader waves do not)"""
    for (N, H, W, Cin, Cout) in [(1, 8, 128, 64, 128), (1, 16, 64, 128, 64)]:
        g = torch.Generator().manual_seed(5)
        x = torch.randint(-2, 3, (N, Cin, H, W), generator=g).float()
        dy = torch.randint(-2, 3, (N, Cout, H, W), generator=g).float()
        dya, xa = act_from_nchw(dy.to(DEV), dt), act_from_nchw(x.to(DEV), dt)
        assert kname(dya, xa).startswith("wgrad9_bf16_")
        assert torch.equal(ops.wgrad(dya, xa, (Cout, Cin, 3, 3), ntaps=9).cpu(), ref_wgrad(x, dy))


def test_rowwalk_full_size_layer_properties():
    """unet's 64 -> 64 layer at BASELINE configs[1] size (B = 16, 256 x 256): linearity in dy (a size-independent property:
    dW(dy1 + dy2) = dW(dy1) + dW(dy2), exact on integers) and agreement with autograd on a corner block of channels"""
    N, H, W, C = 16, 256, 256, 64
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randint(-1, 2, (N * H * W, C), generator=g, device=DEV).to(dt)
    d1 = torch.randint(-1, 2, (N * H * W, C), generator=g, device=DEV).to(dt)
    d2 = torch.randint(-1, 2, (N * H * W, C), generator=g, device=DEV).to(dt)
    xa = Act(x, 0, C, N, H, W)
    w1 = ops.wgrad(Act(d1, 0, C, N, H, W), xa, (C, C, 3, 3), ntaps=9)
    w2 = ops.wgrad(Act(d2, 0, C, N, H, W), xa, (C, C, 3, 3), ntaps=9)
    w12 = ops.wgrad(Act((d1.float() + d2.float()).to(dt), 0, C, N, H, W), xa, (C, C, 3, 3), ntaps=9)
    assert torch.equal(w12, w1 + w2)
    # 8 x 8 channels of the first two images against torch on the GPU (fp32 conv2d of integers is exact)
    xs = x.view(N, H, W, C)[:2, :, :, :8].permute(0, 3, 1, 2).float()
    ds = d1.view(N, H, W, C)[:2, :, :, :8].permute(0, 3, 1, 2).float()
    sub = ops.wgrad(act_from_nchw(ds, dt), act_from_nchw(xs, dt), (8, 8, 3, 3), ntaps=9)
    wz = torch.zeros(8, 8, 3, 3, device=DEV, requires_grad=True)
    F.conv2d(xs, wz, None, padding=1).backward(ds)
    assert torch.equal(sub, wz.grad)


# ---- 2 x 2 gather (ConvTranspose2d k2 s2) weight gradient, four taps per workgroup: unet_zoo_amd/csrc/uz_wgrad_g4.hip ----
def ref_convt_wgrad(x, dy):
    """autograd's weight gradient of F.conv_transpose2d(x, w, stride=2) (common_layers.py:104), cropped / zero-padded to
    dy's size as UpSample_UNet pads odd skips (common_layers.py:110-113)"""
    w = torch.zeros(x.shape[1], dy.shape[1], 2, 2, requires_grad=True)
    y = F.conv_transpose2d(x, w, None, stride=2)
    y = F.pad(y, [0, dy.shape[3] - y.shape[3], 0, dy.shape[2] - y.shape[2]])
    y.backward(dy)
    return w.grad


@pytest.mark.parametrize("N,H,W,Cin,Cout,odd", [
    (2, 16, 16, 256, 128, 0),    # W = 16: four coarse rows per step; 2 x 2 tiles
    (1, 8, 32, 128, 64, 0),      # W = 32
    (2, 4, 64, 64, 96, 0),       # 64-wide x tile (four waves), g tail tile of 32 channels
    (1, 6, 128, 128, 64, 0),     # two strips per coarse row
    (1, 8, 64, 72, 40, 1),       # channel tails on both operands; fine grid (2H + 1) x (2W + 1)
    (3, 16, 16, 1024, 512, 0),   # unet's up_convolution_1 shape: 64 tiles
])
def test_gather4_exact_on_integers(N, H, W, Cin, Cout, odd):
    g = torch.Generator().manual_seed(17)
    x = torch.randint(-2, 3, (N, Cin, H, W), generator=g).float()
    dy = torch.randint(-2, 3, (N, Cout, 2 * H + odd, 2 * W + odd), generator=g).float()
    ref = ref_convt_wgrad(x, dy)
    xa, dya = act_from_nchw(x.to(DEV), dt), act_from_nchw(dy.to(DEV), dt)
    d = L.WgradDesc(L.dtype_code(dt), xa.N, xa.H, xa.W, dya.H, dya.W, xa.C, xa.ld, dya.C, dya.ld, 4, L.TAPS_GATHER2X2, 1)
    assert ops.wgrad_kernel_name(d).startswith("wgrad_g4_bf16_" + ("128" if Cin > 64 else "64")), ops.wgrad_kernel_name(d)
    got = ops.wgrad(xa, dya, (Cin, Cout, 2, 2), ntaps=4, taps_mode=L.TAPS_GATHER2X2)
    assert torch.equal(got.cpu(), ref)
    assert torch.equal(got, ops.wgrad(xa, dya, (Cin, Cout, 2, 2), ntaps=4, taps_mode=L.TAPS_GATHER2X2))


def test_gather4_channel_windows_of_poisoned_buffers():
    """g is the up-slot of a concat buffer (cat([up, skip], 1)), x a window of a wider buffer; the neighbours are NaN"""
    N, H, W, Cin, Cout = 2, 16, 32, 128, 64
    g = torch.Generator().manual_seed(19)
    x = torch.randn(N, Cin, H, W, generator=g).to(dt).float()
    dy = torch.randn(N, Cout, 2 * H, 2 * W, generator=g).to(dt).float()
    ref = ref_convt_wgrad(x, dy)
    xw = torch.full((N * H * W, Cin + 64), float("nan"), dtype=dt, device=DEV)
    xw[:, 32:32 + Cin] = act_from_nchw(x.to(DEV), dt).buf
    gw = torch.full((N * 4 * H * W, 2 * Cout), float("nan"), dtype=dt, device=DEV)
    gw[:, :Cout] = act_from_nchw(dy.to(DEV), dt).buf
    got = ops.wgrad(Act(xw, 32, Cin, N, H, W), Act(gw, 0, Cout, N, 2 * H, 2 * W), (Cin, Cout, 2, 2), ntaps=4,
                    taps_mode=L.TAPS_GATHER2X2)
    assert torch.isfinite(got).all() and relerr(got.cpu(), ref) < 2e-2


# ---- uz_wgrad_multi: several problems issued together ------------------------------------------------------------------
def _multi_entries(g, shapes):
    ents = []
    for (N, H, W, Co, Ci) in shapes:
        dy = ops.Act(torch.randint(-2, 3, (N * H * W, Co), generator=g).to(torch.bfloat16).to(DEV), 0, Co, N, H, W)
        x = ops.Act(torch.randint(-2, 3, (N * H * W, Ci), generator=g).to(torch.bfloat16).to(DEV), 0, Ci, N, H, W)
        ents.append((dy, x, torch.full((Co, Ci), float("nan"), device=DEV), 1, L.TAPS_CONV, 1))
    return ents


def test_wgrad_multi_linear_layers_exact_on_integers():
    """the nn.Linear weight gradients of a swin-like backward range in one uz_wgrad_multi call: 128 x 128 and 64 x 64 tile
    groups, token counts that are no multiple of 64, more problems than one launch carries; integer operands, so every
    summation order gives the same fp32 result as the double-precision product"""
    g = torch.Generator().manual_seed(5)
    shapes = [(2, 56, 56, 288, 96), (2, 56, 56, 96, 96), (2, 56, 56, 384, 96), (2, 56, 56, 96, 384), (2, 28, 28, 576, 192),
              (2, 28, 28, 192, 768), (2, 14, 14, 1152, 384), (2, 7, 7, 768, 3072), (1, 7, 7, 40, 24), (3, 5, 9, 64, 64)]
    shapes = shapes + [(1, 14, 14, 32 + 8 * i, 48) for i in range(30)]      # > 24 problems of the 64 x 64 tile shape
    ents = _multi_entries(g, shapes)
    ops.wgrad_multi(ents)
    for (dy, x, out, *_), sh in zip(ents, shapes):
        ref = dy.buf.double().t() @ x.buf.double()
        assert torch.equal(out.double(), ref), sh


def test_wgrad_multi_mixes_shared_launches_with_single_problems():
    """a nine-tap convolution problem and an fp32 problem in the same call run through uz_wgrad; results as uz_wgrad's"""
    g = torch.Generator().manual_seed(6)
    ents = _multi_entries(g, [(2, 32, 32, 128, 64), (1, 16, 16, 256, 256)])
    N, H, W, C = 2, 32, 32, 64
    dy = ops.Act(torch.randn(N * H * W, C, generator=g).to(torch.bfloat16).to(DEV), 0, C, N, H, W)
    x = ops.Act(torch.randn(N * H * W, C, generator=g).to(torch.bfloat16).to(DEV), 0, C, N, H, W)
    o9 = torch.empty(C, C, 3, 3, device=DEV)
    ents.append((dy, x, o9, 9, L.TAPS_CONV, 1))
    dyf = ops.Act(torch.randn(200, 24, generator=g).to(DEV), 0, 24, 1, 10, 20)
    xf = ops.Act(torch.randn(200, 16, generator=g).to(DEV), 0, 16, 1, 10, 20)
    of = torch.empty(24, 16, device=DEV)
    ents.append((dyf, xf, of, 1, L.TAPS_CONV, 1))
    ops.wgrad_multi(ents)
    assert torch.equal(o9, ops.wgrad(dy, x, (C, C, 3, 3), ntaps=9))
    assert torch.equal(of, ops.wgrad(dyf, xf, (24, 16), ntaps=1))
    for dyb, xb, out, *_ in ents[:2]:
        assert torch.equal(out.double(), dyb.buf.double().t() @ xb.buf.double())


@pytest.mark.parametrize("B,img,ws", [(2, 64, 4), (1, 224, 7)])
def test_swin_step_with_and_without_deferred_linear_weight_gradients(B, img, ws):
    """the engine's deferred nn.Linear weight gradients (Engine.defer_linear_wgrads) against the one-by-one launches on a
    swin_unet_v2 backward: same gradients up to the summation order of the pixel split.  At 224 x 224 / window 7 the
    PatchExpand weight gradients (token maps 7 / 14 / 28 wide) take the space-to-depth + Linear route in the deferred run
    and the 2 x 2 gather kernel in the other."""
    import unet_zoo_amd
    from unet_zoo_amd.engine import Engine
    x = torch.randn(B, 3, img, img, generator=torch.Generator().manual_seed(1)).to(DEV)
    grads, fams = [], []
    for defer in (True, False):
        Engine.defer_linear_wgrads = defer
        try:
            torch.manual_seed(0)
            m = unet_zoo_amd.create_model("swin_unet_v2", image_size=img, in_channels=3, num_classes=1, window_size=ws,
                                          drop_path_rate=0.0)
            m.run_dtype = torch.bfloat16
            m = m.to(DEV).train()
            ops.profile_begin()
            m(x).float().mean().backward()
            fams.append(set(ops.profile_end()))
            grads.append({n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
        finally:
            Engine.defer_linear_wgrads = True
    assert "wgrad_multi" in fams[0] and "wgrad_multi" not in fams[1]
    assert grads[0].keys() == grads[1].keys() and len(grads[0]) > 50
    for n in grads[0]:
        a, b = grads[0][n].double(), grads[1][n].double()
        assert (a - b).abs().max() <= 1e-4 * b.abs().max().item() + 1e-9, n


# ---- dilated 3x3 weight gradients (u2net's RSU4F) on the LDS-DMA kernel ---------------------------------------------------
@pytest.mark.parametrize("N,H,W,Cx,Cdy,dil", [(8, 32, 32, 256, 512, 2), (8, 16, 16, 512, 256, 4), (2, 16, 16, 256, 256, 8),
                                              (2, 32, 32, 64, 40, 2), (1, 32, 64, 128, 128, 4)])
def test_dilated_weight_gradient_on_the_gather_kernel(N, H, W, Cx, Cdy, dil):
    """Conv2d(k3, padding=d, dilation=d) weight gradient (REBNCONV dirate 2 / 4 / 8, u2net.py:10-13) as nine displaced
    one-tap problems: the kernel family by name, exact on integers against autograd, dilation beyond the map's half width"""
    g = torch.Generator().manual_seed(dil * 7 + H)
    x = torch.randint(-2, 3, (N, Cx, H, W), generator=g).float()
    dy = torch.randint(-2, 3, (N, Cdy, H, W), generator=g).float()
    w = torch.zeros(Cdy, Cx, 3, 3, requires_grad=True)
    F.conv2d(x, w, None, padding=dil, dilation=dil).backward(dy)
    La, Ra = act_from_nchw(dy.to(DEV), torch.bfloat16), act_from_nchw(x.to(DEV), torch.bfloat16)
    d = L.WgradDesc(L.dtype_code(torch.bfloat16), N, H, W, H, W, Cdy, Cdy, Cx, Cx, 9, L.TAPS_CONV, dil)
    assert ops.wgrad_kernel_name(d).endswith("_dilated9"), ops.wgrad_kernel_name(d)
    out = ops.wgrad(La, Ra, (Cdy, Cx, 3, 3), ntaps=9, dil=dil)
    assert torch.equal(out.cpu(), w.grad)
    assert torch.equal(ops.wgrad(La, Ra, (Cdy, Cx, 3, 3), ntaps=9, dil=dil), out)
