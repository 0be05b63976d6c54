"""GPU: the split-K form of the LDS-DMA GEMM (uz_gemm_dma.hip: few tiles, long K loop -- Linear layers on the 8 x 8 / 16 x 16
token maps of swin_unet_v2, swin_unet_v2.py:126-159; the spatial-reduction products of MISSFormer): fp32 partial tiles +
fixed-order reduce with bias / residual.  Integer operands: every fp32 sum is exact, so the bf16 result is the correctly
rounded one whatever the split, bit for bit."""
import pytest
import torch
from ctypes import byref

pytestmark = pytest.mark.gpu

from unet_zoo_amd import _lib as L
from unet_zoo_amd import ops
from unet_zoo_amd.ops import Act

DEV = "cuda"
dt = torch.bfloat16


# (batch, map, map, K, N, split expected): the plan splits <= 32 tiles from 16 K slabs on and <= 64 tiles from 48 on
@pytest.mark.parametrize("N,H,W,Cin,Cout,split", [(16, 8, 8, 2304, 768, False), (16, 16, 16, 1536, 384, False), (2, 8, 8, 3072, 768, True),
                                                  (4, 8, 8, 2304, 768, True), (16, 8, 8, 3072, 768, True), (1, 7, 7, 1032, 72, True),
                                                  (16, 8, 8, 768, 768, False)])
@pytest.mark.parametrize("with_res", [False, True])
def test_split_k_gemm_is_exact_on_integers(N, H, W, Cin, Cout, split, with_res):
    g = torch.Generator().manual_seed(Cin + Cout)
    P = N * H * W
    x = torch.randint(-2, 3, (P, Cin), generator=g).to(dt)
    w = torch.randint(-2, 3, (Cout, Cin), generator=g).to(dt)
    b = torch.randint(-3, 4, (Cout,), generator=g).float()
    r = torch.randint(-8, 9, (P, Cout), generator=g).to(dt)
    xa = Act(x.to(DEV), 0, Cin, N, H, W)
    y = ops.new_act(N, H, W, Cout, dt, DEV)
    ra = Act(r.to(DEV), 0, Cout, N, H, W) if with_res else None
    d = L.ConvDesc(L.dtype_code(dt), N, H, W, H, W, Cin, Cin, Cout, Cout, 1, L.TAPS_CONV, 1, L.STORE_PLAIN, 0, 0, 0)
    wsb = L.load().uz_conv_igemm_workspace_bytes(byref(d))
    ops.conv_igemm(xa, w.to(DEV), b.to(DEV), y, ntaps=1, res=ra)
    ref = (x.double() @ w.double().t() + b.double()).to(dt)          # one rounding of the exact sum
    if with_res:
        ref = (ref.float() + r.float()).to(dt)                        # the residual is a separate add of the stored result
    assert torch.equal(y.buf.cpu(), ref), (wsb, (y.buf.cpu().float() - ref.float()).abs().max())
    assert (wsb > 0) == split, wsb                                   # which form the plan took
    y2 = ops.new_act(N, H, W, Cout, dt, DEV)
    ops.conv_igemm(xa, w.to(DEV), b.to(DEV), y2, ntaps=1, res=ra)
    assert torch.equal(y2.buf, y.buf)


def test_split_k_gemm_random_operands_against_fp32():
    g = torch.Generator().manual_seed(3)
    N, H, W, Cin, Cout = 16, 8, 8, 2304, 768
    P = N * H * W
    x, w = torch.randn(P, Cin, generator=g).to(dt), (torch.randn(Cout, Cin, generator=g) * 0.05).to(dt)
    b = torch.randn(Cout, generator=g)
    y = ops.new_act(N, H, W, Cout, dt, DEV)
    ops.conv_igemm(Act(x.to(DEV), 0, Cin, N, H, W), w.to(DEV), b.to(DEV), y, ntaps=1)
    ref = x.float() @ w.float().t() + b
    err = ((y.buf.cpu().float() - ref).abs().max() / ref.abs().max()).item()
    assert err < 1e-2, err


@pytest.mark.parametrize("N,H,W,Cin,Cout,dil", [(8, 16, 16, 512, 512, 2), (8, 32, 32, 512, 256, 4), (2, 16, 16, 256, 512, 8)])
def test_split_k_dilated_convolution_with_statistics(N, H, W, Cin, Cout, dil):
    """the dilated nine-tap form (REBNCONV dirate 2 / 4 / 8 of RSU4F, u2net.py:10-13: K = 9 Cin) split over K, BatchNorm
    partial sums from the reduce pass: exact on integers against F.conv2d, sums of the stored values"""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(dil)
    x = torch.randint(-2, 3, (N, Cin, H, W), generator=g).float()
    w = torch.randint(-1, 2, (Cout, Cin, 3, 3), generator=g).float()
    ref = F.conv2d(x, w, None, padding=dil, dilation=dil).to(dt)
    xa = ops.act_from_nchw(x.to(DEV), dt)
    y = ops.new_act(N, H, W, Cout, dt, DEV)
    d = L.ConvDesc(L.dtype_code(dt), N, H, W, H, W, Cin, Cin, Cout, Cout, 9, L.TAPS_CONV, dil, L.STORE_PLAIN, 0, 0, 0)
    assert L.load().uz_conv_igemm_workspace_bytes(byref(d)) > 0
    stats = ops.conv_igemm(xa, ops.pack_weights(w.to(DEV), L.PACK_CONV_FWD, dt), None, y, ntaps=9, dil=dil, want_stats=True)
    got = y.buf.view(N, H, W, Cout).permute(0, 3, 1, 2).cpu()
    assert torch.equal(got, ref)
    s = stats.double().sum(0).cpu()
    assert torch.allclose(s[0], ref.double().sum((0, 2, 3)), rtol=1e-6, atol=1e-3)
    assert torch.allclose(s[1], (ref.double() ** 2).sum((0, 2, 3)), rtol=1e-6, atol=1e-2)
