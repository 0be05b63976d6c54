"""GPU: the MISSFormer / MiT kernels (uz_mit.hip) through the C ABI against plain PyTorch fp32 on the CPU:
spatial-reduction attention (missformer.py:21-39, :113-128), DWConv (:168-177), GELU, space-to-depth, im2col."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from unet_zoo_amd import ops
from unet_zoo_amd.ops import act_from_nchw

DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16]


def rnd(dt, t):
    return t.to(dt).float()


def relerr(a, b):
    return ((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30)).item()


def tokens(t, dt):
    """(B, N, C) fp32 CPU -> Act with N = B, H = 1, W = N"""
    B, N, C = t.shape
    return act_from_nchw(t.permute(0, 2, 1).reshape(B, C, 1, N).contiguous().to(DEV), dt)


def untokens(a):
    return a.dense().cpu().reshape(a.N, a.C, a.H * a.W).permute(0, 2, 1)


@pytest.mark.parametrize("dt", DTYPES)
def test_gelu_forward_backward(dt):
    g = torch.Generator().manual_seed(5)
    x = rnd(dt, torch.randn(2, 24, 7, 9, generator=g) * 2).requires_grad_(True)
    dy = rnd(dt, torch.randn(2, 24, 7, 9, generator=g))
    ref = F.gelu(x)
    ref.backward(dy)
    xa = act_from_nchw(x.detach().to(DEV), dt)
    y = ops.new_act(2, 7, 9, 24, dt, DEV)
    ops.gelu_fwd(xa, y)
    dx = ops.new_act(2, 7, 9, 24, dt, DEV)
    ops.gelu_bwd(xa, act_from_nchw(dy.to(DEV), dt), dx)
    tol = 2e-6 if dt == torch.float32 else 8e-3
    assert relerr(y.dense().cpu(), ref.detach()) < tol
    assert relerr(dx.dense().cpu(), x.grad) < tol


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("N,C,H,W,skip", [(2, 64, 16, 16, True), (1, 24, 5, 19, False), (2, 256, 9, 8, True)])
def test_dwconv3x3_forward_input_gradient_weight_gradient(dt, N, C, H, W, skip):
    g = torch.Generator().manual_seed(6)
    x = rnd(dt, torch.randn(N, C, H, W, generator=g)).requires_grad_(True)
    w = (torch.randn(C, 1, 3, 3, generator=g) * 0.3).requires_grad_(True)
    b = (torch.randn(C, generator=g) * 0.1).requires_grad_(True)
    dy = rnd(dt, torch.randn(N, C, H, W, generator=g))
    ref = F.conv2d(x, w, b, padding=1, groups=C)
    if skip:
        ref = ref + x
    ref.backward(dy)
    wt = w.detach().reshape(C, 9).t().contiguous().to(DEV)
    xa, ga = act_from_nchw(x.detach().to(DEV), dt), act_from_nchw(dy.to(DEV), dt)
    y, dx = ops.new_act(N, H, W, C, dt, DEV), ops.new_act(N, H, W, C, dt, DEV)
    ops.dwconv3x3(xa, wt, b.detach().to(DEV), y, skip=skip)
    ops.dwconv3x3(ga, wt, None, dx, skip=skip, flip=True)
    dwb = ops.dwconv3x3_wgrad(xa, ga).cpu()
    tol = 5e-6 if dt == torch.float32 else 1e-2
    assert relerr(y.dense().cpu(), ref.detach()) < tol
    assert relerr(dx.dense().cpu(), x.grad) < tol
    assert relerr(dwb[:9].t().reshape(C, 1, 3, 3), w.grad) < 1e-4
    assert relerr(dwb[9], b.grad) < 1e-4


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("r", [2, 4, 8])
def test_space_to_depth_and_back(dt, r):
    g = torch.Generator().manual_seed(7)
    N, C, Ho, Wo = 2, 16, 3, 2
    x = rnd(dt, torch.randn(N, C, Ho * r, Wo * r, generator=g))
    xa = act_from_nchw(x.to(DEV), dt)
    d = ops.new_act(N, Ho, Wo, r * r * C, dt, DEV)
    ops.space_to_depth(xa, d, r)
    ref = x.reshape(N, C, Ho, r, Wo, r).permute(0, 3, 5, 1, 2, 4).reshape(N, r * r * C, Ho, Wo)   # (ty, tx, c)
    assert torch.equal(d.dense().cpu(), ref)
    back = ops.new_act(N, Ho * r, Wo * r, C, dt, DEV)
    ops.space_to_depth(d, back, r, inverse=True)
    assert torch.equal(back.dense().cpu(), x)


@pytest.mark.parametrize("dt", DTYPES)
def test_im2col_nchw_7x7_stride4(dt):
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 3, 32, 40, generator=g)
    a = ops.im2col_nchw(x.to(DEV), 7, 4, 3, 160, dt)
    assert (a.N, a.H, a.W, a.C) == (2, 8, 10, 160)
    cols = F.unfold(x, 7, padding=3, stride=4)                       # (N, C*49, L), index c*49 + tap
    ref = cols.reshape(2, 3, 49, 80).permute(0, 3, 2, 1).reshape(2, 80, 147)
    got = a.buf.float().cpu().reshape(2, 80, 160)
    assert torch.equal(got[:, :, :147], rnd(dt, ref)) and got[:, :, 147:].abs().max() == 0


def _sra_reference(q, kv, heads, scale):
    B, N, C = q.shape
    d = C // heads
    qh = q.reshape(B, N, heads, d).permute(0, 2, 1, 3)
    kvh = kv.reshape(B, -1, 2, heads, d).permute(2, 0, 3, 1, 4)
    attn = ((qh @ kvh[0].transpose(-2, -1)) * scale).softmax(dim=-1)
    return (attn @ kvh[1]).transpose(1, 2).reshape(B, N, C)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,N,NK,heads,nseg", [(2, 256, 16, 1, 1), (1, 200, 49, 2, 1), (2, 1000, 256, 1, 1),
                                                (2, 340, 64, 1, 4), (1, 128, 288, 5, 1), (3, 49, 49, 8, 1)])
def test_spatial_reduction_attention_forward_backward(dt, B, N, NK, heads, nseg):
    g = torch.Generator().manual_seed(9)
    C = heads * 64
    q = rnd(dt, torch.randn(B, N, C, generator=g)).requires_grad_(True)
    kv = rnd(dt, torch.randn(B, NK, 2 * C, generator=g)).requires_grad_(True)
    go = rnd(dt, torch.randn(B, N, C, generator=g))
    scale = 64 ** -0.5
    ref = _sra_reference(q, kv, heads, scale)
    ref.backward(go)
    kps = NK // nseg
    # device layout of the keys: segment-major blocks of [B][kps] rows
    kv_dev = kv.detach().reshape(B, nseg, kps, 2 * C).permute(1, 0, 2, 3).reshape(1, nseg * B * kps, 2 * C)
    qa, kva, goa = tokens(q.detach(), dt), tokens(kv_dev, dt), tokens(go, dt)
    out, dq, dkv = ops.new_act(B, 1, N, C, dt, DEV), ops.new_act(B, 1, N, C, dt, DEV), ops.new_act(1, 1, B * NK, 2 * C, dt, DEV)
    lse = ops.sra_fwd(qa, kva, out, B, heads, kps, scale)
    ops.sra_bwd(qa, kva, out, lse, goa, dq, dkv, B, heads, kps, scale)
    tol = 2e-5 if dt == torch.float32 else 2e-2
    assert relerr(untokens(out), ref.detach()) < tol
    assert relerr(untokens(dq), q.grad) < tol
    dkv_ref = kv.grad.reshape(B, nseg, kps, 2 * C).permute(1, 0, 2, 3).reshape(1, nseg * B * kps, 2 * C)
    assert relerr(untokens(dkv), dkv_ref) < tol


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("C", [64, 256, 1280, 2048])
def test_layernorm_with_gelu_forward_backward(dt, C):
    """act(norm1(.)) of MixFFN_skip (missformer.py:206) as one kernel each way (uz_ln_desc.act = 1)"""
    g = torch.Generator().manual_seed(10)
    B, N = 2, 37
    x = rnd(dt, torch.randn(B, N, C, generator=g) * 1.3 + 0.2).requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.randn(C, generator=g) * 0.3).requires_grad_(True)
    go = rnd(dt, torch.randn(B, N, C, generator=g))
    ref = F.gelu(F.layer_norm(x, (C,), gamma, beta, 1e-5))
    ref.backward(go)
    xa, ga = tokens(x.detach(), dt), tokens(go, dt)
    y, dx = ops.new_act(B, 1, N, C, dt, DEV), ops.new_act(B, 1, N, C, dt, DEV)
    gd, bd = gamma.detach().to(DEV), beta.detach().to(DEV)
    stats = ops.layernorm_fwd(xa, gd, bd, y, gelu=True)
    dgam, dbet = ops.layernorm_bwd(xa, gd, stats, ga, dx, gelu_beta=bd)
    tol = 5e-6 if dt == torch.float32 else 1e-2
    assert relerr(untokens(y), ref.detach()) < tol
    assert relerr(untokens(dx), x.grad) < (2e-5 if dt == torch.float32 else 2e-2)
    assert relerr(dgam.cpu(), gamma.grad) < 2e-4 and relerr(dbet.cpu(), beta.grad) < 2e-4


@pytest.mark.parametrize("dt", DTYPES)
def test_batched_column_sums(dt):
    """uz_colsum_batched: the bias gradients of a backward range in two launches (more than one launch pair's worth
    of tensors, ragged row counts, a channel window of a wider buffer)"""
    g = torch.Generator().manual_seed(12)
    shapes = [(1, 1, 7, 8), (2, 5, 9, 64), (1, 64, 64, 320), (3, 16, 16, 2048), (2, 128, 128, 64), (1, 3, 1, 24)] * 15
    acts, outs, refs = [], [], []
    for i, (N, H, W, C) in enumerate(shapes):
        t = rnd(dt, torch.randn(N, C, H, W, generator=g))
        a = act_from_nchw(t.to(DEV), dt)
        if i % 4 == 1:                      # the middle window of a 3x wider buffer
            wide = ops.new_act(N, H, W, 3 * C, dt, DEV)
            wide.buf.normal_()
            wide.buf[:, C:2 * C] = a.buf
            a = wide.window(C, C)
        acts.append(a)
        outs.append(torch.full((C,), float("nan"), device=DEV))
        refs.append(t.double().sum(dim=(0, 2, 3)))
    ops.colsum_batched(list(zip(acts, outs)))
    again = [torch.empty_like(o) for o in outs]
    ops.colsum_batched(list(zip(acts, again)))
    for o, o2, r in zip(outs, again, refs):
        assert torch.equal(o, o2)
        assert (o.double().cpu() - r).abs().max() <= 2e-5 * (r.abs().max() + 1.0)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("P,K,Nout", [(300, 64, 64), (1000, 320, 96), (77, 128, 512)])
def test_linear_with_residual_in_the_gemm_epilogue(dt, P, K, Nout):
    """uz_conv_igemm_res: y = x W^T + b + res, equal to the stored GEMM result plus res (same rounding)"""
    from unet_zoo_amd import _lib as L
    g = torch.Generator().manual_seed(13)
    x = tokens(rnd(dt, torch.randn(1, P, K, generator=g)), dt)
    res = tokens(rnd(dt, torch.randn(1, P, Nout, generator=g)), dt)
    w = torch.randn(Nout, K, generator=g) * 0.1
    b = torch.randn(Nout, generator=g).to(DEV)
    wp = ops.pack_weights(w.to(DEV), L.PACK_CONV_FWD, dt, 0)
    y0, y1 = ops.new_act(1, 1, P, Nout, dt, DEV), ops.new_act(1, 1, P, Nout, dt, DEV)
    ops.conv_igemm(x, wp, b, y0, ntaps=1)
    ops.conv_igemm(x, wp, b, y1, ntaps=1, res=res)
    ref = (y0.buf.float() + res.buf.float()).to(dt)
    assert torch.equal(y1.buf, ref)
