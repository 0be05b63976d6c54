"""GPU: the dense token attention of U-Transformer on the library's own kernels: batched NT products
(uz_gemm_nt), softmax over either axis and its gradient, F.adaptive_avg_pool2d, and Engine.token_attention as a whole
against torch autograd on the formula of unet_zoo/models/unet_transformer.py:126-137 / :200-213."""
import math

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from unet_zoo_amd import ops
from unet_zoo_amd.engine import Engine
from unet_zoo_amd.ops import Act, act_from_nchw

DEV = "cuda"


def relerr(a, b):
    return ((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30)).item()


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("batch,M,N,K,pad,shared_x,shared_w", [
    (1, 300, 72, 40, 0, False, False),       # one matrix, ragged M tile, N below one tile
    (3, 200, 136, 72, 8, False, False),      # batch, row strides longer than the rows, partial N tile
    (4, 64, 512, 128, 0, True, False),       # x shared by the batch (the transposed projections)
    (2, 520, 256, 512, 16, False, True),     # w shared, several K slabs, three M tiles
    (16, 96, 96, 64, 0, False, False),       # many small matrices
])
def test_gemm_nt_matches_bmm(dt, batch, M, N, K, pad, shared_x, shared_w):
    g = torch.Generator().manual_seed(batch * 1000 + M)
    ldx, ldw, ldy = K + pad, K + pad, N + pad
    x = torch.randn(1 if shared_x else batch, M, ldx, generator=g).to(dt).to(DEV)
    w = torch.randn(1 if shared_w else batch, N, ldw, generator=g).to(dt).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    res = torch.randn(batch, M, ldy, generator=g).to(dt).to(DEV)
    y = torch.full((batch, M, ldy), 7.0, dtype=dt, device=DEV)
    ops.gemm_nt(dt, batch, M, N, K, x.data_ptr(), ldx, 0 if shared_x else M * ldx, w.data_ptr(), ldw, 0 if shared_w else N * ldw,
                y.data_ptr(), ldy, M * ldy, bias=bias, res_ptr=res.data_ptr(), ldres=ldy, resb=M * ldy)
    prod = torch.matmul(x[..., :K].double(), w[..., :K].double().transpose(1, 2)) + bias.double()
    if dt == torch.bfloat16:
        prod = prod.to(dt).double()                     # the product is rounded to the run dtype before the residual add
    ref = prod + res[..., :N].double()
    tol = 1e-5 if dt == torch.float32 else 1.2e-2
    assert relerr(y[..., :N].cpu(), ref.cpu()) < tol
    if pad:
        assert torch.all(y[..., N:] == 7.0)            # nothing written beyond the N columns
    y2 = torch.empty((batch, M, ldy), dtype=dt, device=DEV)
    ops.gemm_nt(dt, batch, M, N, K, x.data_ptr(), ldx, 0 if shared_x else M * ldx, w.data_ptr(), ldw, 0 if shared_w else N * ldw,
                y2.data_ptr(), ldy, M * ldy)
    ref2 = torch.matmul(x[..., :K].double(), w[..., :K].double().transpose(1, 2))
    assert relerr(y2[..., :N].cpu(), ref2.cpu()) < tol


def test_gemm_nt_refuses_what_it_cannot_address():
    from unet_zoo_amd._lib import HipLibraryError
    t = torch.zeros(64, 64, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(HipLibraryError):
        ops.gemm_nt(torch.bfloat16, 1, 64, 64, 60, t.data_ptr(), 64, 0, t.data_ptr(), 64, 0, t.data_ptr(), 64, 0)   # K % 8
    with pytest.raises(HipLibraryError):
        ops.gemm_nt(torch.bfloat16, 1, 64, 64, 64, t.data_ptr(), 32, 0, t.data_ptr(), 64, 0, t.data_ptr(), 64, 0)   # ldx < K
    with pytest.raises(HipLibraryError):
        ops.gemm_nt(torch.bfloat16, 1, 64, 64, 64, t.data_ptr() + 2, 64, 0, t.data_ptr(), 64, 0, t.data_ptr(), 64, 0)   # alignment


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("axis,B,R,C", [(0, 3, 384, 384), (0, 2, 1000, 136), (0, 1, 5, 8), (1, 3, 256, 256), (1, 2, 77, 512),
                                        (1, 1, 9, 1024)])
def test_softmax_forward_and_gradient(dt, axis, B, R, C):
    g = torch.Generator().manual_seed(R + C)
    s = (3.0 * torch.randn(B, R, C, generator=g)).to(dt)
    scale = 0.37
    a_ref = torch.softmax(s.double() * scale, dim=1 + axis)
    a = s.clone().to(DEV)
    ops.softmax_fwd(a, axis, scale)
    tol = 2e-6 if dt == torch.float32 else 6e-3
    assert relerr(a.cpu(), a_ref) < tol
    sums = a.double().sum(dim=1 + axis).cpu()
    assert (sums - 1).abs().max() < (1e-5 if dt == torch.float32 else 2e-2)
    da = torch.randn(B, R, C, generator=g).to(dt)
    ad = a.double().cpu()
    ref = ad * (da.double() - (ad * da.double()).sum(dim=1 + axis, keepdim=True)) * scale
    gdev = da.clone().to(DEV)
    ops.softmax_bwd(a, gdev, axis, scale)
    assert relerr(gdev.cpu(), ref) < (1e-5 if dt == torch.float32 else 1e-2)
    if axis == 0:          # the caller may hand the column sums over
        dot = (ad * da.double()).sum(dim=1).float().to(DEV).contiguous()
        g2 = da.clone().to(DEV)
        ops.softmax_bwd(a, g2, 0, scale, dot)
        assert relerr(g2.cpu(), ref) < (1e-5 if dt == torch.float32 else 1e-2)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Hi,Wi,Ho,Wo", [(32, 32, 64, 64), (128, 128, 64, 64), (12, 20, 16, 24), (7, 9, 3, 4), (16, 24, 16, 24)])
def test_adaptive_avg_pool_matches_torch(dt, Hi, Wi, Ho, Wo):
    g = torch.Generator().manual_seed(Hi * Wo)
    N, C = 2, 24
    x = torch.randn(N, C, Hi, Wi, generator=g).to(dt).float()
    xr = x.clone().requires_grad_(True)
    ref = F.adaptive_avg_pool2d(xr, (Ho, Wo))
    dy = torch.randn(ref.shape, generator=g).to(dt).float()
    ref.backward(dy)
    eng = Engine(dt, torch.device(DEV), True, True)
    xa = act_from_nchw(x.to(DEV), dt)
    y = eng.adaptive_avg_pool(xa, Ho, Wo)
    if (Hi, Wi) == (Ho, Wo):
        assert y is xa
        return
    tol = 1e-6 if dt == torch.float32 else 6e-3
    assert relerr(y.dense().cpu(), ref.detach()) < tol
    y.add_grad(act_from_nchw(dy.to(DEV), dt))
    eng.backward_range(None, len(eng.tape), 0)
    assert relerr(xa.grads[0].dense().cpu(), xr.grad) < tol


def _attention_reference(xq, xv, wq, wk, wv):
    """unet_transformer.py:126-137 on (b, c, h, w) sources"""
    b, c, h, w = xq.shape
    tq = xq.flatten(2).permute(0, 2, 1)
    tv = xv.flatten(2).permute(0, 2, 1)
    Q, K, V = tq @ wq, tq @ wk, tv @ wv
    A = torch.softmax(torch.bmm(Q, K.permute(0, 2, 1)) / math.sqrt(c), dim=1)
    return torch.bmm(A, V).permute(0, 2, 1).reshape(b, c, h, w)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,c,H,W,self_attn", [(2, 64, 16, 24, False), (3, 128, 8, 8, True), (2, 256, 32, 32, False),
                                                (1, 72, 5, 8, False)])
def test_token_attention_forward_and_backward_match_autograd(dt, B, c, H, W, self_attn):
    g = torch.Generator().manual_seed(c + H)
    xq = torch.randn(B, c, H, W, generator=g).to(dt).float()
    xv = xq if self_attn else torch.randn(B, c, H, W, generator=g).to(dt).float()
    ws = [nn.Parameter((torch.randn(c, c, generator=g) * (1.5 / math.sqrt(c))).to(dt).float()) for _ in range(3)]
    xq_r = xq.double().clone().requires_grad_(True)
    xv_r = xq_r if self_attn else xv.double().clone().requires_grad_(True)
    ws_r = [w.detach().double().clone().requires_grad_(True) for w in ws]
    ref = _attention_reference(xq_r, xv_r, *ws_r)
    dy = torch.randn(ref.shape, generator=g).to(dt).double()
    ref.backward(dy)

    ws_d = [nn.Parameter(w.detach().to(DEV)) for w in ws]
    eng = Engine(dt, torch.device(DEV), True, True)
    qa = act_from_nchw(xq.to(DEV), dt)
    va = qa if self_attn else act_from_nchw(xv.to(DEV), dt)
    out = eng.token_attention(qa, va, *ws_d, eng.new_act(B, H, W, c))
    ftol, gtol = (2e-5, 1e-4) if dt == torch.float32 else (2e-2, 4e-2)
    assert relerr(out.dense().cpu(), ref.detach()) < ftol
    out.add_grad(act_from_nchw(dy.float().to(DEV), dt))
    eng.backward_range(None, len(eng.tape), 0)

    def l2(a, b):
        return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()

    for w_d, w_r, name in zip(ws_d, ws_r, ("wq", "wk", "wv")):
        e = l2(eng.param_grads[w_d].cpu(), w_r.grad)
        assert e < gtol, (name, e)
    gq = qa.grads[0].dense().cpu() if len(qa.grads) == 1 else sum(a.dense().cpu() for a in qa.grads)
    assert l2(gq, xq_r.grad) < gtol
    if not self_attn:
        assert l2(va.grads[0].dense().cpu(), xv_r.grad) < gtol


def test_rowdot_and_cast_rows():
    g = torch.Generator().manual_seed(11)
    a = torch.randn(300, 136, generator=g).to(DEV)
    b = torch.randn(300, 136, generator=g).to(torch.bfloat16).to(DEV)
    out = ops.rowdot_f32(a, b)
    assert relerr(out.cpu(), (a.double() * b.double()).sum(1).cpu()) < 1e-5
    dst = ops.new_act(3, 10, 10, 136, torch.bfloat16, DEV)
    ops.cast_rows(a, dst)
    assert torch.equal(dst.buf, a.to(torch.bfloat16))
    ops.cast_rows(a, dst, accumulate=True)
    assert torch.equal(dst.buf, (a.to(torch.bfloat16).float() + a).to(torch.bfloat16))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,C,H,W", [(2, 64, 4, 6), (3, 128, 16, 16)])
def test_row_attention_matches_autograd(dt, B, C, H, W):
    """PAM_Module's core (transatt_unet.py:41-49) on q / k / v given as token maps"""
    g = torch.Generator().manual_seed(C + W)
    q = torch.randn(B, C // 8, H, W, generator=g).to(dt).double().requires_grad_(True)
    k = torch.randn(B, C // 8, H, W, generator=g).to(dt).double().requires_grad_(True)
    v = torch.randn(B, C, H, W, generator=g).to(dt).double().requires_grad_(True)
    att = torch.softmax(torch.bmm(q.flatten(2).transpose(1, 2), k.flatten(2)), dim=-1)
    ref = torch.bmm(v.flatten(2), att.transpose(1, 2)).view(B, C, H, W)
    dy = torch.randn(ref.shape, generator=g).to(dt).double()
    ref.backward(dy)
    eng = Engine(dt, torch.device(DEV), True, True)
    qa, ka, va = (act_from_nchw(t.detach().float().to(DEV), dt) for t in (q, k, v))
    out = eng.row_attention(qa, ka, va, eng.new_act(B, H, W, C))
    ftol, gtol = (2e-5, 1e-4) if dt == torch.float32 else (2e-2, 4e-2)
    assert relerr(out.dense().cpu(), ref.detach()) < ftol
    out.add_grad(act_from_nchw(dy.float().to(DEV), dt))
    eng.backward_range(None, len(eng.tape), 0)
    for a, r, name in ((qa, q, "q"), (ka, k, "k"), (va, v, "v")):
        e = ((a.grads[0].dense().cpu().double() - r.grad).norm() / r.grad.norm()).item()
        assert e < gtol, (name, e)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,d,H,W,p", [(2, 64, 4, 6, 0.0), (2, 512, 16, 16, 0.0), (2, 128, 8, 8, 0.25)])
def test_channel_attention_matches_autograd(dt, B, d, H, W, p):
    """ScaledDotProductAttention as TransAttUNet calls it (transatt_unet.py:91-107); with dropout the mask is recovered
    from the engine's own forward (kept elements are those the unmasked product predicts)"""
    g = torch.Generator().manual_seed(d + W)
    T = d ** 0.5
    x = (0.5 * torch.randn(B, d, H, W, generator=g)).to(dt).double().requires_grad_(True)
    eng = Engine(dt, torch.device(DEV), True, True)
    xa = act_from_nchw(x.detach().float().to(DEV), dt)
    torch.manual_seed(5)
    out = eng.channel_attention(xa, T, p, eng.new_act(B, H, W, d))
    qf = x.view(B, d, -1)
    attn = torch.softmax(torch.matmul(qf / T, qf.transpose(1, 2)), dim=-1)
    if p > 0:
        torch.manual_seed(5)
        keep = (torch.rand((B, d, d), device=DEV) >= p).double().cpu() / (1 - p)      # the draw the engine made
        attn = attn * keep
    ref = torch.matmul(attn, qf).view(B, d, H, W)
    dy = torch.randn(ref.shape, generator=g).to(dt).double()
    ref.backward(dy)
    ftol, gtol = (2e-5, 1e-4) if dt == torch.float32 else (2e-2, 4e-2)
    assert relerr(out.dense().cpu(), ref.detach()) < ftol
    out.add_grad(act_from_nchw(dy.float().to(DEV), dt))
    eng.backward_range(None, len(eng.tape), 0)
    e = ((xa.grads[0].dense().cpu().double() - x.grad).norm() / x.grad.norm()).item()
    assert e < gtol, e


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_learned_row_column_embedding_add(dt):
    g = torch.Generator().manual_seed(2)
    B, F_, H, W = 3, 32, 5, 7
    row_w = nn.Parameter(torch.rand(12, F_, generator=g))
    col_w = nn.Parameter(torch.rand(12, F_, generator=g))
    x = torch.randn(B, 2 * F_, H, W, generator=g).to(dt).float().requires_grad_(True)
    pos = torch.cat([col_w[:W].unsqueeze(0).expand(H, W, -1), row_w[:H].unsqueeze(1).expand(H, W, -1)], dim=-1)
    ref = x + pos.permute(2, 0, 1).unsqueeze(0)
    dy = torch.randn(ref.shape, generator=g).to(dt).float()
    ref.backward(dy)
    rw, cw = nn.Parameter(row_w.detach().to(DEV)), nn.Parameter(col_w.detach().to(DEV))
    eng = Engine(dt, torch.device(DEV), True, True)
    xa = act_from_nchw(x.detach().to(DEV), dt)
    out = eng.add_row_col_embed(xa, rw, cw)
    assert relerr(out.dense().cpu(), ref.detach()) < (1e-6 if dt == torch.float32 else 6e-3)
    out.add_grad(act_from_nchw(dy.to(DEV), dt))
    eng.backward_range(None, len(eng.tape), 0)
    assert relerr(eng.param_grads[rw].cpu(), row_w.grad) < 1e-5
    assert relerr(eng.param_grads[cw].cpu(), col_w.grad) < 1e-5
    assert relerr(xa.grads[0].dense().cpu(), x.grad) < 1e-6


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,P,Ci,Cj,pad", [(3, 256, 256, 512, 0), (2, 1024, 1024, 64, 0), (4, 100, 72, 40, 8), (16, 256, 64, 64, 0),
                                           (1, 4096, 512, 128, 0)])
def test_batched_tn_products(dt, B, P, Ci, Cj, pad):
    """out_b = L_b^T R_b: one launch pair for the batch (bf16: the one-tap LDS-DMA kernel with the problem index on the
    grid; fp32: problem by problem), ragged pixel counts, row strides longer than the rows"""
    g = torch.Generator().manual_seed(P + Ci)
    Lm = torch.randn(B * P, Ci + pad, generator=g).to(dt).to(DEV)
    Rm = torch.randn(B * P, Cj + pad, generator=g).to(dt).to(DEV)
    out = ops.wgrad_batched(Act(Lm, 0, Ci, B, 1, P), Act(Rm, 0, Cj, B, 1, P))
    ref = torch.einsum("bpi,bpj->bij", Lm.view(B, P, -1)[..., :Ci].double(), Rm.view(B, P, -1)[..., :Cj].double())
    assert relerr(out.cpu(), ref.cpu()) < (2e-5 if dt == torch.float32 else 1e-5)      # fp32 accumulation of exact bf16 products
    out2 = ops.wgrad_batched(Act(Lm, 0, Ci, B, 1, P), Act(Rm, 0, Cj, B, 1, P))
    assert torch.equal(out, out2)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,n,H,C,KV", [(2, 64, 4, 16, 240), (3, 49, 4, 128, 240), (2, 64, 2, 32, 72)])
def test_channel_cross_attention_matches_autograd(dt, B, n, H, C, KV):
    """Attention_org.forward between its Linear layers (uctransnet.py:160-199): scores over the tokens, InstanceNorm2d on
    the (C, KV) planes, softmax over KV, context, mean over the heads"""
    g = torch.Generator().manual_seed(C + KV)
    hh = int(math.isqrt(n))
    Q = torch.randn(B, n, H * C, generator=g).to(dt).double().requires_grad_(True)
    K = torch.randn(B, n, H * KV, generator=g).to(dt).double().requires_grad_(True)
    V = torch.randn(B, n, H * KV, generator=g).to(dt).double().requires_grad_(True)
    Qh = Q.view(B, n, H, C).permute(0, 2, 3, 1)                 # (B, H, C, n)
    Kh = K.view(B, n, H, KV).permute(0, 2, 1, 3)                # (B, H, n, KV)
    Vh = V.view(B, n, H, KV).permute(0, 2, 3, 1)                # (B, H, KV, n)
    scores = torch.matmul(Qh, Kh) / math.sqrt(KV)
    probs = torch.softmax(F.instance_norm(scores), dim=3)
    ref = torch.matmul(probs, Vh).permute(0, 3, 2, 1).mean(dim=3)       # (B, n, C)
    dy = torch.randn(ref.shape, generator=g).to(dt).double()
    ref.backward(dy)

    def tok(t, c):
        return Act(t.detach().reshape(B * n, c).to(dt).to(DEV).contiguous(), 0, c, B, hh, n // hh)

    eng = Engine(dt, torch.device(DEV), True, True)
    Qa, Ka, Va = tok(Q, H * C), tok(K, H * KV), tok(V, H * KV)
    ctx = eng.channel_cross_attention(Qa, Ka, Va, H)
    ftol, gtol = (2e-5, 2e-4) if dt == torch.float32 else (2e-2, 5e-2)
    assert relerr(ctx.buf.view(B, n, C).cpu(), ref.detach()) < ftol
    ctx.add_grad(tok(dy, C))
    eng.backward_range(None, len(eng.tape), 0)
    for a, r, name in ((Qa, Q, "Q"), (Ka, K, "K"), (Va, V, "V")):
        got = a.grads[0].buf.view(B, n, -1).cpu().double()
        e = ((got - r.grad).norm() / r.grad.norm()).item()
        assert e < gtol, (name, e)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_dropout_kernel_uses_torchs_draw(dt):
    """Engine.dropout: the mask comes from torch.rand under the current seed (reproducible), kept elements are scaled by
    1 / (1 - p) in fp32, the gradient gets the same mask"""
    g = torch.Generator().manual_seed(4)
    B, C, H, W, p = 2, 24, 6, 10, 0.3
    x = torch.randn(B, C, H, W, generator=g).to(dt).float()
    eng = Engine(dt, torch.device(DEV), True, True)
    xa = act_from_nchw(x.to(DEV), dt)
    torch.manual_seed(77)
    out = eng.dropout(xa, p)
    torch.manual_seed(77)
    u = torch.rand((B * H * W, C), device=DEV)
    keep = (u >= p).float().cpu().reshape(B, H, W, C).permute(0, 3, 1, 2)
    scale = (torch.tensor(1.0) / (torch.tensor(1.0) - torch.tensor(p))).item()      # fp32 1 / (1 - p), as nn.Dropout scales
    ref = (x * keep * scale).to(dt).float()
    assert torch.equal(out.dense().cpu(), ref)
    assert 0.6 < keep.mean() < 0.8
    dy = torch.randn(B, C, H, W, generator=g).to(dt).float()
    out.add_grad(act_from_nchw(dy.to(DEV), dt))
    eng.backward_range(None, len(eng.tape), 0)
    assert torch.equal(xa.grads[0].dense().cpu(), (dy * keep * scale).to(dt).float())
    ev = Engine(dt, torch.device(DEV), False, False)
    assert ev.dropout(xa, p) is xa


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("bias", [False, True])
def test_linear_heads_matches_separate_linears(dt, bias):
    """Engine.linear_heads: the per-head Linear layers of Attention_org (uctransnet.py:104-116) as one stacked product each for
    forward, input gradient and weight gradient; an earlier gradient on the input is added in the product's epilogue"""
    g = torch.Generator().manual_seed(13)
    B, H_, W_, Ci, Co, Hn = 2, 8, 8, 40, 24, 4
    lins = [nn.Linear(Ci, Co, bias=bias) for _ in range(Hn)]
    x = torch.randn(B, Ci, H_, W_, generator=g).to(dt).float()
    xr = x.double().clone().requires_grad_(True)
    tok = xr.flatten(2).transpose(1, 2)
    ref = torch.cat([F.linear(tok, l.weight.to(dt).double(), l.bias.double() if bias else None) for l in lins], -1)   # (B, n, Hn Co)
    dy = torch.randn(ref.shape, generator=g).to(dt).double()
    pre = torch.randn(B, Ci, H_, W_, generator=g).to(dt).float()                  # a gradient x has already collected
    (ref * dy).sum().backward()
    want_w = [torch.autograd.grad((F.linear(tok.detach(), w_, None) * dy[..., h * Co:(h + 1) * Co]).sum(), w_)[0]
              for h, w_ in enumerate(l.weight.to(dt).double().requires_grad_(True) for l in lins)]
    lins_d = [nn.Linear(Ci, Co, bias=bias).to(DEV) for _ in range(Hn)]
    for a, b_ in zip(lins_d, lins):
        a.load_state_dict(b_.state_dict())
    eng = Engine(dt, torch.device(DEV), True, True)
    xa = act_from_nchw(x.to(DEV), dt)
    y = eng.linear_heads(xa, lins_d)
    ftol, gtol = (2e-5, 1e-4) if dt == torch.float32 else (2e-2, 3e-2)
    got = y.buf.view(B, H_ * W_, Hn * Co).cpu()
    assert relerr(got, ref.detach()) < ftol
    xa.add_grad(act_from_nchw(pre.to(DEV), dt))
    y.add_grad(Act(dy.reshape(B * H_ * W_, Hn * Co).to(dt).to(DEV).contiguous(), 0, Hn * Co, B, H_, W_))
    eng.backward_range(None, len(eng.tape), 0)
    assert len(xa.grads) == 1
    gx = xa.grads[0].dense().cpu().double()
    assert ((gx - (xr.grad + pre.double())).norm() / (xr.grad + pre.double()).norm()).item() < gtol
    for h, l in enumerate(lins_d):
        e = ((eng.param_grads[l.weight].cpu().double() - want_w[h]).norm() / want_w[h].norm()).item()
        assert e < gtol, (h, e)
        if bias:
            wb = dy[..., h * Co:(h + 1) * Co].sum((0, 1))
            assert ((eng.param_grads[l.bias].cpu().double() - wb).norm() / wb.norm()).item() < gtol
