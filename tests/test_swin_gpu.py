"""GPU: Swin-UNet V2 pieces (patch extraction, LayerNorm with folded permutations, window attention
core, 96-channel head) and the whole model through the C ABI, against torch on CPU, the CPU oracle
and the reference's golden vectors (tests/golden/swin_unet_v2_*)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import unet_zoo_amd
from oracle import torch_ref
from unet_zoo_amd import _lib as L
from unet_zoo_amd import ops
from unet_zoo_amd.ops import Act, act_from_nchw

DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16]


def rnd(dt, t):
    return t.to(dt).float()


def relerr(a, b):
    return ((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30)).item()


def tokens_to_act(t, dt):
    """(B, H, W, C) -> Act"""
    B, H, W, C = t.shape
    return Act(t.reshape(B * H * W, C).to(dt).to(DEV).contiguous(), 0, C, B, H, W)


def act_to_tokens(a: Act):
    return a.buf[:, a.off:a.off + a.C].float().cpu().reshape(a.N, a.H, a.W, a.C)


@pytest.mark.parametrize("dt", DTYPES)
def test_patchify_matches_strided_conv(dt):
    g = torch.Generator().manual_seed(41)
    x = torch.randn(2, 3, 16, 24, generator=g)
    w = torch.randn(96, 3, 4, 4, generator=g) * 0.1
    p = ops.patchify(x.to(DEV), 4, 64, dt)
    assert (p.N, p.H, p.W, p.C) == (2, 4, 6, 64)
    # K order (kh*4 + kw)*C + c, zero padded: contract with the weight in the same order
    wk = w.permute(0, 2, 3, 1).reshape(96, 48)
    got = (p.buf.float().cpu()[:, :48] @ rnd(dt, wk).t()).reshape(2, 4, 6, 96).permute(0, 3, 1, 2)
    ref = F.conv2d(rnd(dt, x), rnd(dt, w), stride=4)
    assert relerr(got, ref) < 1e-5
    assert float(p.buf[:, 48:].abs().max()) == 0.0


def _ln_ref(x, gamma, beta):
    return F.layer_norm(x, (x.shape[-1],), gamma, beta, 1e-5)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("C", [96, 384, 1536])
def test_layernorm_plain_with_residual_and_drop_scale(dt, C):
    g = torch.Generator().manual_seed(42)
    B, H, W = 2, 3, 5
    x = rnd(dt, torch.randn(B, H, W, C, generator=g) * 2 + 0.5).requires_grad_(True)
    res = rnd(dt, torch.randn(B, H, W, C, generator=g)).requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.randn(C, generator=g) * 0.1).requires_grad_(True)
    sb = torch.tensor([0.0, 1.0 / 0.9])
    dy = rnd(dt, torch.randn(B, H, W, C, generator=g))
    ref = res + _ln_ref(x, gamma, beta) * sb.view(B, 1, 1, 1)
    ref.backward(dy)
    xa, ra = tokens_to_act(x.detach(), dt), tokens_to_act(res.detach(), dt)
    out = ops.new_act(B, H, W, C, dt, DEV)
    gd, bd, sbd = gamma.detach().to(DEV), beta.detach().to(DEV), sb.to(DEV)
    stats = ops.layernorm_fwd(xa, gd, bd, out, res=ra, image_scale=sbd)
    tol = 2e-6 if dt == torch.float32 else 1e-2
    assert relerr(act_to_tokens(out), ref.detach()) < tol
    dx = ops.new_act(B, H, W, C, dt, DEV)
    dgam, dbet = ops.layernorm_bwd(xa, gd, stats, tokens_to_act(dy, dt), dx, image_scale=sbd)
    assert relerr(act_to_tokens(dx), x.grad) < (1e-5 if dt == torch.float32 else 1e-2)
    assert relerr(dgam.cpu(), gamma.grad) < 1e-4 and relerr(dbet.cpu(), beta.grad) < 1e-4


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("K,bias,expand", [(1, False, True), (3, True, True), (4, True, False), (2, False, False)])
def test_layernorm_with_1x1_head_fused(dt, K, bias, expand):
    """uz_ln_head_fwd / bwd: conv1x1(LayerNorm(x)) to NCHW logits without the normalised tensor, with the 4x4
    expand addressing of FinalPatchExpand_X4 (swin_unet_v2.py:381-386) and plain tokens"""
    g = torch.Generator().manual_seed(61)
    B, H, W, C, r = 2, 3, 5, 96, 4
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.randn(C, generator=g) * 0.1).requires_grad_(True)
    w = (torch.randn(K, C, generator=g) * 0.2).requires_grad_(True)
    b = torch.randn(K, generator=g).requires_grad_(True) if bias else None
    if expand:
        x = rnd(dt, torch.randn(B, H, W, r * r * C, generator=g) * 2 + 0.5).requires_grad_(True)
        tok = x.view(B, H, W, r, r, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H * r, W * r, C)   # 'b h w (p1 p2 c) -> b (h p1) (w p2) c'
        mode, Ho, Wo = L.LN_EXPAND, H * r, W * r
    else:
        x = rnd(dt, torch.randn(B, H, W, C, generator=g) * 2 + 0.5).requires_grad_(True)
        tok, mode, Ho, Wo, r = x, L.LN_PLAIN, H, W, 1
    ref = F.conv2d(_ln_ref(tok, gamma, beta).permute(0, 3, 1, 2), w.view(K, C, 1, 1), b)
    dlog = torch.randn(ref.shape, generator=g)
    ref.backward(dlog)
    xa = tokens_to_act(x.detach(), dt)
    gd, bd, wd = gamma.detach().to(DEV), beta.detach().to(DEV), w.detach().to(DEV)
    bb = b.detach().to(DEV) if bias else None
    assert ops.ln_head_supported(C, K, dt)
    logits, stats = ops.ln_head_fwd(xa, gd, bd, wd, bb, B, Ho, Wo, C, mode=mode, r=r)
    assert relerr(logits.cpu(), ref.detach()) < (2e-6 if dt == torch.float32 else 1e-5)  # x is exact in both
    dx = ops.new_act(xa.N, xa.H, xa.W, xa.C, dt, DEV)
    dgam, dbet, dw, db = ops.ln_head_bwd(xa, gd, bd, wd, stats, dlog.to(DEV), dx, mode=mode, r=r)
    assert relerr(act_to_tokens(dx), x.grad) < (1e-5 if dt == torch.float32 else 1e-2)
    assert relerr(dgam.cpu(), gamma.grad) < 1e-4 and relerr(dbet.cpu(), beta.grad) < 1e-4
    assert relerr(dw.cpu(), w.grad) < 1e-4
    if bias:
        assert relerr(db.cpu(), b.grad) < 1e-5
    assert not ops.ln_head_supported(C, 5, dt) and not ops.ln_head_supported(3072, 1, dt)


@pytest.mark.parametrize("dt", DTYPES)
def test_layernorm_patch_merging_addressing(dt):
    """LayerNorm(4C) over cat([x0, x1, x2, x3], -1) of PatchMerging without materialising the concat"""
    g = torch.Generator().manual_seed(43)
    B, H, W, C = 2, 4, 6, 96
    x = rnd(dt, torch.randn(B, H, W, C, generator=g)).requires_grad_(True)
    gamma = (torch.rand(4 * C, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.randn(4 * C, generator=g) * 0.1).requires_grad_(True)
    cat = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
    ref = _ln_ref(cat, gamma, beta)
    dy = rnd(dt, torch.randn(ref.shape, generator=g))
    ref.backward(dy)
    xa = tokens_to_act(x.detach(), dt)
    out = ops.new_act(B, H // 2, W // 2, 4 * C, dt, DEV)
    gd, bd = gamma.detach().to(DEV), beta.detach().to(DEV)
    stats = ops.layernorm_fwd(xa, gd, bd, out, mode=L.LN_MERGE)
    tol = 2e-6 if dt == torch.float32 else 1e-2
    assert relerr(act_to_tokens(out), ref.detach()) < tol
    dx = ops.new_act(B, H, W, C, dt, DEV)
    dx.buf.fill_(float("nan"))                      # every input element must be written exactly once
    dgam, dbet = ops.layernorm_bwd(xa, gd, stats, tokens_to_act(dy, dt), dx, mode=L.LN_MERGE)
    assert relerr(act_to_tokens(dx), x.grad) < (1e-5 if dt == torch.float32 else 1e-2)
    assert relerr(dgam.cpu(), gamma.grad) < 1e-4 and relerr(dbet.cpu(), beta.grad) < 1e-4


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("r,c", [(2, 48), (2, 192), (4, 96)])
def test_layernorm_patch_expand_addressing(dt, r, c):
    """'b h w (p1 p2 c) -> b (h p1) (w p2) c' then LayerNorm(c), as one kernel"""
    g = torch.Generator().manual_seed(44)
    B, H, W = 2, 3, 4
    x = rnd(dt, torch.randn(B, H, W, r * r * c, generator=g)).requires_grad_(True)
    gamma = (torch.rand(c, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.randn(c, generator=g) * 0.1).requires_grad_(True)
    sh = x.view(B, H, W, r, r, c).permute(0, 1, 3, 2, 4, 5).reshape(B, H * r, W * r, c)
    ref = _ln_ref(sh, gamma, beta)
    dy = rnd(dt, torch.randn(ref.shape, generator=g))
    ref.backward(dy)
    xa = tokens_to_act(x.detach(), dt)
    out = ops.new_act(B, H * r, W * r, c, dt, DEV)
    gd, bd = gamma.detach().to(DEV), beta.detach().to(DEV)
    stats = ops.layernorm_fwd(xa, gd, bd, out, mode=L.LN_EXPAND, r=r)
    assert relerr(act_to_tokens(out), ref.detach()) < (2e-6 if dt == torch.float32 else 1e-2)
    dx = ops.new_act(B, H, W, r * r * c, dt, DEV)
    dx.buf.fill_(float("nan"))
    dgam, dbet = ops.layernorm_bwd(xa, gd, stats, tokens_to_act(dy, dt), dx, mode=L.LN_EXPAND, r=r)
    assert relerr(act_to_tokens(dx), x.grad) < (1e-5 if dt == torch.float32 else 1e-2)
    assert relerr(dgam.cpu(), gamma.grad) < 1e-4 and relerr(dbet.cpu(), beta.grad) < 1e-4


def _attention_core_ref(qkv, tau, bias, heads, ws, shift):
    """the reference's roll -> window_partition -> cosine attention -> window_reverse -> roll back
    (swin_unet_v2.py:127-159, 246-262) on a (B, H, W, 3C) qkv tensor, without the qkv / proj Linears"""
    B, H, W, C3 = qkv.shape
    C = C3 // 3
    d = C // heads
    xs = torch.roll(qkv, shifts=(-shift, -shift), dims=(1, 2)) if shift > 0 else qkv
    xw = xs.view(B, H // ws, ws, W // ws, ws, C3).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C3)
    B_, N, _ = xw.shape
    t = xw.reshape(B_, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = t[0] * d ** -0.5, t[1], t[2]
    attn = torch.einsum("bhqd,bhkd->bhqk", q, k) / torch.maximum(
        q.norm(dim=-1, keepdim=True) * k.norm(dim=-1, keepdim=True).transpose(-2, -1), torch.tensor(1e-6))
    attn = attn / torch.clip(tau.unsqueeze(0)[:, :, :N, :N], min=0.01) + bias.unsqueeze(0)
    if shift > 0:
        mask = torch_ref.swin_attention_mask(H, W, ws, shift)
        nW = mask.shape[0]
        attn = (attn.view(B_ // nW, nW, heads, N, N) + mask.unsqueeze(1).unsqueeze(0)).view(-1, heads, N, N)
    o = (attn.softmax(-1) @ v).transpose(1, 2).reshape(B_, N, C)
    o = o.view(B, H // ws, W // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, C)
    return torch.roll(o, shifts=(shift, shift), dims=(1, 2)) if shift > 0 else o


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,H,W,heads,ws,shift,Nt", [(2, 16, 16, 3, 8, 0, 64), (2, 16, 16, 3, 8, 4, 64),
                                                     (1, 14, 14, 6, 7, 3, 49), (2, 8, 8, 12, 4, 2, 16),
                                                     (3, 2, 2, 24, 2, 0, 49), (1, 8, 16, 3, 4, 2, 64)])
def test_window_attention_core_forward_backward(dt, B, H, W, heads, ws, shift, Nt):
    g = torch.Generator().manual_seed(45)
    C, N = heads * 32, ws * ws
    qkv = rnd(dt, torch.randn(B, H, W, 3 * C, generator=g)).requires_grad_(True)
    tau = (torch.rand(heads, Nt, Nt, generator=g) * 1.5 + 0.005)
    tau[:, 0, 1] = 0.002                                      # below the 0.01 clip: zero tau gradient there
    tau.requires_grad_(True)
    bias = (torch.randn(heads, N, N, generator=g) * 0.5).requires_grad_(True)
    dout = rnd(dt, torch.randn(B, H, W, C, generator=g))
    ref = _attention_core_ref(qkv, tau, bias, heads, ws, shift)
    ref.backward(dout)
    qa = tokens_to_act(qkv.detach(), dt)
    out = ops.new_act(B, H, W, C, dt, DEV)
    td, bd = tau.detach().to(DEV).contiguous(), bias.detach().to(DEV).contiguous()
    lse = ops.winattn_fwd(qa, td, bd, out, heads, ws, shift)
    tol = 1e-5 if dt == torch.float32 else 1e-2
    assert relerr(act_to_tokens(out), ref.detach()) < tol
    dqkv = ops.new_act(B, H, W, 3 * C, dt, DEV)
    dqkv.buf.fill_(float("nan"))
    dbias, dtau = ops.winattn_bwd(qa, td, bd, out, lse, tokens_to_act(dout, dt), dqkv, heads, ws, shift)
    gtol = 2e-4 if dt == torch.float32 else 2e-2
    assert relerr(act_to_tokens(dqkv), qkv.grad) < gtol
    assert relerr(dbias.cpu(), bias.grad) < gtol
    assert relerr(dtau.cpu(), tau.grad[:, :N, :N]) < (gtol if dt == torch.float32 else 5e-2)
    assert float(dtau[:, 0, 1].abs().max()) == 0.0


@pytest.mark.parametrize("dt", DTYPES)
def test_output_head_with_96_channels(dt):
    """nn.Conv2d(96, K, 1, bias=False) (swin_unet_v2.py:667): 12 / 24 sixteen-byte chunks per pixel"""
    g = torch.Generator().manual_seed(46)
    N, C, H, W, K = 2, 96, 8, 12, 2
    x = rnd(dt, torch.randn(N, C, H, W, generator=g)).requires_grad_(True)
    w = (torch.randn(K, C, 1, 1, generator=g) * 0.1).requires_grad_(True)
    dy = torch.randn(N, K, H, W, generator=g)
    ref = F.conv2d(x, w)
    ref.backward(dy)
    xa = act_from_nchw(x.detach().to(DEV), dt)
    wd = w.detach().to(DEV).reshape(K, C).contiguous()
    out = ops.outconv_fwd(xa, wd, torch.zeros(K, device=DEV))
    assert relerr(out.cpu(), ref.detach()) < 2e-6
    dx = ops.new_act(N, H, W, C, dt, DEV)
    dw, db = ops.outconv_bwd(xa, wd, dy.to(DEV), dx)
    assert relerr(dx.dense().cpu(), x.grad) < (2e-6 if dt == torch.float32 else 8e-3)
    assert relerr(dw.cpu(), w.grad.reshape(K, C)) < 1e-5


# ---------------------------------------------------------------------------------------------
# whole model
# ---------------------------------------------------------------------------------------------
def _golden(golden_dir, tag):
    with open(os.path.join(golden_dir, tag + ".json")) as f:
        meta = json.load(f)
    return meta, np.load(os.path.join(golden_dir, tag + ".npz"))


def _swin(img, ws, dpr=0.0, dtype=torch.float32, K=1):
    torch.manual_seed(0)
    m = unet_zoo_amd.create_model("swin_unet_v2", image_size=img, in_channels=3, num_classes=K, window_size=ws,
                                  drop_path_rate=dpr)
    m.run_dtype = dtype
    return m


def test_swin_fp32_step_matches_reference_golden(golden_dir):
    meta, arr = _golden(golden_dir, "swin_unet_v2_b2_64_ws4")
    x, mask = torch_ref.synthetic_batch(2, 3, 64, 64, seed=1)
    m = _swin(64, 4).to(DEV).train()
    logits = m(x.to(DEV))
    loss = F.binary_cross_entropy_with_logits(logits, mask.to(DEV))
    loss.backward()
    ref = torch.from_numpy(arr["train_logits"])
    got = logits.detach().cpu()
    assert (got - ref).abs().max() <= 1e-3 * ref.abs().max()
    assert torch.equal(got > 0, ref > 0)                            # bit-exact masks
    assert abs(loss.item() - meta["loss"]) < 1e-5
    named = dict(m.named_parameters())
    assert {n for n, p in named.items() if p.grad is None} == set(meta["unused_parameters"])
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in named.values() if p.grad is not None)).item()
    assert abs(gn - meta["global_grad_norm"]) < 2e-3 * meta["global_grad_norm"]
    for name, rn in meta["grad_l2"].items():
        g = named[name].grad
        assert abs(g.double().norm().item() - rn) <= 1e-2 * rn + 1e-5 * meta["global_grad_norm"], (name, g.norm().item(), rn)
        idx = torch.from_numpy(arr["gidx/" + name]).to(DEV)
        np.testing.assert_allclose(g.flatten()[idx].cpu().numpy(), arr["gval/" + name], rtol=2e-2,
                                   atol=2e-5 * max(rn, 1e-3), err_msg=name)
    m.eval()
    with torch.no_grad():
        ev = m(x.to(DEV)).cpu()
    evr = torch.from_numpy(arr["eval_logits"])
    assert (ev - evr).abs().max() <= 1e-3 * evr.abs().max()
    assert int((ev > 0).sum()) == meta["eval_positive_pixels"]


@pytest.mark.parametrize("img,ws", [(256, 8), (224, 7)])
def test_swin_fp32_full_size_forward_matches_reference_golden(golden_dir, img, ws):
    meta, arr = _golden(golden_dir, f"swin_unet_v2_b1_{img}_ws{ws}")
    x, _ = torch_ref.synthetic_batch(1, 3, img, img, seed=1)
    m = _swin(img, ws).to(DEV).eval()
    with torch.no_grad():
        out = m(x.to(DEV)).cpu()
    ref = torch.from_numpy(arr["eval_logits_sampled"])
    got = out.flatten()[torch.from_numpy(arr["logit_idx"])]
    assert (got - ref).abs().max() <= 1e-3 * ref.abs().max()
    assert int((out > 0).sum()) == meta["train_positive_pixels"]  # eval == train when drop_path is the identity


def test_swin_bf16_two_classes_against_oracle_and_trains():
    m = _swin(128, 8, dpr=0.0, dtype=torch.bfloat16, K=2)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(DEV).train()
    x, mask = torch_ref.synthetic_batch(2, 3, 128, 128, seed=9)
    mask = mask.expand(-1, 2, -1, -1).contiguous()
    out = m(x.to(DEV))
    loss = F.binary_cross_entropy_with_logits(out, mask.to(DEV))
    loss.backward()
    cfg = torch_ref.swin_config(sd0, 128, window_size=8, drop_path_rate=0.0)
    rl, rloss, rg, _ = torch_ref.train_step_reference("swin_unet_v2", sd0, x, mask, cfg=cfg)
    got = out.detach().float().cpu()
    assert (got - rl).abs().max() <= 0.05 * rl.abs().max() + 0.02
    assert abs(loss.item() - rloss.item()) < 2e-2
    named = dict(m.named_parameters())
    gflat = torch.cat([named[n].grad.flatten().cpu() for n in rg])
    rflat = torch.cat([rg[n].flatten() for n in rg])
    cos = F.cosine_similarity(gflat.double(), rflat.double(), dim=0).item()
    assert cos > 0.98, cos
    # optimisation on the deterministic graph
    opt = torch.optim.AdamW([p for p in m.parameters()], lr=2e-4)
    xs, ms = x.to(DEV), mask.to(DEV)
    losses = []
    for _ in range(8):
        opt.zero_grad()
        l = F.binary_cross_entropy_with_logits(m(xs), ms)
        l.backward()
        opt.step()
        losses.append(l.item())
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    # stochastic depth (the reference's default rate 0.1): train-mode passes differ, eval is deterministic
    m2 = _swin(128, 8, dpr=0.5, dtype=torch.bfloat16).to(DEV).train()
    with torch.no_grad():
        outs = [m2(xs).float() for _ in range(4)]
    assert any(not torch.equal(outs[0], o) for o in outs[1:])
    F.binary_cross_entropy_with_logits(m2(xs), ms[:, :1]).backward()
    assert all(torch.isfinite(p.grad).all() for p in m2.parameters() if p.grad is not None)
    m2.eval()
    with torch.no_grad():
        assert torch.equal(m2(xs), m2(xs))


def test_swin_registry_errors_match_the_reference():
    with pytest.raises(ValueError):
        unet_zoo_amd.create_model("swin_unet_v2")                   # image_size is required (models/__init__.py:160-162)
    m = _swin(64, 4).to(DEV)
    with pytest.raises(AssertionError):
        m(torch.zeros(1, 3, 32, 32, device=DEV))                    # size check of PatchEmbed.forward (:550-552)


@pytest.mark.parametrize("heads,N", [(3, 64), (24, 4), (6, 49)])
def test_continuous_position_bias_mlp(heads, N):
    """cpb = Linear(2,256)-ReLU-Linear(256,heads) on the log-spaced offsets (swin_unet_v2.py:58-72, 121-125)"""
    g = torch.Generator().manual_seed(47)
    R = N * N
    idx = torch.randn(R, 2, generator=g)
    w1 = (torch.randn(256, 2, generator=g)).requires_grad_(True)
    b1 = (torch.randn(256, generator=g) * 0.5).requires_grad_(True)
    w2 = (torch.randn(heads, 256, generator=g) * 0.1).requires_grad_(True)
    b2 = torch.randn(heads, generator=g).requires_grad_(True)
    G = torch.randn(heads, R, generator=g)
    ref = F.linear(F.relu(F.linear(idx, w1, b1)), w2, b2).t()
    ref.backward(G)
    d = [t.detach().to(DEV) for t in (idx, w1, b1, w2, b2)]
    bias = ops.cpb_fwd(*d)
    assert relerr(bias.cpu(), ref.detach()) < 2e-6
    outs = [torch.empty_like(t) for t in d[1:]]
    ops.cpb_bwd(d[0], d[1], d[2], d[3], G.to(DEV).contiguous(), *outs)
    for got, want in zip(outs, (w1.grad, b1.grad, w2.grad, b2.grad)):
        assert relerr(got.cpu(), want) < 1e-5


def test_continuous_position_bias_batched_launch():
    """uz_cpb_fwd_batched / uz_cpb_bwd_batched: every module of a model in one launch each way; mixed head
    counts, table sizes (ragged last row block) and hidden widths against autograd"""
    g = torch.Generator().manual_seed(53)
    mods, refs = [], []
    for heads, N, hidden in [(3, 64, 256), (24, 4, 256), (6, 49, 256), (12, 64, 256), (5, 20, 96), (32, 9, 512)]:
        R = N * N
        idx = torch.randn(R, 2, generator=g)
        w1 = torch.randn(hidden, 2, generator=g).requires_grad_(True)
        b1 = (torch.randn(hidden, generator=g) * 0.5).requires_grad_(True)
        w2 = (torch.randn(heads, hidden, generator=g) * 0.1).requires_grad_(True)
        b2 = torch.randn(heads, generator=g).requires_grad_(True)
        G = torch.randn(heads, R, generator=g)
        ref = F.linear(F.relu(F.linear(idx, w1, b1)), w2, b2).t()
        ref.backward(G)
        refs.append((ref.detach(), w1.grad, b1.grad, w2.grad, b2.grad))
        m = {k: t.detach().to(DEV).contiguous() for k, t in (("idx", idx), ("w1", w1), ("b1", b1), ("w2", w2), ("b2", b2), ("G", G))}
        m["bias"] = torch.full((heads, R), float("nan"), device=DEV)
        for k, t in (("dw1", w1), ("db1", b1), ("dw2", w2), ("db2", b2)):
            m[k] = torch.full(t.shape, float("nan"), device=DEV)
        mods.append(m)
    ops.cpb_fwd_batched(mods)
    ops.cpb_bwd_batched(mods)
    for m, (bias, dw1, db1, dw2, db2) in zip(mods, refs):
        assert relerr(m["bias"].cpu(), bias) < 2e-6
        for k, want in (("dw1", dw1), ("db1", db1), ("dw2", dw2), ("db2", db2)):
            assert relerr(m[k].cpu(), want) < 1e-5, k
    # the same numbers as the per-module entry points (different summation order only)
    one = mods[0]
    single = [torch.empty_like(one[k]) for k in ("dw1", "db1", "dw2", "db2")]
    ops.cpb_bwd(one["idx"], one["w1"], one["b1"], one["w2"], one["G"], *single)
    for k, t in zip(("dw1", "db1", "dw2", "db2"), single):
        assert relerr(one[k], t) < 1e-5


def test_swin_options_ape_qk_scale_against_oracle():
    """constructor options the reference accepts beyond its defaults (swin_unet_v2.py:596-700): ape=True, qk_scale -- fp32
    against the oracle, gradients of the absolute position embedding and tau included"""
    torch.manual_seed(0)
    kw = dict(image_size=64, window_size=4, drop_path_rate=0.0, ape=True, qk_scale=0.2)
    m = unet_zoo_amd.create_model("swin_unet_v2", in_channels=3, num_classes=1, **kw)
    m.run_dtype = torch.float32
    with torch.no_grad():
        m.absolute_pos_embed.normal_(0.0, 0.5, generator=torch.Generator().manual_seed(3))
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    assert "absolute_pos_embed" in sd and tuple(sd["absolute_pos_embed"].shape) == (1, 256, 96)
    m = m.to(DEV).train()
    x, mask = torch_ref.synthetic_batch(2, 3, 64, 64, seed=4)
    cfg = torch_ref.swin_config(sd, 64, window_size=4)
    cfg["qk_scale"] = 0.2
    ref_logits, ref_loss, ref_grads, _ = torch_ref.train_step_reference("swin_unet_v2", sd, x, mask, cfg=cfg)
    logits = m(x.to(DEV))
    loss = F.binary_cross_entropy_with_logits(logits, mask.to(DEV))
    loss.backward()
    assert (logits.detach().cpu() - ref_logits).abs().max() <= 1e-3 * ref_logits.abs().max()
    assert abs(loss.item() - ref_loss.item()) < 1e-5
    named = dict(m.named_parameters())
    assert {n for n, p in named.items() if p.grad is not None} == set(ref_grads)
    for n, g in ref_grads.items():
        want = g.double().norm().item()
        assert abs(named[n].grad.double().norm().item() - want) <= 2e-2 * want + 1e-6, n
    for n in ("absolute_pos_embed", "layers.0.blocks.1.attn.tau", "layers.2.blocks.0.attn.cpb.fc2.weight"):
        assert (named[n].grad.cpu() - ref_grads[n]).abs().max() <= 2e-2 * ref_grads[n].abs().max() + 1e-8, n


def test_swin_without_patch_norm_against_oracle():
    """patch_norm=False (swin_unet_v2.py:555, :619): the patch embedding feeds the first stage and the first skip without a
    LayerNorm; fp32 against the oracle, and the state dict holds no `patch_embed.norm.*`"""
    torch.manual_seed(0)
    m = unet_zoo_amd.create_model("swin_unet_v2", in_channels=3, num_classes=1, image_size=64, window_size=4, drop_path_rate=0.0,
                                  patch_norm=False)
    m.run_dtype = torch.float32
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    assert not any(k.startswith("patch_embed.norm") for k in sd)
    m = m.to(DEV).train()
    x, mask = torch_ref.synthetic_batch(2, 3, 64, 64, seed=6)
    cfg = torch_ref.swin_config(sd, 64, window_size=4)
    ref_logits, ref_loss, ref_grads, _ = torch_ref.train_step_reference("swin_unet_v2", sd, x, mask, cfg=cfg)
    logits = m(x.to(DEV))
    loss = F.binary_cross_entropy_with_logits(logits, mask.to(DEV))
    loss.backward()
    assert (logits.detach().cpu() - ref_logits).abs().max() <= 1e-3 * ref_logits.abs().max()
    assert abs(loss.item() - ref_loss.item()) < 1e-5
    named = dict(m.named_parameters())
    assert {n for n, p in named.items() if p.grad is not None} == set(ref_grads)
    for n, g in ref_grads.items():
        want = g.double().norm().item()
        assert abs(named[n].grad.double().norm().item() - want) <= 2e-2 * want + 1e-6, n


def test_swin_refuses_what_no_kernel_takes():
    """head widths other than 32 and dropout on the attention probabilities have no window-attention kernel; rounds 2-4 ran
    them through torch GEMMs and autograd (a second backend) -- now they are refused, in training, with a message that says
    what IS supported; eval mode (dropout off) still runs"""
    x, _ = torch_ref.synthetic_batch(2, 3, 64, 64, seed=4)
    torch.manual_seed(0)
    m = unet_zoo_amd.create_model("swin_unet_v2", in_channels=3, num_classes=1, image_size=64, window_size=4,
                                  num_heads=[6, 12, 24, 48]).to(DEV)
    with pytest.raises(NotImplementedError, match="head_dim 32"):
        m(x.to(DEV))
    m = unet_zoo_amd.create_model("swin_unet_v2", in_channels=3, num_classes=1, image_size=64, window_size=4,
                                  attn_drop_rate=0.1).to(DEV).train()
    with pytest.raises(NotImplementedError, match="attention dropout"):
        m(x.to(DEV))
    m.eval()
    with torch.no_grad():
        assert torch.isfinite(m(x.to(DEV))).all()


def test_swin_dropout_options_train_and_are_off_in_eval():
    """drop_rate (after the embedding and after the attention projection): masks from torch's generator, reproducible under
    a seed, identity in eval mode, and the step still trains -- also replayed from hipGraphs"""
    kw = dict(image_size=64, window_size=4, drop_rate=0.1, ape=True)
    torch.manual_seed(0)
    m = unet_zoo_amd.create_model("swin_unet_v2", in_channels=3, num_classes=1, **kw).to(DEV)
    x, mask = torch_ref.synthetic_batch(2, 3, 64, 64, seed=4)
    x, mask = x.to(DEV), mask.to(DEV)
    m.train()
    outs = []
    for seed in (1, 1, 2):
        torch.manual_seed(seed)
        with torch.no_grad():
            outs.append(m(x).clone())
    assert torch.equal(outs[0], outs[1]) and not torch.equal(outs[0], outs[2])
    m.eval()
    with torch.no_grad():
        e1, e2 = m(x), m(x)
    assert torch.equal(e1, e2)
    torch.manual_seed(0)
    m0 = unet_zoo_amd.create_model("swin_unet_v2", in_channels=3, num_classes=1, image_size=64, window_size=4, ape=True).to(DEV).eval()
    m0.load_state_dict(m.state_dict())
    with torch.no_grad():
        assert torch.equal(m0(x), e1)                     # eval mode: the dropout options change nothing
    m.train()
    gs = unet_zoo_amd.GraphedStep(m, "bce_dice", lr=1e-3)
    losses = [gs(x, mask).item() for _ in range(10)]
    assert all(np.isfinite(losses)) and min(losses[5:]) < losses[0]


@pytest.mark.parametrize("rows,n,n0", [(16, 196608, 98304), (147, 24576, 24576), (64, 16384, 16380), (33, 20000, 20000)])
def test_row_sums_of_few_rows_of_many_columns(rows, n, n0):
    """uz_sum_rows_f32 on the window-attention d(bias) / d(tau) partial shapes (the 16-byte-per-thread kernel of uz_attn.hip)
    against a double-precision torch sum; the split into two destinations falls inside a thread's four columns once"""
    g = torch.Generator().manual_seed(rows)
    part = torch.randn(rows, n, generator=g).to(DEV)
    out0 = torch.empty(n0, device=DEV)
    out1 = torch.empty(n - n0, device=DEV) if n0 < n else None
    ops.sum_rows_f32(part, rows, out0, out1)
    ref = part.double().sum(0).float()
    assert torch.equal(out0, ref[:n0])
    if out1 is not None:
        assert torch.equal(out1, ref[n0:])
