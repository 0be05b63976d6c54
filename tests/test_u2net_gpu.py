"""GPU: U^2-Net pieces (small-channel / dilated convolutions, bilinear resize, residual + pool
gradient merge, side heads, fuse conv) and the whole model through the C ABI, against
torch.nn.functional on CPU and the reference's golden vectors (tests/golden/u2net_*)."""
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


def tol(dt):
    return 2e-5 if dt == torch.float32 else 2e-2


def rnd(dt, t):
    return t.to(dt).float()


def relerr(a, b):
    return ((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30)).item()


# (N, H, W, Cin, Cout, dilation): the REBNCONV shapes of u2net / u2netp that the UNet never used
CONV_CASES = [(2, 16, 16, 16, 16, 1), (1, 32, 32, 32, 16, 1), (2, 16, 16, 16, 16, 2), (1, 16, 32, 32, 32, 2),
              (1, 8, 8, 64, 64, 4), (1, 16, 16, 128, 64, 8), (1, 64, 64, 32, 64, 1), (1, 8, 8, 96, 32, 2),
              (2, 2, 2, 16, 16, 2), (1, 4, 4, 256, 128, 8)]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("N,H,W,Cin,Cout,dil", CONV_CASES)
def test_rebnconv_shapes_fwd_dgrad_wgrad(dt, N, H, W, Cin, Cout, dil):
    """Conv2d(k3, padding=d, dilation=d) with channel counts below one 128-byte slab (u2net.py:10)"""
    g = torch.Generator().manual_seed(31)
    x = rnd(dt, torch.randn(N, Cin, H, W, generator=g)).requires_grad_(True)
    w = rnd(dt, torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1).requires_grad_(True)
    b = torch.randn(Cout, generator=g)
    dy = rnd(dt, torch.randn(N, Cout, H, W, generator=g))
    ref = F.conv2d(x, w, b, padding=dil, dilation=dil)
    ref.backward(dy)
    xa = act_from_nchw(x.detach().to(DEV), dt)
    y = ops.new_act(N, H, W, Cout, dt, DEV)
    stats = ops.conv_igemm(xa, ops.pack_weights(w.detach().to(DEV), L.PACK_CONV_FWD, dt), b.to(DEV), y,
                           ntaps=9, dil=dil, want_stats=True)
    got = y.dense().cpu()
    assert relerr(got, ref.detach()) < tol(dt)
    # the statistics epilogue sums the stored values
    s = stats.sum(0).cpu()
    np.testing.assert_allclose(s[0].numpy(), got.sum((0, 2, 3)).numpy(), rtol=1e-3, atol=1e-2)
    dya = act_from_nchw(dy.to(DEV), dt)
    dw = ops.wgrad(dya, xa, (Cout, Cin, 3, 3), ntaps=9, dil=dil)
    assert relerr(dw.cpu(), w.grad) < tol(dt)
    dx = ops.new_act(N, H, W, Cin, dt, DEV)
    ops.conv_igemm(dya, ops.pack_weights(w.detach().to(DEV), L.PACK_CONV_DGRAD, dt), None, dx, ntaps=9, dil=dil)
    assert relerr(dx.dense().cpu(), x.grad) < tol(dt)


@pytest.mark.parametrize("dt", DTYPES)
def test_conv_reads_a_narrow_window_of_a_wider_buffer(dt):
    """partial K-slabs must zero-fill, not read the neighbouring channels of a concat buffer"""
    g = torch.Generator().manual_seed(32)
    N, H, W, C = 1, 16, 16, 16
    both = rnd(dt, torch.randn(N, 2 * C, H, W, generator=g))
    both[:, C:] = float("nan")                          # the other half must never be touched
    w = rnd(dt, torch.randn(32, C, 3, 3, generator=g) * 0.1)
    full = act_from_nchw(both.to(DEV), dt)
    y = ops.new_act(N, H, W, 32, dt, DEV)
    ops.conv_igemm(full.window(0, C), ops.pack_weights(w.to(DEV), L.PACK_CONV_FWD, dt), None, y, ntaps=9)
    ref = F.conv2d(both[:, :C], w, padding=1)
    assert relerr(y.dense().cpu(), ref) < tol(dt)
    y2 = ops.new_act(N, H, W, 32, dt, DEV)
    ops.conv_igemm(full.window(0, C), ops.pack_weights(w.to(DEV), L.PACK_CONV_FWD, dt), None, y2, ntaps=9, dil=2)
    assert relerr(y2.dense().cpu(), F.conv2d(both[:, :C], w, padding=2, dilation=2)) < tol(dt)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("N,C,hi,wi,ho,wo", [(2, 64, 8, 8, 16, 16), (1, 16, 2, 2, 4, 4), (1, 32, 16, 32, 32, 64),
                                             (1, 8, 6, 6, 16, 10), (1, 16, 1, 1, 2, 2), (1, 24, 5, 7, 5, 7)])
def test_bilinear_resize_nhwc(dt, N, C, hi, wi, ho, wo):
    g = torch.Generator().manual_seed(33)
    x = rnd(dt, torch.randn(N, C, hi, wi, generator=g)).requires_grad_(True)
    dy = rnd(dt, torch.randn(N, C, ho, wo, generator=g))
    ref = F.interpolate(x, size=(ho, wo), mode="bilinear", align_corners=False)
    ref.backward(dy)
    xa = act_from_nchw(x.detach().to(DEV), dt)
    # write into the left half of a wider buffer, as the model does
    full = ops.new_act(N, ho, wo, 2 * C, dt, DEV)
    full.buf.zero_()
    out = full.window(0, C)
    ops.bilinear_fwd(xa, out)
    assert relerr(out.dense().cpu(), ref.detach()) < (1e-6 if dt == torch.float32 else 8e-3)
    assert float(full.buf[:, C:].abs().max()) == 0.0
    gfull = act_from_nchw(torch.cat([dy, torch.full_like(dy, float("nan"))], 1).to(DEV), dt)
    dx = ops.new_act(N, hi, wi, C, dt, DEV)
    ops.bilinear_bwd(gfull.window(0, C), dx)
    assert relerr(dx.dense().cpu(), x.grad) < (2e-6 if dt == torch.float32 else 8e-3)


@pytest.mark.parametrize("n,hi,wi,ho,wo", [(3, 16, 16, 64, 64), (2, 2, 2, 64, 64), (2, 32, 32, 64, 64), (1, 3, 5, 24, 17)])
def test_bilinear_resize_logit_planes(n, hi, wi, ho, wo):
    g = torch.Generator().manual_seed(34)
    x = torch.randn(n, 1, hi, wi, generator=g, requires_grad=True)
    dy = torch.randn(n, 1, ho, wo, generator=g)
    ref = F.interpolate(x, size=(ho, wo), mode="bilinear", align_corners=False)
    ref.backward(dy)
    xd = x.detach().to(DEV).contiguous()
    cat = torch.zeros(n, 3, ho, wo, device=DEV)          # channel 1 of a 3-channel NCHW buffer
    ops.bilinear_planes(xd.data_ptr(), hi * wi, hi, wi, cat.data_ptr() + ho * wo * 4, 3 * ho * wo, ho, wo, n)
    assert relerr(cat[:, 1:2].cpu(), ref.detach()) < 1e-6
    assert float(cat[:, 0].abs().max()) == 0.0 and float(cat[:, 2].abs().max()) == 0.0
    gcat = torch.zeros(n, 3, ho, wo, device=DEV)
    gcat[:, 1:2] = dy.to(DEV)
    dx = torch.empty(n, 1, hi, wi, device=DEV)
    ops.bilinear_planes(gcat.data_ptr() + ho * wo * 4, 3 * ho * wo, hi, wi, dx.data_ptr(), hi * wi, ho, wo, n,
                        backward=True)
    assert relerr(dx.cpu(), x.grad) < 2e-6


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("pool", [False, True])
def test_bn_relu_residual_and_gradient_merge(dt, pool):
    """act = relu(bn(y)) + res; pooled = maxpool(act); total gradient = g0 + g1 + unpool(gp)"""
    g = torch.Generator().manual_seed(35)
    N, C, H, W = 2, 32, 8, 16
    y = rnd(dt, torch.randn(N, C, H, W, generator=g))
    res = rnd(dt, torch.randn(N, C, H, W, generator=g))
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    ref = (F.relu(y * sc.view(1, C, 1, 1) + sh.view(1, C, 1, 1)) + res)
    ya, ra = act_from_nchw(y.to(DEV), dt), act_from_nchw(res.to(DEV), dt)
    act = ops.new_act(N, H, W, C, dt, DEV)
    pooled = ops.new_act(N, H // 2, W // 2, C, dt, DEV) if pool else None
    ops.bn_relu_apply(ya, sc.to(DEV), sh.to(DEV), act, pooled, ra)
    got = act.dense().cpu()
    assert relerr(got, ref) < (1e-6 if dt == torch.float32 else 8e-3)
    g0 = rnd(dt, torch.randn(N, C, H, W, generator=g))
    g1 = rnd(dt, torch.randn(N, C, H, W, generator=g))
    tot = ops.new_act(N, H, W, C, dt, DEV)
    if pool:
        assert torch.equal(pooled.dense().cpu(), F.max_pool2d(got, 2))
        gp = rnd(dt, torch.randn(N, C, H // 2, W // 2, generator=g))
        a = got.clone().requires_grad_(True)            # route through the STORED activation values
        (F.max_pool2d(a, 2) * gp).sum().backward()
        want = g0 + g1 + a.grad
        ops.pool_grad_combine(act, act_from_nchw(g0.to(DEV), dt), act_from_nchw(g1.to(DEV), dt),
                              act_from_nchw(gp.to(DEV), dt), tot)
        assert relerr(tot.dense().cpu(), want) < (1e-6 if dt == torch.float32 else 8e-3)
        ops.pool_grad_combine(act, None, None, act_from_nchw(gp.to(DEV), dt), tot)
        assert relerr(tot.dense().cpu(), a.grad) < (1e-6 if dt == torch.float32 else 8e-3)
    else:
        ops.pool_grad_combine(act, act_from_nchw(g0.to(DEV), dt), act_from_nchw(g1.to(DEV), dt), None, tot)
        assert relerr(tot.dense().cpu(), g0 + g1) < (1e-6 if dt == torch.float32 else 8e-3)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("N,C,H,W", [(2, 64, 16, 16), (1, 512, 4, 8), (1, 128, 2, 2), (2, 256, 8, 8)])
def test_side_head_conv3x3_to_one_channel(dt, N, C, H, W):
    g = torch.Generator().manual_seed(36)
    x = rnd(dt, torch.randn(N, C, H, W, generator=g)).requires_grad_(True)
    w = (torch.randn(1, C, 3, 3, generator=g) * 0.05).requires_grad_(True)
    b = torch.randn(1, generator=g, requires_grad=True)
    dy = torch.randn(N, 1, H, W, generator=g)
    ref = F.conv2d(x, w, b, padding=1)
    ref.backward(dy)
    xa = act_from_nchw(x.detach().to(DEV), dt)
    wd, bd = w.detach().to(DEV).contiguous(), b.detach().to(DEV)
    taps = torch.empty(N * 9 * H * W, device=DEV)
    out = torch.empty(N, 1, H, W, device=DEV)
    ops.sideconv_fwd(xa, wd.data_ptr(), bd.data_ptr(), taps, out.data_ptr(), H * W)
    assert relerr(out.cpu(), ref.detach()) < 2e-6
    dyd = dy.to(DEV).contiguous()
    dx = ops.new_act(N, H, W, C, dt, DEV)
    dw, db = torch.empty(1, C, 3, 3, device=DEV), torch.empty(1, device=DEV)
    ops.sideconv_bwd(xa, wd.data_ptr(), dyd.data_ptr(), H * W, dx, dw.data_ptr(), db.data_ptr())
    assert relerr(dx.dense().cpu(), x.grad) < (2e-6 if dt == torch.float32 else 8e-3)
    assert relerr(dw.cpu(), w.grad) < 1e-5
    assert relerr(db.cpu(), b.grad) < 1e-5


@pytest.mark.parametrize("K", [1, 2])
def test_fuse_conv_and_side_gradient_merge(K):
    g = torch.Generator().manual_seed(37)
    N, H, W, S = 2, 16, 8, 6
    d = torch.randn(N, S * K, H, W, generator=g, requires_grad=True)
    w = torch.randn(K, S * K, 1, 1, generator=g, requires_grad=True)
    b = torch.randn(K, generator=g, requires_grad=True)
    gm = torch.randn(N, K, H, W, generator=g)
    extras = [torch.randn(N, K, H, W, generator=g) if s != 2 else None for s in range(S)]
    ref = F.conv2d(d, w, b)
    tot = (ref * gm).sum()
    for s, e in enumerate(extras):
        if e is not None:
            tot = tot + (d[:, s * K:(s + 1) * K] * e).sum()
    tot.backward()
    dd, wd, bd = d.detach().to(DEV), w.detach().to(DEV).reshape(K, S * K).contiguous(), b.detach().to(DEV)
    out = ops.fuse1x1_fwd(dd, wd, bd)
    assert relerr(out.cpu(), ref.detach()) < 2e-6
    dw, db = torch.empty(K, S * K, device=DEV), torch.empty(K, device=DEV)
    dcat = ops.fuse1x1_bwd(dd, wd, gm.to(DEV), [e.to(DEV) if e is not None else None for e in extras], dw, db)
    assert relerr(dcat.cpu(), d.grad) < 2e-6
    assert relerr(dw.cpu(), w.grad.reshape(K, S * K)) < 1e-5
    assert relerr(db.cpu(), b.grad) < 1e-5


# ---------------------------------------------------------------------------------------------
# whole model
# ---------------------------------------------------------------------------------------------
def _golden(golden_dir, tag):
    with open(os.path.join(golden_dir, tag + ".json")) as f:
        meta = json.load(f)
    return meta, np.load(os.path.join(golden_dir, tag + ".npz"))


def _loss(outs, mask):
    return sum(F.binary_cross_entropy_with_logits(v, mask) for v in outs.values())


def test_u2net_fp32_step_matches_reference_golden(golden_dir):
    meta, arr = _golden(golden_dir, "u2net_b2_64")
    x, mask = torch_ref.synthetic_batch(2, 3, 64, 64, seed=1)
    torch.manual_seed(0)
    m = unet_zoo_amd.create_model("u2net", in_channels=3, num_classes=1)
    m.run_dtype = torch.float32
    m = m.to(DEV).train()
    outs = m(x.to(DEV))
    assert list(outs.keys()) == meta["keys"]
    loss = _loss(outs, mask.to(DEV))
    loss.backward()
    for k, v in outs.items():
        ref = torch.from_numpy(arr["train/" + k])
        got = v.detach().cpu()
        assert got.shape == ref.shape
        assert (got - ref).abs().max() <= 1e-3 * ref.abs().max(), k
        # masks: identical wherever the reference logit is not within rounding noise of zero.  The
        # reference's own fp32 forward sits 5e-4 (absolute) from its fp64 forward on this input
        # (8-sample BatchNorms at the 2x2 levels amplify rounding), so a logit that close to zero
        # has no defined sign; see DESIGN.md "u2net parity".
        sure = ref.abs() > 1e-3 * ref.abs().max()
        assert torch.equal((got > 0)[sure], (ref > 0)[sure]), k
        assert int(((got > 0) != (ref > 0)).sum()) <= 4, k
    assert abs(loss.item() - meta["loss"]) < 2e-5
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters())).item()
    # the reference's own fp32 gradient has 0.83 % global-norm error and cosine 0.9954 against its fp64
    # gradient on this input; any other fp32 summation order lands in the same band
    assert abs(gn - meta["global_grad_norm"]) < 3e-2 * meta["global_grad_norm"]
    gs, rs = [], []
    for name, p in m.named_parameters():
        if not name.endswith("conv_s1.bias"):
            gs.append(p.grad.flatten()[torch.from_numpy(arr["gidx/" + name]).to(DEV)].cpu())
            rs.append(torch.from_numpy(arr["gval/" + name]))
    cos = F.cosine_similarity(torch.cat(gs).double(), torch.cat(rs).double(), dim=0).item()
    assert cos > 0.99, cos
    worst = 0.0
    for name, p in m.named_parameters():
        rn = meta["grad_l2"][name]
        got_n = p.grad.double().norm().item()
        if name.endswith("conv_s1.bias"):
            assert got_n <= 1e-4 * meta["global_grad_norm"], name   # biases in front of a BatchNorm
            continue
        # ill-conditioned on this input: the reference's OWN fp32 gradients differ from its fp64
        # gradients by up to 6 % in norm per tensor (8-sample BatchNorms at the 2x2 levels, ReLU-mask
        # and pool-argmax flips); the whole-model norm above is the tight check
        assert abs(got_n - rn) <= 0.12 * rn + 2e-5 * meta["global_grad_norm"], (name, got_n, rn)
        worst = max(worst, abs(got_n - rn) / (rn + 1e-12))
    sd = m.state_dict()
    for k in meta["bn_keys"]:
        np.testing.assert_allclose(sd[k + ".running_mean"].cpu().numpy(), arr["rm/" + k], rtol=2e-3, atol=1e-5)
        np.testing.assert_allclose(sd[k + ".running_var"].cpu().numpy(), arr["rv/" + k], rtol=2e-3, atol=1e-5)
    m.eval()
    with torch.no_grad():
        ev = m(x.to(DEV))
    for k, v in ev.items():
        evr = torch.from_numpy(arr["eval/" + k])
        assert (v.cpu() - evr).abs().max() <= 1e-3 * evr.abs().max(), k
        assert abs(int((v > 0).sum()) - meta["eval_positive_pixels"][k]) <= 4


@pytest.mark.parametrize("name,K,H,W", [("u2netp", 2, 96, 64), ("u2netp", 1, 64, 64), ("u2net", 1, 64, 96),
                                        ("u2net", 1, 32, 64)])
def test_u2net_family_fp32_against_oracle(name, K, H, W):
    """other member / two classes / non-square: compare with the CPU oracle on the same weights.

    The bound of every head is max(1e-3, 4 x the reference graph's OWN fp32-vs-fp64 spread on this input), computed
    here.  At 2 x 32 x 64 the innermost RSU maps are 1 x 2 pixels, their BatchNorms see four samples, and the reference
    evaluated in fp32 differs from itself in fp64 by 3.4 % on `main` (16 % on the 1 x 2 layers; 1.4e-4 at 64 x 96): the
    5 % seen there in round 1 was the conditioning of the network, not a kernel fault on small maps."""
    torch.manual_seed(3)
    m = unet_zoo_amd.create_model(name, in_channels=3, num_classes=K)
    m.run_dtype = torch.float32
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(DEV).train()
    x, mask = torch_ref.synthetic_batch(2, 3, H, W, seed=5)
    mask = mask.expand(-1, K, -1, -1).contiguous()
    outs = m(x.to(DEV))
    loss = _loss(outs, mask.to(DEV))
    loss.backward()
    st = torch_ref.clone_state(sd0, requires_grad=True)
    ref = torch_ref.u2net_forward(st, x, True)
    rloss = torch_ref.model_loss(ref, mask)
    names = [k for k, v in st.items() if v.requires_grad]
    rg = dict(zip(names, torch.autograd.grad(rloss, [st[k] for k in names])))
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
    with torch.no_grad():
        r64 = torch_ref.u2net_forward(torch_ref.clone_state(sd64), x.double(), True)
    spread = {k: ((ref[k].detach().double() - r64[k]).abs().max() / r64[k].abs().max()).item() for k in ref}
    ill = max(spread.values()) > 2.5e-4
    for k in ref:
        got, want = outs[k].detach().cpu(), ref[k].detach()
        assert (got - want).abs().max() <= max(1e-3, 4 * spread[k]) * want.abs().max(), (k, spread[k])
    assert abs(loss.item() - rloss.item()) < (2e-5 if not ill else 40 * max(spread.values()) * rloss.item())
    if ill:
        assert (H, W) == (32, 64) and spread["main"] > 1e-2      # the documented case, nothing else
        return
    keep = [n for n, _ in m.named_parameters() if not n.endswith("conv_s1.bias")]
    gflat = torch.cat([dict(m.named_parameters())[n].grad.flatten().cpu() for n in keep])
    rflat = torch.cat([rg[n].flatten() for n in keep])
    cos = F.cosine_similarity(gflat.double(), rflat.double(), dim=0).item()
    # the reference's own fp32-vs-fp64 gradients on such inputs: cosine 0.995, norm 0.8 % apart
    # (ill-conditioned, see the golden test above)
    assert cos > 0.995, cos
    assert abs(gflat.double().norm().item() / rflat.double().norm().item() - 1) < 3e-2


@pytest.mark.parametrize("name", ["u2net", "u2netp"])
def test_u2net_bf16_first_stage_tracks_storage_rounded_oracle(name):
    """bf16 run: every REBNCONV output of stage1 (14 convolutions: im2col input, 64/32/16-channel and
    dilated layers, fused pools, bilinear resizes into concat halves, the residual tail) against the
    oracle that rounds to bf16 at the same storage points.  Deeper stages are not comparable in bf16:
    this random-init network amplifies a rounding-level perturbation 10^4 x by the last layer (measured
    in fp32: 9e-8 -> 1e-3, tests/debug/u2_layer_diff.py), so two correct bf16 evaluations decorrelate."""
    from unet_zoo_amd.engine import Engine
    torch.manual_seed(3)
    m = unet_zoo_amd.create_model(name)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(DEV).train()
    x, mask = torch_ref.synthetic_batch(2, 3, 256, 256, seed=5)
    names = {id(mod): n for n, mod in m.named_modules()}
    got, want = {}, {}
    eng_fn, ora_fn = Engine.conv_bn_relu, torch_ref.conv_bn_relu

    def rec(self, x_, conv, bn, **kw):
        act, pooled = eng_fn(self, x_, conv, bn, **kw)
        if names[id(conv)].startswith("stage1."):
            got[names[id(conv)]] = act.dense().cpu()
        return act, pooled

    def rec2(x_, sd, conv, bn, training, dilation=1, residual=None):
        y = ora_fn(x_, sd, conv, bn, training, dilation, residual)
        if conv.startswith("stage1."):
            want[conv] = y.detach()
        return y

    Engine.conv_bn_relu, torch_ref.conv_bn_relu = rec, rec2
    torch_ref.set_storage_rounding(torch.bfloat16)
    try:
        with torch.no_grad():
            outs = m(x.to(DEV))
            ref = torch_ref.u2net_forward(torch_ref.clone_state(sd0), x, True)
    finally:
        Engine.conv_bn_relu, torch_ref.conv_bn_relu = eng_fn, ora_fn
        torch_ref.set_storage_rounding(None)
    assert len(want) == 14 and set(got) == set(want)
    for k in want:
        rms = ((got[k] - want[k]).pow(2).mean().sqrt() / want[k].pow(2).mean().sqrt()).item()
        assert rms < 3e-2, (k, rms)
    first = "stage1.rebnconvin.conv_s1"
    assert ((got[first] - want[first]).pow(2).mean().sqrt() / want[first].pow(2).mean().sqrt()).item() < 1e-4
    lg, lr = _loss(outs, mask.to(DEV)).item(), torch_ref.model_loss(ref, mask).item()
    assert abs(lg - lr) < 0.05 * lr, (lg, lr)


def test_u2net_bf16_trains_and_in_place_gradients_match():
    torch.manual_seed(0)
    m = unet_zoo_amd.create_model("u2net").to(DEV).train()
    x, mask = torch_ref.synthetic_batch(2, 3, 64, 64, seed=7)
    x, mask = x.to(DEV), mask.to(DEV)
    # in-place gradient mode (flat buffers / hipGraph capture) must give the same numbers
    _loss(m(x), mask).backward()
    ref = {n: p.grad.clone() for n, p in m.named_parameters()}
    state = {k: v.clone() for k, v in m.state_dict().items()}
    m.load_state_dict(state)
    for p in m.parameters():
        p.grad = torch.zeros_like(p)
    m.grads_in_place = True
    # same batch statistics: the first step only moved running stats, which train mode does not read
    _loss(m(x), mask).backward()
    for n, p in m.named_parameters():
        assert torch.equal(p.grad, ref[n]), n
    m.grads_in_place = False
    # (4 steps at lr 1e-3 sit inside the warm-up spike of this random-init network -- 5.32, 7.22, 5.41, 5.34 -- and pass
    # or fail by the third decimal of a different but equally correct kernel; 12 steps at 3e-4 are past it)
    opt = torch.optim.AdamW(m.parameters(), lr=3e-4)
    losses = []
    for _ in range(12):
        opt.zero_grad()
        loss = _loss(m(x), mask)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


@pytest.mark.parametrize("name,H,W", [("u2netp", 72, 56), ("u2netp", 50, 83), ("u2net", 37, 64)])
def test_u2net_odd_sizes_ceil_mode_pooling(name, H, W):
    """MaxPool2d(2, 2, ceil_mode=True) with odd maps (clipped border windows, u2net.py:30, 221-229) and
    _upsample_like between unequal sizes: forward + backward against the oracle, fp32"""
    torch.manual_seed(4)
    m = unet_zoo_amd.create_model(name)
    m.run_dtype = torch.float32
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(DEV).train()
    x, mask = torch_ref.synthetic_batch(2, 3, H, W, seed=6)
    outs = m(x.to(DEV))
    loss = _loss(outs, mask.to(DEV))
    loss.backward()
    st = torch_ref.clone_state(sd0, requires_grad=True)
    ref = torch_ref.u2net_forward(st, x, True)
    rloss = torch_ref.model_loss(ref, mask)
    names = [k for k, v in st.items() if v.requires_grad]
    rg = dict(zip(names, torch.autograd.grad(rloss, [st[k] for k in names])))
    for k in ref:
        got, want = outs[k].detach().cpu(), ref[k].detach()
        assert got.shape == want.shape
        assert (got - want).abs().max() <= 2e-3 * want.abs().max(), k
    assert abs(loss.item() - rloss.item()) < 5e-5
    keep = [n for n, _ in m.named_parameters() if not n.endswith("conv_s1.bias")]
    named = dict(m.named_parameters())
    gflat = torch.cat([named[n].grad.flatten().cpu() for n in keep])
    rflat = torch.cat([rg[n].flatten() for n in keep])
    cos = F.cosine_similarity(gflat.double(), rflat.double(), dim=0).item()
    assert cos > 0.99, cos


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("H,W,ceil", [(7, 9, True), (7, 9, False), (1, 5, True), (6, 3, False)])
def test_fused_pool_odd_sizes(dt, H, W, ceil):
    """BN-apply + ReLU + MaxPool2d(2, 2, ceil_mode=ceil) and its backward on odd maps"""
    g = torch.Generator().manual_seed(38)
    N, C = 2, 16
    y = rnd(dt, torch.randn(N, C, H, W, generator=g))
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    a = F.relu(y * sc.view(1, C, 1, 1) + sh.view(1, C, 1, 1))
    a_r = rnd(dt, a).requires_grad_(True)
    pooled_ref = F.max_pool2d(a_r, 2, stride=2, ceil_mode=ceil)
    ya = act_from_nchw(y.to(DEV), dt)
    act = ops.new_act(N, H, W, C, dt, DEV)
    pooled = ops.new_act(N, pooled_ref.shape[2], pooled_ref.shape[3], C, dt, DEV)
    ops.bn_relu_apply(ya, sc.to(DEV), sh.to(DEV), act, pooled, None, ceil)
    assert relerr(act.dense().cpu(), a_r.detach()) < (1e-6 if dt == torch.float32 else 8e-3)
    assert torch.equal(pooled.dense().cpu(), F.max_pool2d(act.dense().cpu(), 2, stride=2, ceil_mode=ceil))
    # gradient merge: direct + pooled gradients, routed to the first maximum of each (clipped) window
    g0 = rnd(dt, torch.randn(N, C, H, W, generator=g))
    gp = rnd(dt, torch.randn(pooled_ref.shape, generator=g))
    stored = act.dense().cpu().requires_grad_(True)
    (F.max_pool2d(stored, 2, stride=2, ceil_mode=ceil) * gp).sum().backward()
    tot = ops.new_act(N, H, W, C, dt, DEV)
    ops.pool_grad_combine(act, act_from_nchw(g0.to(DEV), dt), None, act_from_nchw(gp.to(DEV), dt), tot, ceil)
    assert relerr(tot.dense().cpu(), g0 + stored.grad) < (1e-6 if dt == torch.float32 else 8e-3)
