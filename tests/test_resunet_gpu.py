"""GPU: ResUnet on the HIP engine through the C ABI: the statistics pass + BatchNorm (with and without ReLU) of
a tensor that is not a convolution output, and the whole model against the reference's golden vectors
(tests/golden/resunet_*) and the oracle."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import unet_zoo_amd
from oracle import torch_ref
from unet_zoo_amd import ops
from unet_zoo_amd.ops import act_from_nchw

DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16]


def rnd(dt, t):
    return t.to(dt).float()


def relerr(a, b):
    return ((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30)).item()


def _golden(golden_dir, tag):
    with open(os.path.join(golden_dir, tag + ".json")) as f:
        meta = json.load(f)
    return meta, np.load(os.path.join(golden_dir, tag + ".npz"))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("N,C,H,W", [(2, 64, 16, 16), (3, 128, 5, 7), (1, 24, 33, 9)])
def test_standalone_batchnorm_statistics_apply_backward(dt, relu, N, C, H, W):
    """uz_colstats -> uz_bn_finalize -> uz_bn_relu_add_apply (flag bit 1 = no ReLU) -> two-pass backward, against
    F.batch_norm(training=True) [+ relu] (ResidualConv's pre-activation and skip BatchNorms)"""
    g = torch.Generator().manual_seed(81)
    x = rnd(dt, torch.randn(N, C, H, W, generator=g) * 1.5 + 0.3).requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.randn(C, generator=g) * 0.2).requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    ref = F.batch_norm(x, rm, rv, gamma, beta, training=True, momentum=0.1, eps=1e-5)
    if relu:
        ref = F.relu(ref)
    dy = rnd(dt, torch.randn(N, C, H, W, generator=g))
    ref.backward(dy)
    xa = act_from_nchw(x.detach().to(DEV), dt)
    rmd, rvd = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    vec = ops.bn_finalize(ops.colstats(xa), xa.P, gamma.detach().to(DEV), beta.detach().to(DEV), 1e-5, 0.1, rmd, rvd)
    act = ops.new_act(N, H, W, C, dt, DEV)
    ops.bn_relu_apply(xa, vec[0], vec[1], act, relu=relu)
    assert relerr(act.dense().cpu(), ref.detach()) < (5e-6 if dt == torch.float32 else 1e-2)
    np.testing.assert_allclose(rmd.cpu().numpy(), rm.numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(rvd.cpu().numpy(), rv.numpy(), rtol=1e-4, atol=1e-6)
    dx = ops.new_act(N, H, W, C, dt, DEV)
    sums = torch.empty(2, C, dtype=torch.float64, device=DEV)
    dgam, dbet = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    ops.bn_relu_bwd(xa, vec, act_from_nchw(dy.to(DEV), dt), None, None, sums, dx, dgam, dbet, relu=relu)
    assert relerr(dx.dense().cpu(), x.grad) < (2e-5 if dt == torch.float32 else 2e-2)
    assert relerr(dgam.cpu(), gamma.grad) < 2e-4 and relerr(dbet.cpu(), beta.grad) < 2e-4


def _model(dtype=torch.float32, K=1):
    torch.manual_seed(0)
    m = unet_zoo_amd.create_model("resunet", in_channels=3, num_classes=K)
    m.run_dtype = dtype
    return m


def test_resunet_fp32_step_matches_reference_golden(golden_dir):
    meta, arr = _golden(golden_dir, "resunet_b2_64")
    x, mask = torch_ref.synthetic_batch(2, 3, 64, 64, seed=1)
    m = _model().to(DEV).train()
    logits = m(x.to(DEV))
    loss = F.binary_cross_entropy_with_logits(logits, mask.to(DEV))
    loss.backward()
    ref = torch.from_numpy(arr["train_logits"])
    got = logits.detach().cpu()
    assert (got - ref).abs().max() <= 1e-3 * ref.abs().max()        # north-star bound
    sure = ref.abs() > 1e-4 * ref.abs().max()
    assert torch.equal((got > 0)[sure], (ref > 0)[sure])
    assert abs(loss.item() - meta["loss"]) < 1e-5
    named = dict(m.named_parameters())
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in named.values())).item()
    assert abs(gn - meta["global_grad_norm"]) < 3e-3 * meta["global_grad_norm"]
    for name, rn in meta["grad_l2"].items():
        g = named[name].grad
        if name == "input_layer.0.bias":                            # in front of a train-mode BatchNorm: analytically zero
            assert g.abs().max().item() <= 1e-5 and rn < 1e-4
            continue
        assert abs(g.double().norm().item() - rn) <= 2e-2 * rn + 1e-5 * meta["global_grad_norm"], (name, g.norm().item(), rn)
    sd = m.state_dict()
    for k in ("input_layer.1", "residual_conv_1.conv_block.0", "residual_conv_2.conv_skip.1", "bridge.conv_block.3",
              "up_residual_conv1.conv_block.0", "up_residual_conv3.conv_skip.1"):
        np.testing.assert_allclose(sd[k + ".running_mean"].cpu().numpy(), arr["rm/" + k], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(sd[k + ".running_var"].cpu().numpy(), arr["rv/" + k], rtol=1e-4, atol=1e-6)
    m.eval()
    with torch.no_grad():
        ev = m(x.to(DEV)).cpu()
    evr = torch.from_numpy(arr["eval_logits"])
    assert (ev - evr).abs().max() <= 1e-3 * evr.abs().max()


def test_resunet_bf16_against_oracle_and_trains():
    x, mask = torch_ref.synthetic_batch(4, 3, 64, 96, seed=5)
    m = _model(dtype=torch.bfloat16).to(DEV).train()
    sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    ref_logits, ref_loss, ref_grads, _ = torch_ref.train_step_reference("resunet", sd0, x, mask)
    xs, ms = x.to(DEV), mask.to(DEV)
    logits = m(xs)
    loss = F.binary_cross_entropy_with_logits(logits, ms)
    loss.backward()
    assert relerr(logits.detach().cpu(), ref_logits) < 6e-2
    assert abs(loss.item() - ref_loss.item()) < 2e-2
    a = torch.cat([p.grad.flatten().cpu() for n, p in m.named_parameters() if n in ref_grads])
    b = torch.cat([ref_grads[n].flatten() for n, p in m.named_parameters() if n in ref_grads])
    assert F.cosine_similarity(a.double(), b.double(), dim=0).item() > 0.9
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    losses = []
    for _ in range(6):
        opt.zero_grad(set_to_none=True)
        l = F.binary_cross_entropy_with_logits(m(xs), ms)
        l.backward()
        opt.step()
        losses.append(l.item())
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    m.eval()
    with torch.no_grad():
        assert torch.equal(m(xs), m(xs))


def test_resunet_registry_and_size_check():
    assert "resunet" in unet_zoo_amd.hip_models()
    m = _model().to(DEV)
    with pytest.raises(ValueError):
        m(torch.zeros(1, 3, 36, 64, device=DEV))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 32, 32, 64, 128), (1, 64, 128, 128, 256), (1, 16, 32, 256, 512),
                                            (1, 14, 10, 64, 64), (2, 8, 8, 96, 40)])
def test_conv3x3_stride2_forward_and_weight_gradient(dt, N, H, W, Cin, Cout):
    """UZ_TAPS_CONV_S2: Conv2d(k3, stride 2, padding 1) (ResidualConv, common_layers.py:188) on the LDS-DMA GEMM and
    its weight gradient as a stride-2 gather (bf16 on map widths the LDS-DMA kernel takes)"""
    from unet_zoo_amd import _lib as L
    g = torch.Generator().manual_seed(83)
    x = rnd(dt, torch.randn(N, Cin, H, W, generator=g)).requires_grad_(True)
    w = rnd(dt, torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05).requires_grad_(True)
    ref = F.conv2d(x, w, None, stride=2, padding=1)
    dy = rnd(dt, torch.randn(ref.shape, generator=g))
    ref.backward(dy)
    Ho, Wo = ref.shape[2], ref.shape[3]
    xa = act_from_nchw(x.detach().to(DEV), dt)
    y = ops.new_act(N, Ho, Wo, Cout, dt, DEV)
    wp = ops.pack_weights(w.detach().to(DEV), L.PACK_CONV_FWD, dt)
    ops.conv_igemm(xa, wp, None, y, ntaps=9, taps_mode=L.TAPS_CONV_S2)
    assert relerr(y.dense().cpu(), ref.detach()) < (2e-5 if dt == torch.float32 else 2e-2)
    fast = dt == torch.bfloat16 and (Wo in (16, 32) or (Wo >= 64 and Wo % 64 == 0)) and Ho % (64 // min(Wo, 64)) == 0
    if fast:
        dw = ops.wgrad(act_from_nchw(dy.to(DEV), dt), xa, (Cout, Cin, 3, 3), ntaps=9, taps_mode=L.TAPS_CONV_S2)
        assert relerr(dw.cpu(), w.grad) < 2e-2


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("N,H,W,C", [(2, 8, 12, 64), (1, 7, 5, 32), (3, 16, 16, 96)])
def test_pixel_grid_moves(dt, N, H, W, C):
    """uz_resample2: copy into a wider buffer's slot, keep every second pixel, spread between zeros"""
    g = torch.Generator().manual_seed(85)
    x = rnd(dt, torch.randn(N, C, H, W, generator=g))
    xa = act_from_nchw(x.to(DEV), dt)
    full = ops.new_act(N, H, W, 2 * C, dt, DEV)
    full.buf.fill_(-3.0)
    ops.resample2(xa, full.window(C, C), ops.RESAMPLE_COPY)
    assert torch.equal(full.window(C, C).dense().cpu(), x) and bool((full.buf[:, :C] == -3.0).all())
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    sub = ops.new_act(N, Ho, Wo, C, dt, DEV)
    ops.resample2(xa, sub, ops.RESAMPLE_SUBSAMPLE)
    assert torch.equal(sub.dense().cpu(), x[:, :, ::2, ::2])
    back = ops.new_act(N, H, W, C, dt, DEV)
    back.buf.fill_(7.0)
    ops.resample2(sub, back, ops.RESAMPLE_ZERO_INSERT)
    want = torch.zeros_like(x)
    want[:, :, ::2, ::2] = x[:, :, ::2, ::2]
    assert torch.equal(back.dense().cpu(), want)
