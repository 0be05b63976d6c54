"""GPU: UNet++ (nested_unet) on the HIP engine through the C ABI: the align_corners=True bilinear resize, the
copy-into-concat alias, and the whole model (plain and deep supervision) against the reference's golden
vectors (tests/golden/nested_unet_*) and the oracle."""
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
@pytest.mark.parametrize("N,C,hi,wi,ho,wo", [(2, 32, 8, 8, 16, 16), (1, 64, 4, 6, 8, 12), (2, 16, 1, 1, 2, 2),
                                            (1, 128, 16, 16, 32, 32), (1, 8, 5, 3, 13, 4), (1, 8, 7, 9, 7, 1)])
def test_bilinear_resize_align_corners(dt, N, C, hi, wi, ho, wo):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) (nested_unet.py:32) and general sizes"""
    g = torch.Generator().manual_seed(71)
    x = rnd(dt, torch.randn(N, C, hi, wi, generator=g)).requires_grad_(True)
    dy = rnd(dt, torch.randn(N, C, ho, wo, generator=g))
    ref = F.interpolate(x, size=(ho, wo), mode="bilinear", align_corners=True)
    ref.backward(dy)
    xa = act_from_nchw(x.detach().to(DEV), dt)
    full = ops.new_act(N, ho, wo, 2 * C, dt, DEV)
    full.buf.zero_()
    out = full.window(C, C)                       # right half of a wider buffer, as the model's concat slot
    ops.bilinear_fwd(xa, out, align_corners=True)
    assert relerr(out.dense().cpu(), ref.detach()) < (1e-6 if dt == torch.float32 else 8e-3)
    assert float(full.buf[:, :C].abs().max()) == 0.0
    dx = ops.new_act(N, hi, wi, C, dt, DEV)
    ops.bilinear_bwd(act_from_nchw(dy.to(DEV), dt), dx, align_corners=True)
    assert relerr(dx.dense().cpu(), x.grad) < (2e-6 if dt == torch.float32 else 8e-3)


def _model(K=1, deep=False, dtype=torch.float32):
    torch.manual_seed(0)
    m = unet_zoo_amd.create_model("nested_unet", in_channels=3, num_classes=K, deep_supervision=deep)
    m.run_dtype = dtype
    return m


def test_nested_unet_fp32_step_matches_reference_golden(golden_dir):
    meta, arr = _golden(golden_dir, "nested_unet_b2_64")
    x, mask = torch_ref.synthetic_batch(2, 3, 64, 64, seed=1)
    m = _model().to(DEV).train()
    logits = m(x.to(DEV))
    loss = F.binary_cross_entropy_with_logits(logits, mask.to(DEV))
    loss.backward()
    ref = torch.from_numpy(arr["train_logits"])
    got = logits.detach().cpu()
    assert (got - ref).abs().max() <= 1e-3 * ref.abs().max()        # north-star bound
    margin = 1e-4 * ref.abs().max()                                 # masks: bit-exact outside fp32 summation noise
    sure = ref.abs() > margin
    assert torch.equal((got > 0)[sure], (ref > 0)[sure])
    assert abs(loss.item() - meta["loss"]) < 1e-5
    named = dict(m.named_parameters())
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in named.values())).item()
    assert abs(gn - meta["global_grad_norm"]) < 3e-3 * meta["global_grad_norm"]
    for name, rn in meta["grad_l2"].items():
        g = named[name].grad
        if name.endswith(("conv1.bias", "conv2.bias")):             # analytically zero in front of a train-mode BN
            assert g.abs().max().item() <= 1e-5 and rn < 1e-4, name
            continue
        assert abs(g.double().norm().item() - rn) <= 2e-2 * rn + 1e-5 * meta["global_grad_norm"], (name, g.norm().item(), rn)
    sd = m.state_dict()
    for k in ("conv0_0.bn1", "conv2_0.bn2", "conv4_0.bn1", "conv1_2.bn1", "conv0_4.bn2"):
        np.testing.assert_allclose(sd[k + ".running_mean"].cpu().numpy(), arr["rm/" + k], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(sd[k + ".running_var"].cpu().numpy(), arr["rv/" + k], rtol=1e-4, atol=1e-6)
    m.eval()
    with torch.no_grad():
        ev = m(x.to(DEV)).cpu()
    evr = torch.from_numpy(arr["eval_logits"])
    assert (ev - evr).abs().max() <= 1e-3 * evr.abs().max()


def test_nested_unet_deep_supervision_fp32_matches_reference_golden(golden_dir):
    meta, arr = _golden(golden_dir, "nested_unet_ds_b2_32x48")
    x, mask = torch_ref.synthetic_batch(2, 3, 32, 48, seed=3)
    mask2 = torch.cat([mask, 1.0 - mask], 1).to(DEV)
    m = _model(K=2, deep=True).to(DEV).train()
    outs = m(x.to(DEV))
    assert isinstance(outs, list) and len(outs) == 4                # nested_unet.py:95-101
    loss = sum(F.binary_cross_entropy_with_logits(o, mask2) for o in outs)
    loss.backward()
    for i, o in enumerate(outs):
        ref = torch.from_numpy(arr[f"train/{i}"])
        assert (o.detach().cpu() - ref).abs().max() <= 1e-3 * ref.abs().max(), i
    assert abs(loss.item() - meta["loss"]) < 2e-5
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters())).item()
    assert abs(gn - meta["global_grad_norm"]) < 3e-3 * meta["global_grad_norm"]


def test_nested_unet_bf16_against_oracle_and_trains():
    x, mask = torch_ref.synthetic_batch(4, 3, 64, 96, seed=5)
    m = _model(dtype=torch.bfloat16).to(DEV).train()
    sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    ref_logits, ref_loss, ref_grads, _ = torch_ref.train_step_reference("nested_unet", sd0, x, mask)
    xs, ms = x.to(DEV), mask.to(DEV)
    logits = m(xs)
    loss = F.binary_cross_entropy_with_logits(logits, ms)
    loss.backward()
    assert relerr(logits.detach().cpu(), ref_logits) < 5e-2
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


def test_nested_unet_registry_and_size_check():
    assert "nested_unet" in unet_zoo_amd.hip_models()
    m = _model().to(DEV)
    with pytest.raises(ValueError):
        m(torch.zeros(1, 3, 40, 64, device=DEV))
