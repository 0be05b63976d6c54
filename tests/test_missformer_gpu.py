"""GPU: MISSFormer on the HIP engine through the C ABI against the reference's golden vectors
(tests/golden/missformer_*, image_size 128) and against the oracle at the registry's 512x512."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import unet_zoo_amd
from oracle import torch_ref
from unet_zoo_amd.models.missformer import MISSFormer

DEV = "cuda"


def relerr(a, b):
    return ((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30)).item()


def _golden(golden_dir, tag):
    with open(os.path.join(golden_dir, tag + ".json")) as f:
        meta = json.load(f)
    return meta, np.load(os.path.join(golden_dir, tag + ".npz"))


def _model(size, dtype=torch.float32):
    torch.manual_seed(0)
    m = MISSFormer(num_classes=1, in_channels=3, image_size=size)
    m.run_dtype = dtype
    return m


def test_missformer_fp32_step_matches_reference_golden(golden_dir):
    meta, arr = _golden(golden_dir, "missformer_b2_128")
    x, mask = torch_ref.synthetic_batch(2, 3, 128, 128, seed=1)
    m = _model(128).to(DEV).train()
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
    assert {n for n, p in named.items() if p.grad is not None} == set(meta["grad_l2"])
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in named.values() if p.grad is not None)).item()
    assert abs(gn - meta["global_grad_norm"]) < 3e-3 * meta["global_grad_norm"]
    for name, rn in meta["grad_l2"].items():
        g = named[name].grad
        assert abs(g.double().norm().item() - rn) <= 2e-2 * rn + 1e-5 * meta["global_grad_norm"], (name, g.norm().item(), rn)
    for name in ("backbone.patch_embed1.proj.weight", "bridge.bridge_layer2.attn.scale_reduce.sr_convs.0.weight",
                 "decoder_1.layer_former_1.mlp.dwconv.dwconv.weight", "backbone.block3.0.attn.kv.weight"):
        gv = named[name].grad.flatten().cpu()[arr["gidx/" + name]].numpy()
        np.testing.assert_allclose(gv, arr["gval/" + name], rtol=5e-2, atol=2e-3 * np.abs(arr["gval/" + name]).max())
    m.eval()
    with torch.no_grad():
        ev = m(x.to(DEV)).cpu()
    evr = torch.from_numpy(arr["eval_logits"])
    assert (ev - evr).abs().max() <= 1e-3 * evr.abs().max()


def test_missformer_bf16_against_oracle_and_trains():
    x, mask = torch_ref.synthetic_batch(2, 3, 256, 256, seed=5)
    m = _model(256, torch.bfloat16).to(DEV).train()
    sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    ref_logits, ref_loss, ref_grads, _ = torch_ref.train_step_reference("missformer", sd0, x, mask, image_size=256)
    xs, ms = x.to(DEV), mask.to(DEV)
    logits = m(xs)
    loss = F.binary_cross_entropy_with_logits(logits, ms)
    loss.backward()
    assert relerr(logits.detach().cpu(), ref_logits) < 6e-2
    assert abs(loss.item() - ref_loss.item()) < 2e-2
    a = torch.cat([p.grad.flatten().cpu() for n, p in m.named_parameters() if n in ref_grads])
    b = torch.cat([ref_grads[n].flatten() for n, p in m.named_parameters() if n in ref_grads])
    assert F.cosine_similarity(a.double(), b.double(), dim=0).item() > 0.97
    opt = torch.optim.AdamW(m.parameters(), lr=2e-4)
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


def test_missformer_registry_builds_for_512_like_the_reference():
    """create_model drops image_size (models/__init__.py:145-148): the model is always the 512x512 one"""
    assert "missformer" in unet_zoo_amd.hip_models()
    torch.manual_seed(0)
    m = unet_zoo_amd.create_model("missformer", in_channels=3, num_classes=1, image_size=224, depth=4)
    assert m.image_size == 512
    m.run_dtype = torch.bfloat16
    m = m.to(DEV).train()
    with pytest.raises(ValueError):
        m(torch.zeros(1, 3, 224, 224, device=DEV))
    x, mask = torch_ref.synthetic_batch(1, 3, 512, 512, seed=3)
    sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        ref = torch_ref.missformer_forward(sd0, x, True, image_size=512)
    logits = m(x.to(DEV))
    assert logits.shape == (1, 1, 512, 512)
    assert relerr(logits.detach().cpu(), ref) < 6e-2
    F.binary_cross_entropy_with_logits(logits, mask.to(DEV)).backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)


def test_token_mlp_modes_mix_and_fc():
    """MISSFormer(token_mlp_mode=...) (missformer.py:253-263): 'mix' = MixFFN (no skip, no LayerNorm) against the
    oracle in fp32; 'fc' constructs and fails on the first forward with the reference's own TypeError"""
    from unet_zoo_amd.models import MISSFormer
    torch.manual_seed(0)
    m = MISSFormer(num_classes=1, in_channels=3, token_mlp_mode="mix", image_size=128)
    m.run_dtype = torch.float32
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    assert not any(".mlp.norm1." in k for k in sd if k.startswith("backbone.block"))
    m = m.to(DEV).train()
    x, mask = torch_ref.synthetic_batch(2, 3, 128, 128, seed=3)
    ref_logits, ref_loss, ref_grads, _ = torch_ref.train_step_reference("missformer", sd, x, mask, image_size=128)
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
    torch.manual_seed(0)
    fc = MISSFormer(num_classes=1, in_channels=3, token_mlp_mode="fc", image_size=128).to(DEV)
    with pytest.raises(TypeError):
        fc(x.to(DEV))
