"""GPU: the whole UNet hot path (forward + backward through the C ABI) against the golden vectors
generated from the reference and against the CPU oracle."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import unet_zoo_amd
from oracle import torch_ref

DEV = "cuda"


def _golden(golden_dir, tag):
    with open(os.path.join(golden_dir, tag + ".json")) as f:
        meta = json.load(f)
    return meta, np.load(os.path.join(golden_dir, tag + ".npz"))


def _model(dtype):
    torch.manual_seed(0)
    m = unet_zoo_amd.create_model("unet", in_channels=3, num_classes=1)
    m.run_dtype = dtype
    return m.to(DEV)


def _step(m, x, mask):
    m.train()
    m.zero_grad()
    logits = m(x.to(DEV))
    loss = F.binary_cross_entropy_with_logits(logits, mask.to(DEV))
    loss.backward()
    return logits.detach().cpu(), loss.item()


def test_fp32_step_matches_reference_golden(golden_dir):
    """B=2, 64x64, fp32 run dtype: logits, loss, every parameter gradient, running statistics and
    the following eval-mode forward, all within the north-star tolerance (1e-3 relative)."""
    meta, arr = _golden(golden_dir, "unet_b2_64")
    x, mask = torch_ref.synthetic_batch(2, 3, 64, 64, seed=1)
    m = _model(torch.float32)
    logits, loss = _step(m, x, mask)
    ref = torch.from_numpy(arr["train_logits"])
    assert (logits - ref).abs().max() <= 1e-3 * ref.abs().max()
    assert torch.equal(logits > 0, ref > 0) or ((logits > 0) != (ref > 0)).sum() == 0
    assert abs(loss - meta["loss"]) < 1e-5
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters())).item()
    assert abs(gn - meta["global_grad_norm"]) < 1e-3 * meta["global_grad_norm"]
    # Gradients: every kernel is individually exact to ~1e-6 (test_ops_gpu.py), but the network's
    # backward is discontinuous in the forward values: one pre-activation within 1e-6 of zero flips
    # its ReLU mask (or a pool argmax) between two fp32 summation orders, which moves a layer's
    # gradient by ~1/sqrt(pixels).  Measured spread vs the reference at this size: <= 5e-3.
    for name, p in m.named_parameters():
        rn = meta["grad_l2"][name]
        got = p.grad.double().norm().item()
        if name.endswith("conv_op.0.bias") or name.endswith("conv_op.3.bias"):
            assert got <= 1e-5          # analytically zero under train-mode BN; reference = rounding noise
            assert rn <= 1e-5
            continue
        assert abs(got - rn) <= 1e-2 * rn + 1e-7, (name, got, rn)
        idx = arr["gidx/" + name]
        gv = p.grad.flatten()[torch.from_numpy(idx).to(DEV)].cpu().numpy()
        scale = np.abs(arr["gval/" + name]).max() + 1e-12
        err = np.abs(gv - arr["gval/" + name])
        assert np.median(err) <= 5e-3 * scale + 1e-7, name          # the bulk agrees (first layers: ~3e-3)
        assert err.max() <= 0.15 * scale + 1e-7, name               # a flipped mask moves single entries
    sd = m.state_dict()
    for k in ("down_convolution_1.conv.conv_op.1", "bottle_neck.conv_op.4", "up_convolution_4.conv.conv_op.4"):
        np.testing.assert_allclose(sd[k + ".running_mean"].cpu().numpy(), arr["rm/" + k], rtol=1e-3, atol=1e-5)
        np.testing.assert_allclose(sd[k + ".running_var"].cpu().numpy(), arr["rv/" + k], rtol=1e-3, atol=1e-5)
        assert int(sd[k + ".num_batches_tracked"]) == 1
    m.eval()
    with torch.no_grad():
        ev = m(x.to(DEV)).cpu()
    evr = torch.from_numpy(arr["eval_logits"])
    assert (ev - evr).abs().max() <= 1e-3 * evr.abs().max()


def test_fp32_config0_b2_256_matches_reference_golden(golden_dir):
    """BASELINE.json configs[0] shape (B=2, 3x256x256)."""
    meta, arr = _golden(golden_dir, "unet_b2_256")
    x, mask = torch_ref.synthetic_batch(2, 3, 256, 256, seed=1)
    m = _model(torch.float32)
    logits, loss = _step(m, x, mask)
    flat = logits.flatten()
    ref = torch.from_numpy(arr["train_logits_sampled"])
    got = flat[torch.from_numpy(arr["logit_idx"])]
    assert (got - ref).abs().max() <= 1e-3 * ref.abs().max()
    assert abs(loss - meta["loss"]) < 1e-5
    assert int((logits > 0).sum()) == meta["train_positive_pixels"]   # bit-exact mask
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters())).item()
    assert abs(gn - meta["global_grad_norm"]) < 1e-3 * meta["global_grad_norm"]


def test_bf16_step_close_to_reference_with_margin_aware_mask(golden_dir):
    """bf16 throughput mode: logits within a bf16-sized tolerance; masks must agree wherever the
    reference logit is not within that tolerance of the threshold."""
    meta, arr = _golden(golden_dir, "unet_b2_64")
    x, mask = torch_ref.synthetic_batch(2, 3, 64, 64, seed=1)
    m = _model(torch.bfloat16)
    logits, loss = _step(m, x, mask)
    ref = torch.from_numpy(arr["train_logits"])
    tol = 0.06 * ref.abs().max()
    assert (logits - ref).abs().max() <= tol
    safe = ref.abs() > tol
    assert torch.equal((logits > 0)[safe], (ref > 0)[safe])
    assert abs(loss - meta["loss"]) < 2e-2
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters())).item()
    assert abs(gn - meta["global_grad_norm"]) < 0.1 * meta["global_grad_norm"]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_against_oracle_on_other_shape_and_classes(dtype):
    """non-square input, 2 classes, different seed: HIP path vs the CPU oracle on the same state"""
    torch.manual_seed(11)
    m = unet_zoo_amd.create_model("unet", in_channels=3, num_classes=2)
    m.run_dtype = dtype
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(DEV).train()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, 64, 96, generator=g)
    t = (torch.rand(2, 2, 64, 96, generator=g) > 0.5).float()
    logits = m(x.to(DEV))
    F.binary_cross_entropy_with_logits(logits, t.to(DEV)).backward()
    st = torch_ref.clone_state(sd, requires_grad=True)
    # bf16: the oracle rounds to bf16 where the engine stores bf16 (tests/test_bf16_rounded_oracle_gpu.py has the
    # layer-by-layer version); what is left is summation order and half-way roundings
    torch_ref.set_storage_rounding(None if dtype == torch.float32 else torch.bfloat16)
    try:
        ref = torch_ref.unet_forward(st, x, True)
        F.binary_cross_entropy_with_logits(ref, t).backward()
    finally:
        torch_ref.set_storage_rounding(None)
    rel = 1e-3 if dtype == torch.float32 else 2e-2
    assert (logits.detach().cpu() - ref.detach()).abs().max() <= rel * ref.detach().abs().max()
    dots = []
    for name, p in m.named_parameters():
        rg = st[name].grad
        if rg.abs().max() < 1e-6:
            continue
        err = (p.grad.cpu() - rg).norm() / rg.norm()
        if dtype == torch.float32:
            assert err <= 2e-2, (name, err.item())
        else:
            # against the storage-rounded oracle the ReLU masks are (nearly) the same on both sides; round 1 compared
            # with the fp32 reference, where ~0.3 % of the masks flip, and had to accept cosine 0.75 per layer
            cos = F.cosine_similarity(p.grad.cpu().flatten(), rg.flatten(), dim=0)
            if rg.norm() > 2e-3 * torch.sqrt(sum((v.grad.double() ** 2).sum() for v in st.values() if v.grad is not None)):
                assert cos >= 0.9, (name, cos.item())
        dots.append((p.grad.cpu().flatten(), rg.flatten()))
    a, b = torch.cat([d[0] for d in dots]), torch.cat([d[1] for d in dots])
    assert F.cosine_similarity(a, b, dim=0) >= (0.9999 if dtype == torch.float32 else 0.99)


def test_full_size_properties_bf16():
    """BASELINE configs[1] size (B=16, 256x256, bf16): size-independent properties — finite
    outputs, BN invariants (per-channel mean of the BN output gradient is ~0 so conv-bias grads
    vanish), running stats moved, deterministic forward."""
    torch.manual_seed(0)
    m = unet_zoo_amd.create_model("unet").to(DEV).train()
    x, mask = torch_ref.synthetic_batch(16, 3, 256, 256, seed=2)
    x, mask = x.to(DEV), mask.to(DEV)
    a = m(x)
    loss = F.binary_cross_entropy_with_logits(a, mask)
    loss.backward()
    assert a.shape == (16, 1, 256, 256) and torch.isfinite(a).all()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())
    assert 0.5 < loss.item() < 1.0
    rm = m.state_dict()["down_convolution_1.conv.conv_op.1.running_mean"]
    assert rm.abs().max() > 0
    with torch.no_grad():
        b = m(x)
    assert torch.equal(a.detach(), b)   # same batch statistics -> identical forward


def test_three_training_steps_track_the_oracle():
    """several optimizer steps (weights change between forwards): the packed kernel-layout copies
    must follow the master weights; loss trajectory vs the CPU oracle with the same AdamW"""
    torch.manual_seed(0)
    m = unet_zoo_amd.create_model("unet")
    m.run_dtype = torch.float32
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(DEV).train()
    x, mask = torch_ref.synthetic_batch(2, 3, 32, 32, seed=3)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-5)
    st = torch_ref.clone_state(sd, requires_grad=True)
    ropt = torch.optim.AdamW([v for v in st.values() if v.requires_grad], lr=1e-3, weight_decay=1e-5)
    for step in range(3):
        opt.zero_grad()
        loss = F.binary_cross_entropy_with_logits(m(x.to(DEV)), mask.to(DEV))
        loss.backward()
        opt.step()
        ropt.zero_grad()
        rloss = F.binary_cross_entropy_with_logits(torch_ref.unet_forward(st, x, True), mask)
        rloss.backward()
        ropt.step()
        assert abs(loss.item() - rloss.item()) < 2e-3 * (step + 1), (step, loss.item(), rloss.item())
    assert loss.item() < 0.72   # and it actually trains


def test_in_place_gradient_mode_matches_autograd_mode():
    """grads_in_place: kernels overwrite pre-allocated p.grad views of one flat buffer (what
    bench.py's hipGraph path and a flat all-reduce use); must equal the ordinary autograd result"""
    torch.manual_seed(0)
    m = unet_zoo_amd.create_model("unet").to(DEV).train()
    m.run_dtype = torch.float32
    x, mask = torch_ref.synthetic_batch(2, 3, 32, 32, seed=4)
    x, mask = x.to(DEV), mask.to(DEV)
    F.binary_cross_entropy_with_logits(m(x), mask).backward()
    ref = [p.grad.clone() for p in m.parameters()]
    params = list(m.parameters())
    flat = torch.full((sum(p.numel() for p in params),), 0.0, device=DEV)
    off = 0
    for p in params:
        p.grad = flat[off:off + p.numel()].view_as(p)
        off += p.numel()
    m.grads_in_place = True
    for _ in range(2):   # twice: overwrite semantics, not accumulation
        F.binary_cross_entropy_with_logits(m(x), mask).backward()
    for p, r in zip(params, ref):
        assert p.grad.data_ptr() >= flat.data_ptr() and p.grad.data_ptr() < flat.data_ptr() + flat.numel() * 4
        scale = r.abs().max() + 1e-12
        # running statistics moved between the passes only through BN buffers, not the math of this batch
        assert (p.grad - r).abs().max() <= 1e-4 * scale + 1e-9


@pytest.mark.parametrize("name", ["unet", "attention_unet", "u2netp", "nested_unet", "resunet", "swin_unet_v2", "missformer"])
def test_phased_backward_equals_one_shot_backward(name):
    """graph.PhasedStep (bench.py's multi-GPU graph mode): the backward cut into phases, each
    phase's parameters laid out contiguously in one flat buffer, gives bit-identical gradients to the
    autograd path; the plan covers every parameter exactly once and its cuts descend to 0."""
    from unet_zoo_amd.graph import PhasedStep
    torch.manual_seed(0)
    kw = dict(image_size=64, window_size=4, drop_path_rate=0.0) if name == "swin_unet_v2" else {}
    if name == "missformer":       # the registry always builds the 512x512 model (as the reference): take the class
        from unet_zoo_amd.models.missformer import MISSFormer
        m = MISSFormer(image_size=64).to(DEV).train()
    else:
        m = unet_zoo_amd.create_model(name, **kw).to(DEV).train()
    x, mask = torch_ref.synthetic_batch(2, 3, 64, 64, seed=4)
    x, mask = x.to(DEV), mask.to(DEV)

    def loss_fn(out, t):
        if isinstance(out, dict):
            return sum(F.binary_cross_entropy_with_logits(v, t) for v in out.values())
        return F.binary_cross_entropy_with_logits(out, t)

    loss_ref = loss_fn(m(x), mask)
    loss_ref.backward()
    # swin_unet_v2 constructs mlp / norm2 members that its forward never calls (swin_unet_v2.py:264-267):
    # they have no gradient and stay out of the plan, as in bench.py
    params = [p for p in m.parameters() if p.grad is not None]
    ref = {p: p.grad.clone() for p in params}
    for p in params:
        p.grad = torch.zeros_like(p)
    ps = PhasedStep(m, loss_fn)
    ps.forward(x, mask)
    ps.backward(ps.n_entries, 0, True)
    cuts, groups = ps.plan((0.4, 0.8))
    ps.finish()
    assert cuts[0] > cuts[-1] == 0 and all(a >= b for a, b in zip(cuts, cuts[1:])) and len(groups) == len(cuts) - 1
    flat_ids = [id(p) for g in groups for p in g]
    assert sorted(flat_ids) == sorted(id(p) for p in params)          # every parameter exactly once
    assert len(groups) == 3 and all(len(g) > 0 for g in groups)
    flat = torch.zeros(sum(p.numel() for p in params), device=DEV)
    off, spans = 0, []
    for g in groups:
        a0 = off
        for p in g:
            p.grad = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        spans.append((a0, off))
    loss = ps.forward(x, mask)
    for k in range(len(groups)):
        ps.backward(cuts[k], cuts[k + 1], k == 0)
        # after phase k its span must already hold the final values (that is what gets all-reduced)
        a0, a1 = spans[k]
        snap = flat[a0:a1].clone()
        spans[k] = (a0, a1, snap)
    ps.finish()
    assert abs(loss.item() - loss_ref.item()) < 1e-6
    for (a0, a1, snap) in spans:
        assert torch.equal(flat[a0:a1], snap)                        # later phases did not touch it
    for p in params:
        assert torch.equal(p.grad, ref[p])


@pytest.mark.parametrize("H,W", [(50, 70), (37, 44), (65, 31)])
def test_unet_sizes_not_divisible_by_16(H, W):
    """floor-mode pooling of odd maps and UpSample_UNet's zero padding of the transposed-conv output to the
    skip's size (common_layers.py:90, 110-113): forward + backward against the oracle, fp32"""
    torch.manual_seed(5)
    m = unet_zoo_amd.create_model("unet", in_channels=3, num_classes=2)
    m.run_dtype = torch.float32
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(DEV).train()
    x, mask = torch_ref.synthetic_batch(2, 3, H, W, seed=8)
    mask = mask.expand(-1, 2, -1, -1).contiguous()
    logits = m(x.to(DEV))
    loss = F.binary_cross_entropy_with_logits(logits, mask.to(DEV))
    loss.backward()
    ref_logits, ref_loss, ref_grads, st = torch_ref.train_step_reference("unet", sd0, x, mask)
    got = logits.detach().cpu()
    assert got.shape == ref_logits.shape == (2, 2, H, W)
    assert (got - ref_logits).abs().max() <= 1e-3 * ref_logits.abs().max()
    assert abs(loss.item() - ref_loss.item()) < 1e-5
    named = dict(m.named_parameters())
    keep = [n for n in ref_grads if not (n.endswith(".0.bias") or n.endswith(".3.bias"))]   # conv biases before a BatchNorm
    gflat = torch.cat([named[n].grad.flatten().cpu() for n in keep])
    rflat = torch.cat([ref_grads[n].flatten() for n in keep])
    cos = F.cosine_similarity(gflat.double(), rflat.double(), dim=0).item()
    assert cos > 0.999, cos
    assert abs(gflat.double().norm().item() / rflat.double().norm().item() - 1) < 1e-2
    sd = m.state_dict()
    for k in ("down_convolution_1.conv.conv_op.1", "bottle_neck.conv_op.4", "up_convolution_4.conv.conv_op.4"):
        np.testing.assert_allclose(sd[k + ".running_var"].cpu().numpy(), st[k + ".running_var"].numpy(), rtol=2e-3, atol=1e-5)


@pytest.mark.parametrize("name", ["unet", "resunet", "attention_unet"])
def test_backward_through_eval_mode_batchnorm_matches_the_oracle(name):
    """model.eval() + loss.backward() -- fine-tuning with frozen BatchNorm, which the reference's nn.Modules allow
    (running statistics as constants: dy = scale * g * mask, the conv bias receives a gradient): fp32 run mode against
    autograd on the oracle in eval mode, every parameter"""
    torch.manual_seed(3)
    m = unet_zoo_amd.create_model(name, in_channels=3, num_classes=1)
    m.run_dtype = torch.float32
    # running statistics that are not the identity, so that the test sees them used
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.normal_(0, 0.05)
                mod.running_var.uniform_(0.5, 1.5)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(DEV).eval()
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 3, 64, 64, generator=g)
    t = (torch.rand(2, 1, 64, 64, generator=g) > 0.5).float()
    logits = m(x.to(DEV))
    F.binary_cross_entropy_with_logits(logits, t.to(DEV)).backward()
    st = torch_ref.clone_state(sd, requires_grad=True)
    ref = torch_ref.FORWARDS[name](st, x, False)
    F.binary_cross_entropy_with_logits(ref, t).backward()
    assert (logits.detach().cpu() - ref.detach()).abs().max() <= 1e-3 * ref.detach().abs().max()
    checked = 0
    for pname, p in m.named_parameters():
        rg = st[pname].grad
        assert p.grad is not None, pname
        if rg.abs().max() < 1e-7:
            continue
        err = (p.grad.cpu() - rg).norm() / rg.norm()
        assert err <= 2e-2, (pname, err.item())
        checked += 1
    assert checked > 20
    # the running statistics did not move
    for k, v in m.state_dict().items():
        if "running_" in k:
            assert torch.equal(v.cpu(), sd[k]), k
