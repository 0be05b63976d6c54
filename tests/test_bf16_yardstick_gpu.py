"""GPU, bf16 run mode of the models whose bf16 tests still compared against loose fixed bounds (VERDICT r2 weak list:
"the four new models are still compared with the fp32 oracle at 5-6 % / cosine 0.9"): a self-calibrating bound.

Three evaluations of the same train step from the same state:
  E  the engine in bf16,
  O  the oracle rounding to bf16 where the engine stores bf16 (oracle/torch_ref.set_storage_rounding),
  J  the same oracle with jitter of fp32-rounding size (1e-6) in front of every storage rounding: another correct
     bf16-storage implementation, one whose fp32 sums come out in a different order.
|O - J| is what the NETWORK makes of rounding-size noise (train-mode BatchNorm over a handful of samples, ReLU masks and
pool selections flipping); |E - O| beyond a small multiple of it would be a kernel computing something else."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import unet_zoo_amd
from oracle import torch_ref

DEV = "cuda"

# (model, constructor kwargs, H, W, oracle kwargs)
CASES = [
    ("transatt_unet", {}, 96, 64, {}),
    ("unet_transformer", {"common_attn_res_for_QK_V": (16, 24)}, 64, 96, {"res": (16, 24)}),
    ("uctransnet", {"image_size": 128}, 128, 128, {}),
    ("multiresunet", {}, 64, 96, {}),
    ("missformer", {"image_size": 128}, 128, 128, {"image_size": 128}),
    # (at 64 x 64 U2Net's six-level RSU stack over 2 x 2 ... 4 x 4 maps decorrelates the GRADIENT completely: cosine between
    # oracle and jittered oracle 0.005; 256 x 256 is the size its own tests use)
    ("u2net", {}, 256, 256, {}),
]


def _logical_grad(m, named, n, want):
    """the engine's gradient of parameter n in the reference's shape: a channel-padded model (multiresunet: holes in the middle
    of a concatenated channel axis) maps it through the same index lists its state_dict uses"""
    g = named[n].grad
    if g is None or g.shape == want.shape:
        return g
    mod_name, pname = n.rsplit(".", 1)
    mod = m.get_submodule(mod_name)
    g = g.detach()
    for dim, idx in getattr(mod, "_maps", {}).get(pname, ()):
        g = g.index_select(dim, torch.as_tensor(idx, device=g.device))
    assert g.shape == want.shape, (n, tuple(g.shape), tuple(want.shape))
    return g


def _first(out):
    if isinstance(out, dict):
        return next(iter(out.values()))
    return out[0] if isinstance(out, (list, tuple)) else out


def _build(name, kw):
    if name == "missformer":       # create_model always builds the 512 x 512 model (models/__init__.py:145-148)
        from unet_zoo_amd.models import MISSFormer
        return MISSFormer(num_classes=1, in_channels=3, **kw)
    return unet_zoo_amd.create_model(name, in_channels=3, num_classes=1, **kw)


@pytest.mark.parametrize("name,kw,H,W,okw", CASES, ids=[c[0] for c in CASES])
def test_engine_is_as_close_to_the_oracle_as_another_correct_implementation(name, kw, H, W, okw):
    torch.manual_seed(0)
    m = _build(name, kw)
    m.run_dtype = torch.bfloat16
    for mod in m.modules():            # dropout draws differ between the engine (device generator) and the oracle
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
    if hasattr(m, "sdpa"):
        m.sdpa.dropout.p = 0.0
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(DEV).train()
    x, mask = torch_ref.synthetic_batch(2, 3, H, W, seed=5)
    out = m(x.to(DEV))
    torch_ref.model_loss(out, mask.to(DEV)).backward()
    runs = {}
    for tag, jit in (("O", 0.0), ("J", 1e-6)):
        torch_ref.set_storage_rounding(torch.bfloat16, jitter=jit, seed=1)
        try:
            runs[tag] = torch_ref.train_step_reference(name, sd, x, mask, **okw)
        finally:
            torch_ref.set_storage_rounding(None)
    (ol, _, og, _), (jl, _, jg, _) = runs["O"], runs["J"]
    ol, jl, el = _first(ol), _first(jl), _first(out).detach().cpu().float()
    named = dict(m.named_parameters())
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in og.values())).item()
    # not the analytically-zero ones; multiresunet's parameters are stored channel-padded on the engine side -- a concatenated
    # channel axis keeps each part's padding, so the logical channels are NOT the leading block (round 4: the leading-block
    # slice this test used until then compared the wrong input channels for every layer that reads a concat, and that, not
    # the engine, was multiresunet's "2.4 x the yardstick")
    def eng_grad(n):
        return _logical_grad(m, named, n, og[n])

    keep = [n for n, g in og.items() if g.norm() > 2e-3 * total and n in named and named[n].grad is not None]
    e = torch.cat([eng_grad(n).flatten().cpu().double() for n in keep])
    o = torch.cat([og[n].flatten().double() for n in keep])
    j = torch.cat([jg[n].flatten().double() for n in keep])
    d_e, d_j = ((el - ol).norm() / ol.norm()).item(), ((jl - ol).norm() / ol.norm()).item()
    c_e, c_j = F.cosine_similarity(e, o, dim=0).item(), F.cosine_similarity(j, o, dim=0).item()
    print(f"{name}: logits rms engine-oracle {d_e:.4f} jitter-oracle {d_j:.4f}; gradient cosine engine {c_e:.4f} jitter {c_j:.4f}; "
          f"norm ratio {(e.norm() / o.norm()).item():.4f}")
    assert torch.isfinite(el).all() and torch.isfinite(e).all()
    # within three times the distance between two correct implementations (+ a floor of one bf16 rounding of the logits)
    assert d_e <= 3 * d_j + 4e-3, (d_e, d_j)
    assert 1 - c_e <= 3 * (1 - c_j) + 5e-3, (c_e, c_j)
    assert abs((e.norm() / o.norm()).item() - 1) <= 3 * abs((j.norm() / o.norm()).item() - 1) + 0.05
