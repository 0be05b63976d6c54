"""GPU, bf16 run mode against the oracle that ROUNDS TO bf16 WHERE THE ENGINE STORES bf16
(oracle/torch_ref.set_storage_rounding): what remains between the two is summation order and the rounding of values
that sit exactly between two bf16 numbers -- "the product rounds here" is taken out of the comparison, "a kernel
computes something else" is not.  Every Conv + BN + ReLU output of the network is compared, then the logits, the loss
and the gradients (round 1 compared bf16 with the fp32 reference and had to allow 6 % / cosine 0.75).

Round 3 adds the yardstick for the bounds that still look loose: the oracle against itself with jitter of fp32-rounding
size (1e-6) in front of every storage rounding (`set_storage_rounding(dtype, jitter=...)`), i.e. against another correct
bf16-storage implementation whose sums come out in a different order.  The engine must be within twice that distance."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import unet_zoo_amd
from oracle import torch_ref
from unet_zoo_amd.engine import Engine

DEV = "cuda"


def _run(name, H, W, seed=5, B=2):
    torch.manual_seed(0)
    m = unet_zoo_amd.create_model(name, in_channels=3, num_classes=1)
    m.run_dtype = torch.bfloat16
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(DEV).train()
    x, mask = torch_ref.synthetic_batch(B, 3, H, W, seed=seed)
    conv_name = {id(mod): n for n, mod in m.named_modules()}
    got, want = {}, {}
    orig = Engine.conv_bn_relu

    def rec(self, x_, conv, bn, **kw):
        act, pooled = orig(self, x_, conv, bn, **kw)
        got[conv_name[id(conv)]] = act.dense().cpu()
        return act, pooled

    o2 = torch_ref.conv_bn_relu

    def rec2(x_, sd_, conv, bn, training, *a, **kw):
        y = o2(x_, sd_, conv, bn, training, *a, **kw)
        want[conv] = y.detach()
        return y

    Engine.conv_bn_relu, torch_ref.conv_bn_relu = rec, rec2
    torch_ref.set_storage_rounding(torch.bfloat16)
    try:
        logits = m(x.to(DEV))
        loss = F.binary_cross_entropy_with_logits(logits, mask.to(DEV))
        loss.backward()
        rl, rloss, rgrads, _ = torch_ref.train_step_reference(name, sd, x, mask)
    finally:
        Engine.conv_bn_relu, torch_ref.conv_bn_relu = orig, o2
        torch_ref.set_storage_rounding(None)
    # the yardstick: the same rounded oracle with fp32-rounding-size jitter in front of every storage rounding --
    # "another correct implementation" (oracle/torch_ref.py) -- against the oracle itself
    torch_ref.set_storage_rounding(torch.bfloat16, jitter=1e-6, seed=1)
    try:
        jl, jloss, jgrads, _ = torch_ref.train_step_reference(name, sd, x, mask)
    finally:
        torch_ref.set_storage_rounding(None)
    layers = [(k, ((got[k] - want[k]).norm() / want[k].norm()).item()) for k in want]
    named = dict(m.named_parameters())
    per_param = {}
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in rgrads.values())).item()
    for n, g in rgrads.items():
        # parameters whose gradient is analytically zero (a per-channel constant in front of a train-mode BatchNorm:
        # conv biases, resunet's skip-branch beta and input_skip bias) hold rounding noise in the oracle -- bf16-sized under
        # storage rounding -- and (near) zeros here
        if g.norm() > 2e-3 * total:
            per_param[n] = F.cosine_similarity(named[n].grad.flatten().cpu(), g.flatten(), dim=0).item()
    a = torch.cat([named[n].grad.flatten().cpu() for n in per_param])
    b = torch.cat([rgrads[n].flatten() for n in per_param])
    jb = torch.cat([jgrads[n].flatten() for n in per_param])
    jit = dict(logits_rms=((jl - rl).norm() / rl.norm()).item(), cos=F.cosine_similarity(jb, b, dim=0).item(),
               worst=min(F.cosine_similarity(jgrads[n].flatten(), rgrads[n].flatten(), dim=0).item() for n in per_param))
    return dict(layers=layers, logits=logits.detach().cpu(), ref=rl, loss=loss.item(), rloss=rloss.item(),
                cos=F.cosine_similarity(a, b, dim=0).item(), per_param=per_param,
                gn=a.norm().item(), rgn=b.norm().item(), jit=jit)


# (model, H, W, rms bound of any layer, rms bound of the logits, global gradient cosine, per-parameter cosine)
CASES = [
    # UNet: the decoder re-reads the accurate high-resolution encoder features, the error shrinks towards the output
    # measured: worst layer 1.5e-2 / 1.8e-2 (the 4x4 bottleneck: BatchNorm over 32 samples), logits 4.8e-3 / 5.0e-3, loss
    # equal to 1e-5, gradient cosine 0.9981 / 0.9987, worst parameter 0.968 / 0.964, norm ratio 1.0004
    ("unet", 64, 64, 3e-2, 1e-2, 0.995, 0.92),
    ("unet", 64, 96, 3e-2, 1e-2, 0.995, 0.92),
    # Attention U-Net: BOTH inputs of every decoder convolution (the gated skip psi * x and the upsampled path) carry
    # the decoder's error, which therefore grows ~1.25x per layer instead of shrinking (tests/debug/layer_diff.py)
    # Measured: engine vs oracle logits rms 0.083, gradient cosine 0.83, worst parameter 0.76 -- while the oracle against
    # ITSELF with 1e-6 jitter in front of its storage roundings is 0.14 / 0.70 / 0.60: the network, not a kernel
    ("attention_unet", 64, 64, 0.15, 0.15, 0.75, 0.5),
    ("nested_unet", 64, 64, 4e-2, 2.5e-2, 0.95, 0.9),
    ("resunet", 64, 64, 3e-2, 1e-2, 0.995, 0.98),
]


@pytest.mark.parametrize("name,H,W,layer_rms,logit_rms,cos_all,cos_each", CASES, ids=[f"{c[0]}_{c[1]}x{c[2]}" for c in CASES])
def test_bf16_matches_the_storage_rounded_oracle(name, H, W, layer_rms, logit_rms, cos_all, cos_each):
    r = _run(name, H, W)
    layers = r["layers"]
    print(name, "oracle vs jittered oracle:", r["jit"])
    print(name, "worst layer", max(e for _, e in layers) if layers else None, "logits rms",
          ((r["logits"] - r["ref"]).norm() / r["ref"].norm()).item(), "loss", r["loss"], r["rloss"], "cos", r["cos"],
          "worst param cos", min(r["per_param"].values()), "norm ratio", r["gn"] / r["rgn"])
    if layers:
        assert layers[0][1] < 1e-4, layers[0]                   # same rounded operands, one layer of summation order
        for k, e in layers:
            assert e < layer_rms, (k, e)
    ref, got = r["ref"], r["logits"]
    assert (got - ref).norm() <= logit_rms * ref.norm()
    margin = 4 * logit_rms * ref.abs().max()
    sure = ref.abs() > margin
    assert torch.equal((got > 0)[sure], (ref > 0)[sure])        # masks agree outside the rounding margin
    assert abs(r["loss"] - r["rloss"]) < 5e-3 * max(1.0, logit_rms / 1.5e-2)
    assert r["cos"] > cos_all
    assert abs(r["gn"] / r["rgn"] - 1.0) < 0.1
    worst = min(r["per_param"].items(), key=lambda kv: kv[1])
    assert worst[1] > cos_each, worst
    # and, self-calibrating: the engine is no further from the oracle than twice what the oracle is from "another correct
    # implementation" of itself (fp32-rounding-size jitter in front of every bf16 storage rounding)
    j = r["jit"]
    assert (got - ref).norm() / ref.norm() <= 2 * j["logits_rms"] + 1e-3, j
    assert 1 - r["cos"] <= 2 * (1 - j["cos"]) + 2e-3, j
    assert 1 - worst[1] <= 2 * (1 - j["worst"]) + 2e-2, j


def test_bf16_unet_on_the_kernels_the_benchmark_times():
    """VERDICT r3 "weak" 1: at 2 x 64 x 64 the plan never selects the 512-pixel x 64-channel ping-pong configuration (the
    north-star 64 -> 64 layer needs >= 128 tiles) nor the row-walk weight gradient's long ranges.  unet at B = 4 128 x 128
    does (128 tiles of 512 pixels): same storage-rounded-oracle comparison, and the kernel families are asserted BY NAME
    from the library's own plan (ops.profile_*: uz_conv_igemm_kernel_name / uz_wgrad_kernel_name)."""
    from unet_zoo_amd import ops
    ops.profile_begin()
    try:
        r = _run("unet", 128, 128, B=4)
    finally:
        fams = set(ops.profile_end())
    print(sorted(fams))
    # (the 64 -> 64 convolution reads its input through the BatchNorm + ReLU in front of it since round 5: "_xf")
    assert "conv3x3_pp512x64_bf16_xf" in fams and "wgrad9_bf16_64x64_rowwalk_xf" in fams, fams
    assert "conv3x3_pp512x64_bf16" in fams, fams
    assert "conv3x3_pp512_bf16" in fams or "conv3x3_pp256_bf16" in fams, fams
    assert "wgrad9_bf16_64x64_rowwalk" in fams, fams
    layers = r["layers"]
    assert layers[0][1] < 1e-4, layers[0]
    for k, e in layers:
        assert e < 3e-2, (k, e)
    ref, got = r["ref"], r["logits"]
    assert (got - ref).norm() <= 1e-2 * ref.norm()
    assert abs(r["loss"] - r["rloss"]) < 5e-3
    assert r["cos"] > 0.995 and abs(r["gn"] / r["rgn"] - 1.0) < 0.05
    worst = min(r["per_param"].items(), key=lambda kv: kv[1])
    assert worst[1] > 0.9, worst
    j = r["jit"]
    assert (got - ref).norm() / ref.norm() <= 2 * j["logits_rms"] + 1e-3, j
    assert 1 - r["cos"] <= 2 * (1 - j["cos"]) + 2e-3, j
