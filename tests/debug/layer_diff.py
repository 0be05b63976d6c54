"""Debug tool (GPU): per-layer divergence of a model's HIP-engine forward from the oracle (every Conv + BN + ReLU
output), optionally with the oracle rounding to bf16 at the engine's storage points.
Usage: python tests/debug/layer_diff.py MODEL H W [bf16|fp32] [key=value ...]   (values are eval'd: res=(16,24))"""
import sys

import torch

sys.path.insert(0, ".")
import unet_zoo_amd
from oracle import torch_ref
from unet_zoo_amd.engine import Engine

name, H, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dtype = torch.float32 if (len(sys.argv) > 4 and sys.argv[4] == "fp32") else torch.bfloat16
kw = {a.split("=")[0]: eval(a.split("=", 1)[1]) for a in sys.argv[5:]}
okw = {"res": kw["common_attn_res_for_QK_V"]} if "common_attn_res_for_QK_V" in kw else {}
torch.manual_seed(0)
m = unet_zoo_amd.create_model(name, in_channels=3, num_classes=1, **kw)
m.run_dtype = dtype
sd0 = {k: v.clone() for k, v in m.state_dict().items()}
m = m.cuda().train()
x, mask = torch_ref.synthetic_batch(2, 3, H, W, seed=5)

conv_name = {id(mod): n for n, mod in m.named_modules()}
got, want = {}, {}
orig = Engine.conv_bn_relu


def rec(self, x_, conv, bn, **k2):
    act, pooled = orig(self, x_, conv, bn, **k2)
    got[conv_name[id(conv)]] = act.dense().cpu()
    return act, pooled


Engine.conv_bn_relu = rec
with torch.no_grad():
    out = m(x.cuda())
Engine.conv_bn_relu = orig

o2, o3 = torch_ref.conv_bn_relu, getattr(torch_ref, "_conv1x1_bn_relu", None)


def rec2(x_, sd, conv, bn, training, *a, **k2):
    y = o2(x_, sd, conv, bn, training, *a, **k2)
    want[conv] = y.detach()
    return y


def rec3(x_, sd, conv, bn, training):
    y = o3(x_, sd, conv, bn, training)
    want[conv] = y.detach()
    return y


torch_ref.conv_bn_relu = rec2
if o3 is not None:
    torch_ref._conv1x1_bn_relu = rec3
torch_ref.set_storage_rounding(None if dtype == torch.float32 else torch.bfloat16)
with torch.no_grad():
    ref = torch_ref.FORWARDS[name](torch_ref.clone_state(sd0), x, True, **okw)
torch_ref.set_storage_rounding(None)
for k in want:
    if k not in got:
        print(f"{k:40s} (not an engine conv_bn_relu)")
        continue
    a, b = got[k], want[k]
    err = ((a - b).abs().max() / b.abs().max()).item()
    rms = ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()
    print(f"{k:40s} {tuple(b.shape)!s:22s} max {err:9.2e} rms {rms:9.2e}")
if torch.is_tensor(ref):
    a = out.cpu()
    print("logits max", ((a - ref).abs().max() / ref.abs().max()).item(), "rms", ((a - ref).norm() / ref.norm()).item())
