"""Debug tool (GPU): per-layer divergence of the HIP engine's u2net forward from the oracle with the
same bf16 storage-rounding points.  Usage: python tests/debug/u2_layer_diff.py [u2net|u2netp] [size]"""
import sys

import torch

sys.path.insert(0, ".")
import unet_zoo_amd
from oracle import torch_ref
from unet_zoo_amd.engine import Engine

name = sys.argv[1] if len(sys.argv) > 1 else "u2net"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dtype = torch.bfloat16 if (len(sys.argv) <= 3 or sys.argv[3] == "bf16") else torch.float32
torch.manual_seed(3)
m = unet_zoo_amd.create_model(name)
m.run_dtype = dtype
sd0 = {k: v.clone() for k, v in m.state_dict().items()}
m = m.cuda().train()
x, mask = torch_ref.synthetic_batch(2, 3, size, size, seed=5)

conv_name = {id(mod): n for n, mod in m.named_modules()}
got = {}
orig = Engine.conv_bn_relu


def rec(self, x_, conv, bn, **kw):
    act, pooled = orig(self, x_, conv, bn, **kw)
    got[conv_name[id(conv)]] = act.dense().cpu()
    return act, pooled


Engine.conv_bn_relu = rec
with torch.no_grad():
    outs = m(x.cuda())
Engine.conv_bn_relu = orig

want = {}
o2 = torch_ref.conv_bn_relu


def rec2(x_, sd, conv, bn, training, dilation=1, residual=None):
    y = o2(x_, sd, conv, bn, training, dilation, residual)
    want[conv] = y.detach()
    return y


torch_ref.conv_bn_relu = rec2
torch_ref.set_storage_rounding(None if dtype == torch.float32 else torch.bfloat16)
with torch.no_grad():
    ref = torch_ref.u2net_forward(torch_ref.clone_state(sd0), x, True)
torch_ref.set_storage_rounding(None)

for k in want:
    a, b = got[k], want[k]
    err = ((a - b).abs().max() / b.abs().max()).item()
    rms = ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()
    print(f"{k:32s} {tuple(b.shape)!s:22s} max {err:9.2e} rms {rms:9.2e}")
for k in ref:
    a, b = outs[k].cpu(), ref[k]
    print(k, ((a - b).abs().max() / b.abs().max()).item())
