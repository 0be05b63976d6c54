"""per-parameter difference of a bf16 backward with / without the deferred one-tap weight gradients (Engine.defer_linear_wgrads)"""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import unet_zoo_amd
from unet_zoo_amd.engine import Engine

name = sys.argv[1] if len(sys.argv) > 1 else "uctransnet"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 224
x = torch.randn(2, 3, size, size, generator=torch.Generator().manual_seed(1)).cuda()
grads = []
for defer in (True, False, False):
    Engine.defer_linear_wgrads = defer
    torch.manual_seed(0)
    kw = {"image_size": size} if name in ("uctransnet", "swin_unet_v2") else {}
    m = unet_zoo_amd.create_model(name, in_channels=3, num_classes=1, **kw)
    m.run_dtype = torch.bfloat16
    m = m.cuda().train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    out = m(x)
    out = out[0] if isinstance(out, (list, tuple)) else out
    out.float().mean().backward()
    grads.append({n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
for n in grads[0]:
    a, b, c = grads[0][n].double(), grads[1][n].double(), grads[2][n].double()
    e1 = ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()
    e0 = ((c - b).abs().max() / (b.abs().max() + 1e-12)).item()
    if e1 > 1e-3 or e0 > 1e-3:
        print(f"{n:60s} {tuple(grads[0][n].shape)} deferred-vs-direct {e1:.3e}   direct-vs-direct {e0:.3e}")
print("done", len(grads[0]))
