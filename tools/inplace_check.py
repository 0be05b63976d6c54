import sys
import torch
sys.path.insert(0, ".")
from bench import make_model, model_loss
torch.manual_seed(0)
m, _ = make_model("swin_unet_v2", 128)
m = m.cuda().train()
g = torch.Generator().manual_seed(1)
x = torch.randn(4, 3, 128, 128, generator=g).cuda()
mask = (torch.rand(4, 1, 128, 128, generator=g) > 0.5).float().cuda()
model_loss(m(x), mask).backward()
used = [p for p in m.parameters() if p.grad is not None]
for p in m.parameters():
    p.grad = None
flat = torch.zeros(sum(p.numel() for p in used), device="cuda")
off = 0
for p in used:
    p.grad = flat[off:off + p.numel()].view_as(p)
    off += p.numel()
m.grads_in_place = True
for it in range(3):
    model_loss(m(x), mask).backward()
    torch.cuda.synchronize()
    big = sorted(((p.grad.abs().max().item(), n) for n, p in m.named_parameters() if p.grad is not None), reverse=True)[:3]
    print("in-place eager: norm", flat.norm().item(), big)
