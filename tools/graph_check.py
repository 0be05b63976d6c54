"""Debug: hipGraph replay vs eager for one model (loss trajectory, gradient equality)."""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
import unet_zoo_amd
from bench import make_model, model_loss

name = sys.argv[1] if len(sys.argv) > 1 else "swin_unet_v2"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 128
torch.manual_seed(0)
m, _ = make_model(name, size)
m = m.cuda().train()
g = torch.Generator().manual_seed(1)
x = torch.randn(4, 3, size, size, generator=g).cuda()
mask = (torch.rand(4, 1, size, size, generator=g) > 0.5).float().cuda()
params = list(m.parameters())
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        for p in params:
            p.grad = None
        loss = model_loss(m(x), mask)
        loss.backward()
        print("eager loss", loss.item())
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
eager = {i: p.grad.clone() for i, p in enumerate(params) if p.grad is not None}
used = [p for p in params if p.grad is not None]
for p in params:
    p.grad = None
flat = torch.zeros(sum(p.numel() for p in used), device="cuda")
off = 0
for p in used:
    p.grad = flat[off:off + p.numel()].view_as(p)
    off += p.numel()
m.grads_in_place = True
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    out = m(x)
    sl = model_loss(out, mask)
    sl.backward()
for i in range(3):
    gr.replay()
    torch.cuda.synchronize()
    print("graph loss", sl.item(), "out finite", bool(torch.isfinite(out if not isinstance(out, dict) else out["main"]).all()))
bad = 0
for i, p in enumerate(params):
    if i in eager:
        d = (p.grad - eager[i]).abs().max().item()
        if d > 1e-2 * eager[i].abs().max().item() + 1e-6:
            bad += 1
print("params with differing grads:", bad, "of", len(eager))
opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-5, fused=True, capturable=True)
go = torch.cuda.CUDAGraph()
with torch.cuda.graph(go):
    torch.nn.utils.clip_grad_norm_(params, 1.0, foreach=True)
    opt.step()
for i in range(6):
    gr.replay()
    torch.cuda.synchronize()
    nf = [n for n, p in m.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
    big = sorted(((p.grad.abs().max().item(), n) for n, p in m.named_parameters() if p.grad is not None), reverse=True)[:4]
    print("pre-opt grad norm", flat.norm().item(), "non-finite grads:", nf[:6], "largest:", big)
    go.replay()
    torch.cuda.synchronize()
    print("train loss", sl.item(), "grad norm", flat.norm().item(), "param finite",
          all(bool(torch.isfinite(p).all()) for p in params))
