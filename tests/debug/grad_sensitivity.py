"""fp32 gradient of one model under input perturbations of 1e-7 ... 1e-5: how much of a golden-vector gradient mismatch is
the network amplifying rounding noise (ReLU masks, max-pool selections) rather than a kernel.  usage: python tests/debug/grad_sensitivity.py"""
import torch, torch.nn.functional as F, sys
sys.path.insert(0, ".")
import unet_zoo_amd
from oracle import torch_ref
x, mask = torch_ref.synthetic_batch(2, 3, 64, 64, seed=1)
def run(xx):
    torch.manual_seed(0)
    m = unet_zoo_amd.create_model("unet_transformer", in_channels=3, num_classes=1)
    m.run_dtype = torch.float32
    m = m.to("cuda").train()
    lg = m(xx.to("cuda"))
    loss = F.binary_cross_entropy_with_logits(lg, mask.to("cuda"))
    loss.backward()
    g = torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None])
    return lg.detach(), g
l0, g0 = run(x)
l0b, g0b = run(x)
print("repeat: logits", (l0 - l0b).abs().max().item(), "grad", (g0 - g0b).abs().max().item())
for eps in (1e-7, 1e-6, 1e-5):
    torch.manual_seed(123)
    xp = x * (1 + eps * torch.randn_like(x))
    l1, g1 = run(xp)
    print("eps", eps, "logits rel", ((l1 - l0).abs().max() / l0.abs().max()).item(), "grad norm", g0.norm().item(), g1.norm().item(),
          "rel diff of norms", abs(g1.norm().item() - g0.norm().item()) / g0.norm().item(), "rel L2", ((g1 - g0).norm() / g0.norm()).item())
