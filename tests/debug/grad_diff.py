"""Debug helper: per-parameter gradient error of the HIP path vs the CPU oracle."""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
import unet_zoo_amd
from oracle import torch_ref

H = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dt = torch.float32 if (len(sys.argv) < 3 or sys.argv[2] == "fp32") else torch.bfloat16
torch.manual_seed(0)
m = unet_zoo_amd.create_model("unet")
m.run_dtype = dt
sd = {k: v.clone() for k, v in m.state_dict().items()}
m = m.cuda().train()
x, mask = torch_ref.synthetic_batch(2, 3, H, H, seed=1)
logits = m(x.cuda())
F.binary_cross_entropy_with_logits(logits, mask.cuda()).backward()
rl, rloss, rg, st = torch_ref.train_step_reference("unet", sd, x, mask)
print("logits rel", ((logits.detach().cpu() - rl).abs().max() / rl.abs().max()).item())
for n, p in m.named_parameters():
    g, r = p.grad.cpu().double(), rg[n].double()
    print(f"{n:50s} l2rel {((g - r).norm() / (r.norm() + 1e-30)).item():.3e} maxrel {((g - r).abs().max() / (r.abs().max() + 1e-30)).item():.3e} refnorm {r.norm().item():.3e}")
