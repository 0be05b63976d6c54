"""Where a model's bf16 gradient leaves the oracle: per parameter, the engine (E), the storage-rounded oracle (O) and the
same oracle with fp32-rounding-size jitter in front of every storage rounding (J), as tests/test_bf16_yardstick_gpu.py
builds them.  For every parameter |E - O| / |O|, |J - O| / |O| and the two cosines, in forward order; several jitter seeds
show how far two correct implementations scatter.
Usage: python tests/debug/yardstick_by_param.py MODEL H W [seeds]"""
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, ".")
import unet_zoo_amd
from oracle import torch_ref

name, H, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
seeds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
torch.manual_seed(0)
m = unet_zoo_amd.create_model(name, in_channels=3, num_classes=1)
m.run_dtype = torch.bfloat16
for mod in m.modules():
    if isinstance(mod, nn.Dropout):
        mod.p = 0.0
sd = {k: v.clone() for k, v in m.state_dict().items()}
m = m.cuda().train()
x, mask = torch_ref.synthetic_batch(2, 3, H, W, seed=5)
out = m(x.cuda())
torch_ref.model_loss(out, mask.cuda()).backward()


def oracle(jit, seed):
    torch_ref.set_storage_rounding(torch.bfloat16, jitter=jit, seed=seed)
    try:
        return torch_ref.train_step_reference(name, sd, x, mask)
    finally:
        torch_ref.set_storage_rounding(None)


_, _, og, _ = oracle(0.0, 1)
js = [oracle(1e-6, s)[2] for s in range(1, seeds + 1)]
named = dict(m.named_parameters())


def eng(n):
    g, want = named[n].grad, og[n]
    if g is not None and g.shape != want.shape:      # channel-padded storage on the engine side: the model's own index maps
        mod = m.get_submodule(n.rsplit(".", 1)[0])
        g = g.detach()
        for dim, idx in getattr(mod, "_maps", {}).get(n.rsplit(".", 1)[1], ()):
            g = g.index_select(dim, torch.as_tensor(idx, device=g.device))
    return None if g is None else g.cpu().double()


total = torch.sqrt(sum((g.double() ** 2).sum() for g in og.values())).item()
print(f"# {name} {H}x{W} bf16; |O| total {total:.4e}; columns: share of |O|^2, |E-O|/|O|, |J-O|/|O| (min..max over {seeds} seeds), cos(E,O), cos(J,O) min")
num_e = num_j = den = 0.0
for n, o in og.items():
    e = eng(n)
    if e is None or o.norm() < 2e-3 * total:
        continue
    o = o.double()
    de = ((e - o).norm() / o.norm()).item()
    dj = [((j[n].double() - o).norm() / o.norm()).item() for j in js]
    ce = F.cosine_similarity(e.flatten(), o.flatten(), dim=0).item()
    cj = min(F.cosine_similarity(j[n].double().flatten(), o.flatten(), dim=0).item() for j in js)
    share = (o.norm().item() / total) ** 2
    flag = "  <-- engine further than every jittered oracle" if de > 1.5 * max(dj) else ""
    print(f"{n:58s} {share:6.3f}  E {de:6.3f}  J {min(dj):6.3f}..{max(dj):6.3f}  cos {ce:6.3f} / {cj:6.3f}{flag}")
