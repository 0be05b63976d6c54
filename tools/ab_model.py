"""Same-box, same-process A/B of a whole GraphedStep between kernel generations (ablation library: UZ_TUNE is read when
the plans are made, i.e. at capture).   python tools/ab_model.py MODEL SIZE BATCH [--tune=0,33554432] [--steps=10]"""
import os
import sys
import time


def _synthetic_batch(B, C, H, W, seed=1):
    """randn image, rand > 0.5 mask, one generator (the fixture protocol; tools/ do not import the oracle)"""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, C, H, W, generator=g), (torch.rand(B, 1, H, W, generator=g) > 0.5).float()


_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("UNET_ZOO_AMD_LIB", os.path.join(_ROOT, "unet_zoo_amd", "libunetzoo_hip_ablate.so"))
import torch
sys.path.insert(0, _ROOT)
import unet_zoo_amd
from unet_zoo_amd import ops


def main():
    name, size, batch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    tunes, steps = ["0", str(0x2000000)], 10
    for a in sys.argv[4:]:
        if a.startswith("--tune="):
            tunes = a.split("=")[1].split(",")
        if a.startswith("--steps="):
            steps = int(a.split("=")[1])
    dev = torch.device("cuda", 0)
    x, mask = _synthetic_batch(batch, 3, size, size, seed=5)
    x, mask = x.to(dev), mask.to(dev)
    res = {}
    for rep in range(2):
        for t in tunes:
            os.environ["UZ_TUNE"] = t
            torch.manual_seed(0)
            kw = {}
            if name == "swin_unet_v2":
                kw = dict(image_size=size, window_size=7 if size % 7 == 0 else 8, drop_path_rate=0.0)
            if name == "attention_unet":
                kw = dict(depth=5)
            m = unet_zoo_amd.create_model(name, in_channels=3, num_classes=1, **kw)
            m.run_dtype = torch.bfloat16
            m = m.to(dev).train()
            gs = unet_zoo_amd.GraphedStep(m, "bce_dice", lr=1e-4, weight_decay=1e-5)
            for _ in range(3):
                gs(x, mask)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                gs(x, mask)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            res.setdefault(t, []).append((ms, float(gs.loss)))
            del gs, m
            torch.cuda.empty_cache()
    for t in tunes:
        print(f"{name} B={batch} {size}^2 UZ_TUNE={t}: {min(r[0] for r in res[t]):8.3f} ms/step  loss {res[t][0][1]:.5f}", flush=True)


main()
