"""A/B timing of the graphed UNet step in ONE process on ONE box (boxes differ by +-10 % on the MFMA kernels, so
separate runs cannot resolve a 1 % change): every variant = (name, {Engine class attribute: value}), measured interleaved.
   python tools/ab_step.py [--model unet] [--rounds 3]"""
import argparse
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_zoo_amd
from unet_zoo_amd.engine import Engine

_ON = dict(BN_BWD_ALTERNATE=True, reverse_element_passes=True, fuse_bn_finalize=False, fold_bn_apply=True, fold_bn_apply_head=True,
           fuse_bn_reduce=True, fuse_bn_reduce_convt=True)
VARIANTS = [("shipped", dict(_ON)),
            ("head: apply pass kept", dict(_ON, fold_bn_apply_head=False)),
            ("bwd passes not alternating", dict(_ON, BN_BWD_ALTERNATE=False)),
            ("element passes ascending", dict(_ON, BN_BWD_ALTERNATE=False, reverse_element_passes=False)),
            ("finalize in consumer", dict(_ON, fuse_bn_finalize=True)),
            ("no fold of BN apply (xf)", dict(_ON, fold_bn_apply=False, fold_bn_apply_head=False)),
            ("fused conv only", dict(_ON, fold_bn_apply=False, fold_bn_apply_head=False, fuse_bn_reduce_convt=False)),
            ("two-pass everywhere", dict(_ON, fold_bn_apply=False, fold_bn_apply_head=False, fuse_bn_reduce=False, fuse_bn_reduce_convt=False))]


def build(attrs, model_name, B, S):
    for k, v in attrs.items():
        if k == "UZ_TUNE":       # plan switches of the ablation build (UNET_ZOO_AMD_LIB=.../libunetzoo_hip_ablate.so): the
            os.environ["UZ_TUNE"] = str(v)     # plans are taken at capture time, so a captured step keeps its variant
        elif k == "BN_BWD_ALTERNATE":
            from unet_zoo_amd import ops as _ops
            _ops.BN_BWD_ALTERNATE = v
        else:
            setattr(Engine, k, v)
    torch.manual_seed(0)
    import bench   # the benchmark's own create_model call (swin: image_size and a window that divides the token maps)
    model = bench.make_model(model_name, S)[0].cuda()
    model.run_dtype = torch.bfloat16
    step = unet_zoo_amd.GraphedStep(model, "bce_dice", lr=1e-4, weight_decay=1e-5, max_norm=1.0)
    x = torch.randn(B, 3, S, S, device="cuda")
    t = (torch.rand(B, 1, S, S, device="cuda") > 0.5).float()
    for _ in range(3):
        step(x, t)
    torch.cuda.synchronize()
    return step, x, t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="unet")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--tunes", default=None, help="comma list of UZ_TUNE values instead of the Engine variants (ablation build)")
    a = ap.parse_args()
    global VARIANTS
    if a.tunes:
        VARIANTS = [(f"UZ_TUNE={t}", {"UZ_TUNE": int(t)}) for t in a.tunes.split(",")]
    built = [(name, build(attrs, a.model, a.batch, a.size)) for name, attrs in VARIANTS]
    best = {name: 1e9 for name, _ in VARIANTS}
    for r in range(a.rounds):
        for name, (step, x, t) in built:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step(x, t)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / a.steps * 1e3
            best[name] = min(best[name], ms)
            print(f"round {r} {name:30s} {ms:7.3f} ms/step", flush=True)
    for name, ms in best.items():
        print(f"best  {name:30s} {ms:7.3f} ms/step")


main()
