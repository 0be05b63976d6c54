"""The BatchNorm + ReLU backward of unet's four encoder outputs (gradient = skip part + unpooled part; reduce pass, finalize,
apply pass) and the plain form beside it, microseconds per launch group (B = 16 256x256):
   python tools/bnbwd_bench.py            (UNET_ZOO_AMD_LIB=... for a variant build)"""
import os
import statistics
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
sys.path.insert(0, _ROOT)
from unet_zoo_amd import ops

DEV, dt, B = "cuda", torch.bfloat16, 16


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    for hw, c in ((256, 64), (128, 128), (64, 256), (32, 512)):
        y = ops.new_act(B, hw, hw, c, dt, DEV); y.buf.normal_()
        g0 = ops.new_act(B, hw, hw, c, dt, DEV); g0.buf.normal_()
        gp = ops.new_act(B, hw // 2, hw // 2, c, dt, DEV); gp.buf.normal_()
        dy = ops.new_act(B, hw, hw, c, dt, DEV)
        vec = torch.stack([torch.rand(c, device=DEV) + 0.5, torch.randn(c, device=DEV) * 0.3, torch.zeros(c, device=DEV),
                           torch.ones(c, device=DEV)])
        sums = torch.zeros(2, c, dtype=torch.float64, device=DEV)
        dg, db = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
        fns = {"pool": lambda: ops.bn_relu_bwd(y, vec, g0, None, gp, sums, dy, dg, db),
               "plain": lambda: ops.bn_relu_bwd(y, vec, g0, None, None, sums, dy, dg, db)}
        t = {k: [] for k in fns}
        for _ in range(7):
            for k, f in fns.items():
                t[k].append(timeit(f))
        mb = 2 * B * hw * hw * c / 1e6
        print(f"{c:4d}ch @{hw:3d}: " + "  ".join(f"{k} {statistics.median(v):7.1f} us" for k, v in t.items())
              + f"   (tensor {mb:.0f} MB; pool form moves {mb * (2 + 2.25 + 1.25):.0f} MB, plain {mb * 5:.0f} MB)", flush=True)


if __name__ == "__main__":
    main()
