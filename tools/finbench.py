"""BatchNorm finalize + apply as two launches against the one launch with the finalize inside (uz_bn_relu_add_apply_fin), and
the backward pair likewise, at unet's level shapes (B = 16 256x256); microseconds per layer, median of the rounds.
   python tools/finbench.py            (UNET_ZOO_AMD_LIB=... for a variant build)"""
import os
import statistics
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
sys.path.insert(0, _ROOT)
from unet_zoo_amd import ops

DEV, dt, B, N_IT = "cuda", torch.bfloat16, 16, 20


def timed(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(N_IT):
        fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N_IT * 1e3


def main():
    for hw, c, rows in ((256, 64, 256), (128, 128, 256), (64, 256, 256), (32, 512, 256), (16, 1024, 128)):
        y = ops.new_act(B, hw, hw, c, dt, DEV); y.buf.normal_()
        act = ops.new_act(B, hw, hw, c, dt, DEV)
        g0 = ops.new_act(B, hw, hw, c, dt, DEV); g0.buf.normal_()
        dy = ops.new_act(B, hw, hw, c, dt, DEV)
        stats = torch.randn(rows, 2, c, device=DEV); stats[:, 1] = stats[:, 1].abs() * 40 + 60
        parts = torch.randn(rows, 2, c, device=DEV)
        gamma, beta = torch.rand(c, device=DEV) + 0.5, torch.randn(c, device=DEV)
        rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
        sums = torch.zeros(2, c, dtype=torch.float64, device=DEV)
        dg, db = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
        flags = torch.zeros(4 * N_IT, dtype=torch.int32, device=DEV)
        vec0 = ops.bn_finalize(stats, y.P, gamma, beta, 1e-5, 0.1, rm, rv)

        def two(i):
            v = ops.bn_finalize(stats, y.P, gamma, beta, 1e-5, 0.1, rm, rv)
            ops.bn_relu_apply(y, v[0], v[1], act)

        def one(i):
            ops.bn_relu_apply_fin(y, stats, y.P, gamma, beta, 1e-5, 0.1, rm, rv, flags[i:i + 1], act)

        def alone(i):
            ops.bn_relu_apply(y, vec0[0], vec0[1], act)

        def btwo(i):
            ops.bn_relu_bwd(y, vec0, g0, None, None, sums, dy, dg, db, partials=parts)

        def bone(i):
            ops.bn_relu_bwd(y, vec0, g0, None, None, sums, dy, dg, db, partials=parts, fin_flag=flags[2 * N_IT + i:2 * N_IT + i + 1])

        res = {k: [] for k in ("apply", "fin+apply", "fused", "bwd fin+apply", "bwd fused")}
        for _ in range(5):
            res["apply"].append(timed(alone))
            res["fin+apply"].append(timed(two))
            flags.zero_()
            res["fused"].append(timed(one))
            res["bwd fin+apply"].append(timed(btwo))
            flags.zero_()
            res["bwd fused"].append(timed(bone))
        print(f"{c:5d}ch @{hw:3d}: " + "  ".join(f"{k} {statistics.median(v):6.1f}" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
