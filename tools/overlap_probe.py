"""Do a matrix-core-bound kernel and an HBM-bound element pass run faster side by side on two HIP streams than back to
back on one?  (Round 5: the unet step is the serial sum of ~4.1 ms of MFMA-bound and ~1.6 ms of HBM-bound kernels; in the
backward the weight gradient of layer N and the BatchNorm-backward apply of layer N-1 are independent.)
   python tools/overlap_probe.py
For each pair: A alone, B alone, A;B on one stream, A || B on two streams (events on both, max of the two ends).
Microseconds per iteration, median of the rounds."""
import os
import statistics
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
sys.path.insert(0, _ROOT)
from unet_zoo_amd import _lib as L, ops

DEV = "cuda"
dt = torch.bfloat16
B = 16


def make(hw, c):
    x = ops.new_act(B, hw, hw, c, dt, DEV); x.buf.normal_()
    a = ops.new_act(B, hw, hw, c, dt, DEV)
    y = ops.new_act(B, hw, hw, c, dt, DEV)
    dy = ops.new_act(B, hw, hw, c, dt, DEV); dy.buf.normal_()
    w = torch.randn(c, c, 3, 3, device=DEV) * 0.05
    wp = ops.pack_weights(w, L.PACK_CONV_FWD, dt)
    sc, sh = torch.rand(c, device=DEV) + 0.5, torch.randn(c, device=DEV) * 0.3
    dw = torch.empty(c, c, 3, 3, device=DEV)
    x2 = ops.new_act(B, hw, hw, c, dt, DEV); x2.buf.normal_()
    a2 = ops.new_act(B, hw, hw, c, dt, DEV)
    vec = torch.stack([sc, sh, torch.zeros(c, device=DEV), torch.ones(c, device=DEV)])
    sums = torch.zeros(2, c, dtype=torch.float64, device=DEV)
    dg, db = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
    dy2 = ops.new_act(B, hw, hw, c, dt, DEV)
    parts = torch.zeros(8, 2, c, device=DEV)         # as a convolution's epilogue leaves them: finalize + the apply pass only
    return {
        "conv": lambda: ops.conv_igemm(a, wp, None, y, ntaps=9, want_stats=True),
        "wgrad": lambda: ops.wgrad(dy, x, (c, c, 3, 3), ntaps=9, out=dw),
        "apply": lambda: ops.bn_relu_apply(x2, sc, sh, a2),
        "bnbwd": lambda: ops.bn_relu_bwd(x2, vec, dy, None, None, sums, dy2, dg, db, partials=parts),
    }


def run(fa, fb, mode, n=10):
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    e0 = torch.cuda.Event(enable_timing=True)
    e1, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        e0.record()
    s2.wait_event(e0)
    if mode == "a":
        with torch.cuda.stream(s1):
            for _ in range(n):
                fa()
            e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    if mode == "b":
        with torch.cuda.stream(s1):
            for _ in range(n):
                fb()
            e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    if mode == "serial":
        with torch.cuda.stream(s1):
            for _ in range(n):
                fa(); fb()
            e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    with torch.cuda.stream(s1):
        for _ in range(n):
            fa()
        e1.record()
    with torch.cuda.stream(s2):
        for _ in range(n):
            fb()
        e2.record()
    torch.cuda.synchronize()
    return max(e0.elapsed_time(e1), e0.elapsed_time(e2)) / n * 1e3


def main():
    for hw, c in ((256, 64), (128, 128), (64, 256)):
        f = make(hw, c)
        for f_ in f.values():
            f_()
        torch.cuda.synchronize()
        for na, nb in (("wgrad", "apply"), ("wgrad", "bnbwd"), ("conv", "apply"), ("conv", "bnbwd"), ("conv", "wgrad")):
            res = {m: [] for m in ("a", "b", "serial", "par")}
            for _ in range(5):
                for m in res:
                    res[m].append(run(f[na], f[nb], m))
            med = {m: statistics.median(v) for m, v in res.items()}
            print(f"{c:4d}ch @{hw:3d}  {na:5s} {med['a']:7.1f}  {nb:5s} {med['b']:7.1f}  one stream {med['serial']:7.1f}  "
                  f"two streams {med['par']:7.1f}  ({med['par'] / med['serial']:.2f} of serial)", flush=True)


if __name__ == "__main__":
    main()
