"""The vendor library on the SAME shapes: torch's conv2d (MIOpen behind it: bf16, channels_last) forward, input gradient and
weight gradient at the 3x3 layers of unet's levels (B = 16 3x256x256), beside this library's kernels for the same layers.
   python tools/miopen_yardstick.py
Microseconds per call (10 calls between events after a warm-up that includes MIOpen's search), TFLOP/s in brackets."""
import os
import statistics
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
import torch.nn.functional as F
sys.path.insert(0, _ROOT)
from unet_zoo_amd import _lib as L, ops

DEV, dt, B = "cuda", torch.bfloat16, 16
LAYERS = [(256, 64, 64), (128, 64, 128), (128, 128, 128), (64, 256, 256), (32, 512, 512), (16, 1024, 1024), (256, 128, 64)]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return statistics.median(ts)


def main():
    torch.backends.cudnn.benchmark = True      # MIOpen: search for the fastest solver per shape
    print("layer (B=16)                 |  vendor fwd   dgrad    wgrad   |  here  fwd    dgrad    wgrad   (us; TFLOP/s of the faster fwd)")
    for hw, ci, co in LAYERS:
        gf = 2.0 * B * hw * hw * ci * co * 9 / 1e12
        x = torch.randn(B, ci, hw, hw, device=DEV, dtype=dt).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        w = (torch.randn(co, ci, 3, 3, device=DEV, dtype=dt) * 0.05).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        y = F.conv2d(x, w, padding=1)
        gy = torch.randn_like(y)
        v_f = timeit(lambda: F.conv2d(x, w, padding=1))
        v_d = timeit(lambda: torch.autograd.grad(y, x, gy, retain_graph=True))
        v_w = timeit(lambda: torch.autograd.grad(y, w, gy, retain_graph=True))
        # this library: NHWC activations, packed weights (packing is per step, not per call)
        xa = ops.new_act(B, hw, hw, ci, dt, DEV); xa.buf.normal_()
        ya = ops.new_act(B, hw, hw, co, dt, DEV)
        ga = ops.new_act(B, hw, hw, co, dt, DEV); ga.buf.normal_()
        dxa = ops.new_act(B, hw, hw, ci, dt, DEV)
        wf = torch.randn(co, ci, 3, 3, device=DEV) * 0.05
        wp = ops.pack_weights(wf, L.PACK_CONV_FWD, dt)
        wd = ops.pack_weights(wf, L.PACK_CONV_DGRAD, dt)
        dw = torch.empty(co, ci, 3, 3, device=DEV)
        h_f = timeit(lambda: ops.conv_igemm(xa, wp, None, ya, ntaps=9, want_stats=True))
        h_d = timeit(lambda: ops.conv_igemm(ga, wd, None, dxa, ntaps=9))
        h_w = timeit(lambda: ops.wgrad(ga, xa, (co, ci, 3, 3), ntaps=9, out=dw))
        # the plain matrix product of the same size through the vendor's GEMM library (hipBLASLt / rocBLAS behind
        # torch.matmul): [P, 9 Cin] x [9 Cin, Cout] with the im2col matrix GIVEN -- what the matrix cores deliver on this
        # arithmetic shape on this box when nothing but a GEMM has to be done (no halo gather, no statistics)
        P = B * hw * hw
        am = torch.randn(P, 9 * ci, device=DEV, dtype=dt)
        bm = torch.randn(9 * ci, co, device=DEV, dtype=dt)
        g_f = timeit(lambda: torch.matmul(am, bm))
        del am, bm
        print(f"{ci:4d} -> {co:<4d} @ {hw:3d}x{hw:<3d} {gf * 1e3:5.0f} GF | {v_f:8.1f} {v_d:8.1f} {v_w:8.1f}  | {h_f:8.1f} {h_d:8.1f} {h_w:8.1f}"
              f"   (vendor {gf / v_f * 1e6:6.0f}, here {gf / h_f * 1e6:6.0f} TFLOP/s; vendor GEMM on the given im2col matrix"
              f" {g_f:6.1f} us = {gf / g_f * 1e6:6.0f} TFLOP/s)", flush=True)


if __name__ == "__main__":
    main()
