"""Every 3x3 layer of the unet step (B = 16 3x256x256, bf16) against BOTH roofs: forward (with BatchNorm statistics), input
gradient and weight gradient (main kernel + slab reduction) through the C ABI, microseconds per call (10 calls between events,
best of 3), and for each: TFLOP/s and its fraction of the 2.5 PFLOP/s dense bf16 peak; algorithmic bytes (every
operand once: input + output, or dy + x + dW) and their fraction of 8 TB/s; the fraction of whichever roof is the tighter
one for the layer (floor = max(flops / peak, bytes / peak bandwidth)).
   python tools/layer_roofline.py"""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
sys.path.insert(0, _ROOT)
from unet_zoo_amd import _lib as L, ops

DEV, dt, B = "cuda", torch.bfloat16, 16
PEAK_TF, PEAK_GBS = 2500.0, 8000.0
# (layer of unet.py, H = W, Cin, Cout) -- encoder convs, bottleneck, decoder convs (the first conv 3 -> 64 has kernels of its own)
LAYERS = [("down1.conv2", 256, 64, 64), ("down2.conv1", 128, 64, 128), ("down2.conv2", 128, 128, 128), ("down3.conv1", 64, 128, 256),
          ("down3.conv2", 64, 256, 256), ("down4.conv1", 32, 256, 512), ("down4.conv2", 32, 512, 512), ("bottle.conv1", 16, 512, 1024),
          ("bottle.conv2", 16, 1024, 1024), ("up1.conv1", 32, 1024, 512), ("up1.conv2", 32, 512, 512), ("up2.conv1", 64, 512, 256),
          ("up2.conv2", 64, 256, 256), ("up3.conv1", 128, 256, 128), ("up3.conv2", 128, 128, 128), ("up4.conv1", 256, 128, 64),
          ("up4.conv2", 256, 64, 64)]


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


def row(name, us, gflop, mb):
    tf, gbs = gflop / us * 1e3, mb / us * 1e3
    floor = max(gflop / PEAK_TF, mb / PEAK_GBS) * 1e3     # us (GFLOP / (TFLOP/s) and MB / (GB/s) are milliseconds)
    bound = "mfma" if gflop / PEAK_TF > mb / PEAK_GBS else "hbm"
    return (f"  {name:6s} {us:7.1f} us  {tf:6.0f} TFLOP/s = {tf / PEAK_TF:4.2f}   {mb:6.1f} MB  {gbs:5.0f} GB/s = {gbs / PEAK_GBS:4.2f}"
            f"   tighter roof: {bound:4s} floor {floor:5.1f} us -> {floor / us:4.2f}")


def main():
    tot = {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0], "wgrad": [0.0, 0.0]}
    for name, hw, ci, co in LAYERS:
        P = B * hw * hw
        gflop = 2.0 * P * 9 * ci * co / 1e9
        x = ops.new_act(B, hw, hw, ci, dt, DEV); x.buf.normal_()
        y = ops.new_act(B, hw, hw, co, dt, DEV)
        g = ops.new_act(B, hw, hw, co, dt, DEV); g.buf.normal_()
        dx = ops.new_act(B, hw, hw, ci, dt, DEV)
        w = torch.randn(co, ci, 3, 3, device=DEV) * 0.05
        wp, wd = ops.pack_weights(w, L.PACK_CONV_FWD, dt), ops.pack_weights(w, L.PACK_CONV_DGRAD, dt)
        dw = torch.empty(co, ci, 3, 3, device=DEV)
        act_mb = 2.0 * P * (ci + co) / 1e6 + 2.0 * 9 * ci * co / 1e6          # input + output + packed weights
        wg_mb = 2.0 * P * (ci + co) / 1e6 + 4.0 * 9 * ci * co / 1e6           # dy + x + fp32 dW
        d = L.ConvDesc(L.dtype_code(dt), B, hw, hw, hw, hw, ci, ci, co, co, 9, L.TAPS_CONV, 1, L.STORE_PLAIN, 0, 0, 0)
        print(f"{name:13s} {ci:4d} -> {co:<4d} @ {hw:3d}x{hw:<3d}  {gflop:6.1f} GFLOP   [{ops.conv_kernel_name(d)}]", flush=True)
        for kind, fn, mb in (("fwd", lambda: ops.conv_igemm(x, wp, None, y, ntaps=9, want_stats=True), act_mb),
                             ("dgrad", lambda: ops.conv_igemm(g, wd, None, dx, ntaps=9), act_mb),
                             ("wgrad", lambda: ops.wgrad(g, x, (co, ci, 3, 3), ntaps=9, out=dw), wg_mb)):
            us = timeit(fn)
            floor = max(gflop / PEAK_TF, mb / PEAK_GBS) * 1e3
            tot[kind][0] += us
            tot[kind][1] += floor
            print(row(kind, us, gflop, mb), flush=True)
    print("\nsum over the 17 layers (eager launches, ~5 % above the same kernels inside the graph):")
    for kind, (us, floor) in tot.items():
        print(f"  {kind:6s} {us:8.1f} us   sum of the layers' tighter-roof floors {floor:7.1f} us -> {floor / us:4.2f}")


if __name__ == "__main__":
    main()
