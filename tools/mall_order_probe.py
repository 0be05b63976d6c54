"""Does an element pass run faster when it walks its input from the END -- the part the producing kernel wrote last and the
256 MB Infinity Cache still holds -- instead of from the start?  conv (64 -> 64 @ 256x256 ... ) followed by the BatchNorm
apply pass, forward and reverse walk, and the pass alone on a cold tensor; microseconds (events around 10 pairs).
   python tools/mall_order_probe.py"""
import os
import statistics
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
sys.path.insert(0, _ROOT)
from unet_zoo_amd import _lib as L, ops

DEV, dt, B = "cuda", torch.bfloat16, 16


def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    for hw, c in ((256, 64), (128, 128), (64, 256)):
        x = ops.new_act(B, hw, hw, c, dt, DEV); x.buf.normal_()
        y = ops.new_act(B, hw, hw, c, dt, DEV)
        a = ops.new_act(B, hw, hw, c, dt, DEV)
        pooled = ops.new_act(B, hw // 2, hw // 2, c, dt, DEV)
        wp = ops.pack_weights(torch.randn(c, c, 3, 3, device=DEV) * 0.05, L.PACK_CONV_FWD, dt)
        sc, sh = torch.rand(c, device=DEV) + 0.5, torch.randn(c, device=DEV) * 0.3
        big = torch.empty(512 * 1024 * 1024 // 2, dtype=dt, device=DEV)      # 512 MB: flushes the Infinity Cache when written
        conv = lambda: ops.conv_igemm(x, wp, None, y, ntaps=9, want_stats=True)
        res = {}
        for name, fn in (("conv", conv),
                         ("conv + apply", lambda: (conv(), ops.bn_relu_apply(y, sc, sh, a))),
                         ("conv + apply reversed", lambda: (conv(), ops.bn_relu_apply(y, sc, sh, a, reverse=True))),
                         ("conv + apply+pool", lambda: (conv(), ops.bn_relu_apply(y, sc, sh, a, pooled))),
                         ("conv + apply+pool reversed", lambda: (conv(), ops.bn_relu_apply(y, sc, sh, a, pooled, reverse=True))),
                         ("flush + apply", lambda: (big.zero_(), ops.bn_relu_apply(y, sc, sh, a))),
                         ("flush", lambda: big.zero_())):
            res[name] = statistics.median(timed(fn) for _ in range(5))
        c0 = res["conv"]
        print(f"{c:4d}ch @{hw:3d}: conv {c0:6.1f} | apply after conv {res['conv + apply'] - c0:6.1f}  reversed {res['conv + apply reversed'] - c0:6.1f}"
              f" | with pool {res['conv + apply+pool'] - c0:6.1f}  reversed {res['conv + apply+pool reversed'] - c0:6.1f}"
              f" | apply on a cold tensor {res['flush + apply'] - res['flush']:6.1f}", flush=True)


if __name__ == "__main__":
    main()
