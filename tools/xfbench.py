"""The input-transform ("XF") forms against what they replace, interleaved in one process on one box:
   conv:   uz_conv_igemm alone | uz_bn_relu_apply + uz_conv_igemm | uz_conv_igemm_xf
   wgrad:  uz_wgrad alone      | (the apply pass is the forward's)   | uz_wgrad_xf
at the second-convolution shapes of unet's DoubleConvs (B = 16 3x256x256: C -> C at every level).
   python tools/xfbench.py [--reps=5] [--only=64,128]
Times are microseconds per launch (hipEvents around 10 back-to-back launches, median of the rounds)."""
import os
import sys
import statistics

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
sys.path.insert(0, _ROOT)
from unet_zoo_amd import _lib as L, ops

DEV = "cuda"
dt = torch.bfloat16
B = int(os.environ.get("CB_BATCH", "16"))
LAYERS = [(256, 64), (128, 128), (64, 256), (32, 512), (16, 1024)]


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    reps, only = 5, None
    for a in sys.argv[1:]:
        if a.startswith("--reps="):
            reps = int(a.split("=")[1])
        elif a.startswith("--only="):
            only = [int(v) for v in a.split("=")[1].split(",")]
    for hw, c in LAYERS:
        if only and c not in only:
            continue
        x = ops.new_act(B, hw, hw, c, dt, DEV); x.buf.normal_()
        a = ops.new_act(B, hw, hw, c, dt, DEV)
        y = ops.new_act(B, hw, hw, c, dt, DEV)
        dy = ops.new_act(B, hw, hw, c, dt, DEV); dy.buf.normal_()
        w = (torch.randn(c, c, 3, 3, device=DEV) * 0.05)
        wp = ops.pack_weights(w, L.PACK_CONV_FWD, dt)
        sc = torch.rand(c, device=DEV) + 0.5
        sh = torch.randn(c, device=DEV) * 0.3
        bias = torch.randn(c, device=DEV)
        dw = torch.empty(c, c, 3, 3, device=DEV)
        has_xf = ops.conv_xform_supported(x, c, y.ld)
        has_wxf = hasattr(ops, "wgrad_xform_supported") and ops.wgrad_xform_supported(dy, x, 9)
        fns = {
            "apply": lambda: ops.bn_relu_apply(x, sc, sh, a),
            "conv": lambda: ops.conv_igemm(a, wp, bias, y, ntaps=9, want_stats=True),
        }
        if has_xf:
            fns["conv_xf"] = lambda: ops.conv_igemm(x, wp, bias, y, ntaps=9, want_stats=True, xform=(sc, sh))
        fns["wgrad"] = lambda: ops.wgrad(dy, a, (c, c, 3, 3), ntaps=9, out=dw)
        if has_wxf:
            fns["wgrad_xf"] = lambda: ops.wgrad(dy, x, (c, c, 3, 3), ntaps=9, out=dw, xform=(sc, sh))
        t = {k: [] for k in fns}
        for _ in range(reps):
            for k, f in fns.items():
                t[k].append(timeit(f))
        med = {k: statistics.median(v) for k, v in t.items()}
        line = f"{c:5d} -> {c:<5d} @ {hw:3d}x{hw:<3d}  " + "  ".join(f"{k} {v:7.1f}" for k, v in med.items())
        if has_xf:
            line += f"   | conv: apply+conv {med['apply'] + med['conv']:7.1f} -> xf {med['conv_xf']:7.1f}"
        if has_wxf:
            line += f"   | wgrad {med['wgrad']:7.1f} -> xf {med['wgrad_xf']:7.1f}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
