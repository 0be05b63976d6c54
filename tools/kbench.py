"""In-process kernel micro-benchmark: UNet layer shapes (B=16, 256x256, bf16) through the C ABI.
   python tools/kbench.py [conv|wgrad|all] [--tune A,B,...]   (UZ_TUNE variants measured interleaved)"""
import os
import sys
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the UZ_TUNE switches exist only in the ablation build (make -C unet_zoo_amd/csrc ABLATE=1)
os.environ.setdefault("UNET_ZOO_AMD_LIB", os.path.join(_ROOT, "unet_zoo_amd", "libunetzoo_hip_ablate.so"))
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from unet_zoo_amd import _lib as L, ops

DEV = "cuda"
dt = torch.bfloat16
B = 16
# (name, H=W, Cin, Cout)
ONLY = os.environ.get("KB_ONLY")
LAYERS = [("e1b", 256, 64, 64), ("e2a", 128, 64, 128), ("e2b", 128, 128, 128), ("e3a", 64, 128, 256),
          ("e3b", 64, 256, 256), ("e4a", 32, 256, 512), ("e4b", 32, 512, 512), ("bna", 16, 512, 1024),
          ("bnb", 16, 1024, 1024), ("d1a", 32, 1024, 512), ("d2a", 64, 512, 256), ("d3a", 128, 256, 128),
          ("d4a", 256, 128, 64), ("d4ad", 256, 64, 128), ("n32", 256, 32, 32)]

def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us

CONVT = [("up1", 16, 1024, 512), ("up2", 32, 512, 256), ("up3", 64, 256, 128), ("up4", 128, 128, 64)]


def convt_wgrad():
    """the four ConvTranspose2d(k2 s2) weight gradients of unet (B = 16): x on the coarse grid, g in the up-slot of the
    decoder's concat buffer (ld = 2 Cout)"""
    tunes = ["0"]
    for a in sys.argv:
        if a.startswith("--tune="):
            tunes = a.split("=")[1].split(",")
    for name, hc, cin, cout in CONVT:
        x = ops.new_act(B, hc, hc, cin, dt, DEV); x.buf.normal_()
        full = ops.new_act(B, 2 * hc, 2 * hc, 2 * cout, dt, DEV); full.buf.normal_()
        g = full.window(0, cout)
        mb = (x.buf.numel() + g.P * cout) * 2 / 1e6
        line = f"{name:4s} {hc:3d}->{2 * hc:3d} {cin:4d}->{cout:4d} {mb:6.1f} MB |"
        for rep in range(2):
            for t in tunes:
                os.environ["UZ_TUNE"] = t
                us = timeit(lambda: ops.wgrad(x, g, (cin, cout, 2, 2), ntaps=4, taps_mode=L.TAPS_GATHER2X2))
                if rep == 1:
                    line += f" [{t}] {us:7.1f}us {mb / us * 1e3 / 1e3:5.2f}TB/s |"
        print(line, flush=True)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what == "convt":
        return convt_wgrad()
    tunes = ["0"]
    for a in sys.argv:
        if a.startswith("--tune="):
            tunes = a.split("=")[1].split(",")
    for name, hw, cin, cout in LAYERS:
        if ONLY and name not in ONLY.split(','):
            continue
        x = ops.new_act(B, hw, hw, cin, dt, DEV); x.buf.normal_()
        dy = ops.new_act(B, hw, hw, cout, dt, DEV); dy.buf.normal_()
        w = torch.randn(cout, cin, 3, 3, device=DEV) * 0.05
        wp = ops.pack_weights(w, L.PACK_CONV_FWD, dt)
        y = ops.new_act(B, hw, hw, cout, dt, DEV)
        gflop = 2.0 * B * hw * hw * 9 * cin * cout / 1e9
        res = []
        for rep in range(2):
            for t in tunes:
                os.environ["UZ_TUNE"] = t
                if what in ("conv", "all"):
                    us = timeit(lambda: ops.conv_igemm(x, wp, None, y, ntaps=9, want_stats=True))
                    res.append((t, "conv", us))
                if what in ("wgrad", "all"):
                    us = timeit(lambda: ops.wgrad(dy, x, (cout, cin, 3, 3), ntaps=9))
                    res.append((t, "wgrad", us))
        if what in ("wgrad", "all") and len(tunes) > 1:      # the variants must agree (fp32 summation order aside)
            ref = None
            for t in tunes:
                os.environ["UZ_TUNE"] = t
                o = ops.wgrad(dy, x, (cout, cin, 3, 3), ntaps=9).double()
                if ref is None:
                    ref = o
                else:
                    err = ((o - ref).abs().max() / ref.abs().max()).item()
                    if err > 1e-4:
                        print(f"  !! wgrad[{t}] differs from wgrad[{tunes[0]}]: rel {err:.2e}", flush=True)
        out = {}
        for t, k, us in res:
            out.setdefault((t, k), []).append(us)
        line = f"{name:4s} {hw:3d} {cin:4d}->{cout:4d} {gflop:7.1f} GF |"
        for (t, k), v in out.items():
            us = min(v)
            line += f" {k}[{t}] {us:7.1f}us {gflop / us * 1e3:5.0f}TF |"
        print(line, flush=True)

main()
