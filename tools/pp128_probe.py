"""The 8 x 16-pixel ping-pong configuration (the plan's choice; UZ_TUNE bit 2 of the ablation build turns it off) against the
16 x 16 / 8 x 32 ones on the layers
whose whole-map tiles leave half of the CUs without a tile: unet's 16 x 16 maps at B = 16 (forward with statistics, input
gradient) and the same channel counts at B = 8.  Results must be bit-identical (the K order of an output does not depend on
the tile it lies in).
   python tools/pp128_probe.py"""
import os
import sys
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("UNET_ZOO_AMD_LIB", os.path.join(_ROOT, "unet_zoo_amd", "libunetzoo_hip_ablate.so"))
import torch
sys.path.insert(0, _ROOT)
from unet_zoo_amd import _lib as L, ops

DEV, dt = "cuda", torch.bfloat16
CASES = [(16, 16, 512, 1024), (16, 16, 1024, 1024), (16, 16, 1024, 512), (8, 16, 512, 512), (16, 16, 256, 256), (16, 12, 512, 1024),
         (4, 16, 512, 1024), (16, 32, 512, 256), (16, 32, 256, 128), (8, 32, 512, 512)]


def timeit(fn, n=20):
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


def main():
    for B, hw, ci, co in CASES:
        x = ops.new_act(B, hw, hw, ci, dt, DEV); x.buf.normal_()
        w = torch.randn(co, ci, 3, 3, device=DEV) * 0.03
        bias = torch.randn(co, device=DEV)
        wp = ops.pack_weights(w, L.PACK_CONV_FWD, dt)
        d = L.ConvDesc(L.dtype_code(dt), B, hw, hw, hw, hw, ci, ci, co, co, 9, L.TAPS_CONV, 1, L.STORE_PLAIN, 0, 0, 0)
        line = f"B={B:2d} {hw:2d}x{hw:<2d} {ci:4d}->{co:<4d}"
        outs = {}
        for t in ("4", "0"):
            os.environ["UZ_TUNE"] = t
            y = ops.new_act(B, hw, hw, co, dt, DEV)
            st = ops.conv_igemm(x, wp, bias, y, ntaps=9, want_stats=True)
            torch.cuda.synchronize()
            outs[t] = (y.buf.clone(), None if st is None else [s.clone() for s in st] if isinstance(st, (tuple, list)) else st.clone())
            us_s = timeit(lambda: ops.conv_igemm(x, wp, bias, y, ntaps=9, want_stats=True))
            us_p = timeit(lambda: ops.conv_igemm(x, wp, bias, y, ntaps=9))
            line += f" | [{t}] {ops.conv_kernel_name(d):24s} stats {us_s:6.1f} us  plain {us_p:6.1f} us"
        same = torch.equal(outs["0"][0], outs["4"][0])
        line += f" | outputs {'bit-identical' if same else 'DIFFER'}"
        print(line, flush=True)
    os.environ["UZ_TUNE"] = "0"


if __name__ == "__main__":
    main()
