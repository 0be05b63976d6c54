"""BatchNorm passes at UNet tensor shapes (B=16, bf16) through the C ABI: forward apply, backward reduce, backward apply.
   python tools/bnbench.py [--tune=A,B,...]     (UZ_TUNE variants of the ablation build, measured interleaved)"""
import os
import sys
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("UNET_ZOO_AMD_LIB", os.path.join(_ROOT, "unet_zoo_amd", "libunetzoo_hip_ablate.so"))
sys.path.insert(0, _ROOT)
import torch
from ctypes import byref
from unet_zoo_amd import _lib as L, ops

DEV, dt, B = "cuda", torch.bfloat16, 16
SHAPES = [(256, 64), (128, 128), (64, 256), (32, 512), (16, 1024)]


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    tunes = ["0"]
    for a in sys.argv:
        if a.startswith("--tune="):
            tunes = a.split("=")[1].split(",")
    lib = L.load()
    for hw, C in SHAPES:
        y = ops.new_act(B, hw, hw, C, dt, DEV); y.buf.normal_()
        g = ops.new_act(B, hw, hw, C, dt, DEV); g.buf.normal_()
        act = ops.new_act(B, hw, hw, C, dt, DEV)
        dx = ops.new_act(B, hw, hw, C, dt, DEV)
        vec = torch.rand(4, C, device=DEV) + 0.5
        sums = torch.zeros(2, C, dtype=torch.float64, device=DEV)
        dgb = torch.empty(2, C, device=DEV)
        d = L.BnBwdDesc(L.dtype_code(dt), B, hw, hw, C, y.ld, g.ld, 0, 0, dx.ld, 0)
        args = (y.ptr(), vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(), vec[3].data_ptr(), g.ptr(), None, None)
        mb = B * hw * hw * C * 2 / 1e6
        line = f"{hw:3d}^2 x {C:4d} ({mb:5.0f} MB/tensor) |"
        for rep in range(2):
            for t in tunes:
                os.environ["UZ_TUNE"] = t
                wsb = lib.uz_bn_relu_bwd_workspace_bytes(byref(d), 0)
                ws = torch.empty(wsb // 4, dtype=torch.float32, device=DEV)
                s = L.stream_ptr()
                fwd = timeit(lambda: ops.bn_relu_apply(y, vec[0], vec[1], act, None, None, False))
                red = timeit(lambda: lib.uz_bn_relu_bwd_reduce(byref(d), *args, ws.data_ptr(), sums.data_ptr(),
                                                               dgb[0].data_ptr(), dgb[1].data_ptr(), s))
                app = timeit(lambda: lib.uz_bn_relu_bwd_apply(byref(d), *args, sums.data_ptr(), float(y.P), dx.ptr(), s))
                if rep == 1:
                    line += (f" [{t}] fwd {fwd:6.1f}us {2 * mb / fwd / 1e3:4.1f}TB/s  reduce {red:6.1f}us {2 * mb / red / 1e3:4.1f}TB/s"
                             f"  apply {app:6.1f}us {3 * mb / app / 1e3:4.1f}TB/s |")
        print(line, flush=True)


main()
