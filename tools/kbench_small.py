"""Micro-benchmark of the small bandwidth-bound kernels at the B=16 256x256 UNet sizes."""
import os
import sys
os.environ.setdefault("UNET_ZOO_AMD_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "unet_zoo_amd", "libunetzoo_hip_ablate.so"))
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from unet_zoo_amd import _lib as L, ops

DEV, dt = "cuda", torch.bfloat16

def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

B, H = 16, 256
x = ops.new_act(B, H, H, 64, dt, DEV); x.buf.normal_()
w = torch.randn(1, 64, device=DEV); b = torch.zeros(1, device=DEV)
g = torch.randn(B, 1, H, H, device=DEV)
dx = ops.new_act(B, H, H, 64, dt, DEV)
print("outconv_fwd", timeit(lambda: ops.outconv_fwd(x, w, b)))
print("outconv_bwd", timeit(lambda: ops.outconv_bwd(x, w, g, dx)))
print("outconv_bwd(no dx)", timeit(lambda: ops.outconv_bwd(x, w, g, None)))
print("colsum 64", timeit(lambda: ops.colsum(x)))
cat = ops.new_act(B, H, H, 128, dt, DEV); cat.buf.normal_()
print("colsum window 64 of 128", timeit(lambda: ops.colsum(cat.window(0, 64))))
img = torch.randn(B, 3, H, H, device=DEV)
print("im2col", timeit(lambda: ops.im2col3x3_nchw(img, 64, dt)))
wt = torch.randn(1024, 1024, 3, 3, device=DEV)
print("pack fwd 1024x1024", timeit(lambda: ops.pack_weights(wt, L.PACK_CONV_FWD, dt)))
print("pack dgrad 1024x1024", timeit(lambda: ops.pack_weights(wt, L.PACK_CONV_DGRAD, dt)))
y = ops.new_act(B, H, H, 64, dt, DEV); y.buf.normal_()
vec = torch.rand(4, 64, device=DEV) + 0.5
act = ops.new_act(B, H, H, 64, dt, DEV)
print("bn_relu_apply", timeit(lambda: ops.bn_relu_apply(y, vec[0], vec[1], act, None)))
pooled = ops.new_act(B, H // 2, H // 2, 64, dt, DEV)
print("bn_relu_apply+pool", timeit(lambda: ops.bn_relu_apply(y, vec[0], vec[1], act, pooled)))
sums = torch.zeros(2, 64, dtype=torch.float64, device=DEV)
dgb = torch.empty(2, 64, device=DEV)
dy = ops.new_act(B, H, H, 64, dt, DEV)
print("bn_relu_bwd (1 src)", timeit(lambda: ops.bn_relu_bwd(y, vec, x, None, None, sums, dy, dgb[0], dgb[1])))
print("bn_relu_bwd (src+pool)", timeit(lambda: ops.bn_relu_bwd(y, vec, x, None, pooled, sums, dy, dgb[0], dgb[1])))

for C, HH in ((64, 256), (128, 128), (256, 64), (512, 32)):
    yy = ops.new_act(B, HH, HH, C, dt, DEV); yy.buf.normal_()
    gg = ops.new_act(B, HH, HH, C, dt, DEV); gg.buf.normal_()
    vv = torch.rand(4, C, device=DEV) + 0.5
    ss = torch.zeros(2, C, dtype=torch.float64, device=DEV)
    db = torch.empty(2, C, device=DEV)
    dd = ops.new_act(B, HH, HH, C, dt, DEV)
    for tune in os.environ.get("KB_TUNES", "0").split(","):
        os.environ["UZ_TUNE"] = tune
        print(f"bn_relu_bwd C={C} HW={HH} tune={tune}", timeit(lambda: ops.bn_relu_bwd(yy, vv, gg, None, None, ss, dd, db[0], db[1])))
