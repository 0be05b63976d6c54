"""Token GEMMs of swin_unet_v2 (B = 16, 256 x 256, window 8) and missformer through the C ABI: y = x W^T (+ bias), bf16.
   python tools/gemmbench.py     -> us per launch, algorithmic GB/s (read x + write y + weights once)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_zoo_amd import _lib as L, ops

DEV, dt = "cuda", torch.bfloat16
# (name, tokens M, K, N)
SHAPES = [("swin s0 qkv", 65536, 96, 288), ("swin s0 proj", 65536, 96, 96), ("swin s1 qkv", 16384, 192, 576),
          ("swin s1 proj", 16384, 192, 192), ("swin s2 qkv", 4096, 384, 1152), ("swin s2 proj", 4096, 384, 384),
          ("swin s3 qkv", 1024, 768, 2304), ("swin merge0", 16384, 384, 192), ("swin expand", 16384, 192, 384),
          ("swin x4 expand", 65536, 96, 1536), ("mit fc1 s0", 131072, 64, 256), ("mit fc2 s0", 131072, 256, 64),
          ("mit q s1", 32768, 128, 128), ("mit fc1 s1", 32768, 128, 512)]


def timeit(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, M, K, N in SHAPES:
    x = ops.new_act(1, 1, M, K, dt, DEV); x.buf.normal_()
    w = torch.randn(N, K, device=DEV) * 0.05
    wp = ops.pack_weights(w.reshape(N, K, 1, 1), L.PACK_CONV_FWD, dt)
    b = torch.randn(N, device=DEV)
    y = ops.new_act(1, 1, M, N, dt, DEV)
    us = min(timeit(lambda: ops.conv_igemm(x, wp, b, y, ntaps=1)) for _ in range(2))
    mb = (M * K + M * N + N * K) * 2 / 1e6
    gf = 2.0 * M * K * N / 1e9
    d = L.ConvDesc(L.dtype_code(dt), 1, 1, M, 1, M, K, x.ld, N, y.ld, 1, L.TAPS_CONV, 1, L.STORE_PLAIN, 0, 0, 0)
    print(f"{name:16s} M {M:6d} K {K:4d} N {N:4d} | {us:7.1f} us | {mb / us * 1e3 / 1e3:5.2f} TB/s | {gf / us * 1e3 / 1e3:6.1f} TF | {ops.conv_kernel_name(d)}", flush=True)
