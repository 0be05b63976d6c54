"""Window attention forward / backward at the swin_unet_v2 B=16 256x256 stage sizes (us per call)."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from unet_zoo_amd import _lib as L, ops

DEV, dt = "cuda", torch.bfloat16


def timeit(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def case(B, H, heads, ws=8, shift=4):
    C = 32 * heads
    N = ws * ws
    qkv = ops.new_act(B, H, H, 3 * C, dt, DEV); qkv.buf.normal_()
    out = ops.new_act(B, H, H, C, dt, DEV)
    dout = ops.new_act(B, H, H, C, dt, DEV); dout.buf.normal_()
    dqkv = ops.new_act(B, H, H, 3 * C, dt, DEV)
    tau = torch.rand(heads, N, N, device=DEV) + 0.5
    bias = torch.randn(heads, N, N, device=DEV)
    lse = ops.winattn_fwd(qkv, tau, bias, out, heads, ws, shift)
    tf = timeit(lambda: ops.winattn_fwd(qkv, tau, bias, out, heads, ws, shift))
    tb = timeit(lambda: ops.winattn_bwd(qkv, tau, bias, out, lse, dout, dqkv, heads, ws, shift))
    units = B * (H // ws) ** 2 * heads
    print(f"B{B} {H}x{H} heads {heads}: {units} units  fwd {tf:7.1f} us   bwd(+row sums) {tb:7.1f} us")


case(16, 64, 3)
case(16, 32, 6)
case(16, 16, 12)
case(16, 8, 24, shift=0)
