"""LayerNorm forward / backward at the swin_unet_v2 B=16 256x256 sizes: us and GB/s per call."""
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


def case(B, H, C, mode=L.LN_PLAIN, r=1, res=True):
    if mode == L.LN_EXPAND:
        x = ops.new_act(B, H // r, H // r, r * r * C, dt, DEV)
    else:
        x = ops.new_act(B, H, H, C, dt, DEV)
    x.buf.normal_()
    y = ops.new_act(B, H, H, C, dt, DEV)
    rr = ops.new_act(B, H, H, C, dt, DEV) if res else None
    g = ops.new_act(B, H, H, C, dt, DEV); g.buf.normal_()
    dx = ops.new_act(x.N, x.H, x.W, x.C, dt, DEV)
    gam, bet = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    stats = ops.layernorm_fwd(x, gam, bet, y, mode=mode, r=r, res=rr)
    tf = timeit(lambda: ops.layernorm_fwd(x, gam, bet, y, mode=mode, r=r, res=rr))
    lib = L.load()
    tb = timeit(lambda: ops.layernorm_bwd(x, gam, stats, g, dx, mode=mode, r=r))
    mb = B * H * H * C * 2 / 1e6
    print(f"B{B} {H}x{H} C{C} mode{mode}: fwd {tf:7.1f} us {mb * (2 + res) / tf * 1e-3:5.2f} TB/s   "
          f"bwd(+row sum) {tb:7.1f} us {mb * 3 / tb * 1e-3:5.2f} TB/s")


case(16, 64, 96)
case(16, 32, 192)
case(16, 16, 384)
case(16, 8, 768)
case(16, 256, 96, L.LN_EXPAND, 4, res=False)
if "--missformer" in sys.argv:   # the B=8 512x512 missformer sizes (stage-1 tokens, MixFFN width, bridge rows)
    case(8, 128, 64, res=False)
    case(8, 128, 256, res=False)
    case(8, 64, 128, res=False)
    case(8, 64, 512, res=False)
    case(8, 32, 1280, res=False)
