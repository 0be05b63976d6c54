"""ConvTranspose input-gradient GEMM with and without the fused BatchNorm-backward reduction, UNet decoder shapes (B=16)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("UNET_ZOO_AMD_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "unet_zoo_amd", "libunetzoo_hip_ablate.so"))
from unet_zoo_amd import _lib as L, ops
DEV, dt = "cuda", torch.bfloat16
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (N, H, W, Cin, Co) in [(16, 128, 128, 128, 64), (16, 64, 64, 256, 128), (16, 32, 32, 512, 256), (16, 16, 16, 1024, 512)]:
    g = ops.new_act(N, 2 * H, 2 * W, Co, dt, DEV); g.buf.normal_()
    w = torch.randn(Cin, Co, 2, 2, device=DEV) * 0.05
    wp = ops.pack_weights(w, L.PACK_CONVT_DGRAD, dt)
    y = ops.new_act(N, H, W, Cin, dt, DEV); y.buf.normal_()
    vec = torch.rand(4, Cin, device=DEV) + 0.5
    dx = ops.new_act(N, H, W, Cin, dt, DEV)
    line = f"{H}^2 {Cin}->{Co} |"
    for t in ("0", "0"):
        os.environ["UZ_TUNE"] = t
        a = timeit(lambda: ops.conv_igemm(g, wp, None, dx, ntaps=4, taps_mode=L.TAPS_GATHER2X2))
        b = timeit(lambda: ops.conv_igemm(g, wp, None, dx, ntaps=4, taps_mode=L.TAPS_GATHER2X2, bnred=(y, vec)))
        line += f" [{t}] plain {a:6.1f} fused {b:6.1f} |"
    print(line, flush=True)
