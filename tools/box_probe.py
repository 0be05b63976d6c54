"""What differs between MI355X boxes?  One call, one box: the clock held under dense MFMA load, HBM copy and cached-read
bandwidths, LDS-DMA fill rate from L2 / from HBM (the ping-pong convolution on a tiny-K and a long-K layer separates the
two), the unet layers that dominate the step, and device sensors.  Prints one JSON line; collect one from a fast and one
from a slow box (tools: gpurun_out/boxprobe_*.json) and compare field by field.
   python tools/box_probe.py"""
import json
import os
import sys
import time
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
sys.path.insert(0, _ROOT)
from unet_zoo_amd import _lib as L, ops

DEV = "cuda"
dt = torch.bfloat16


def ev_time(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us


def conv(N, S, Cin, Cout):
    x = ops.new_act(N, S, S, Cin, dt, DEV); x.buf.normal_()
    w = torch.randn(Cout, Cin, 3, 3, device=DEV) * 0.05
    wp = ops.pack_weights(w, L.PACK_CONV_FWD, dt)
    y = ops.new_act(N, S, S, Cout, dt, DEV)
    us = ev_time(lambda: ops.conv_igemm(x, wp, None, y, ntaps=9, want_stats=True))
    return round(us, 1), round(2.0 * N * S * S * 9 * Cin * Cout / us / 1e6, 0)


def wgrad(N, S, Cin, Cout):
    x = ops.new_act(N, S, S, Cin, dt, DEV); x.buf.normal_()
    dy = ops.new_act(N, S, S, Cout, dt, DEV); dy.buf.normal_()
    us = ev_time(lambda: ops.wgrad(dy, x, (Cout, Cin, 3, 3), ntaps=9))
    return round(us, 1), round(2.0 * N * S * S * 9 * Cin * Cout / us / 1e6, 0)


def main():
    out = {}
    out["mfma_clock"] = L.mfma_clock_ghz()
    # HBM: 1 GiB copy (read + write), far beyond the 256 MB Infinity Cache
    a = torch.empty(1 << 28, dtype=torch.float32, device=DEV); b = torch.empty_like(a)
    a.normal_()
    us = ev_time(lambda: b.copy_(a), n=5)
    out["hbm_copy_gbs"] = round(2 * a.numel() * 4 / us / 1e3, 0)
    # Infinity Cache / L2: 64 MB buffer summed repeatedly (reads only)
    c = torch.empty(1 << 24, dtype=torch.float32, device=DEV).normal_()
    us = ev_time(lambda: c.sum(), n=20)
    out["read_64mb_gbs"] = round(c.numel() * 4 / us / 1e3, 0)
    d = torch.empty(1 << 20, dtype=torch.float32, device=DEV).normal_()
    us = ev_time(lambda: d.sum(), n=50)
    out["read_4mb_us"] = round(us, 2)
    del a, b, c, d
    # the kernels of the unet step: MFMA-bound with little (d3a) / much (e2a) traffic per flop, the weight gradient
    out["conv_d3a_512to256_64"] = conv(16, 64, 512, 256)
    out["conv_e2b_128to128_128"] = conv(16, 128, 128, 128)
    out["conv_e2a_64to128_128"] = conv(16, 128, 64, 128)
    out["conv_e1b_64to64_256"] = conv(16, 256, 64, 64)
    out["conv_e4b_512to512_32"] = conv(16, 32, 512, 512)
    out["wgrad_e3b_256to256_64"] = wgrad(16, 64, 256, 256)
    out["wgrad_d2a_256to128_128"] = wgrad(16, 128, 256, 128)
    # a pure bandwidth kernel of the step
    y = ops.new_act(16, 256, 256, 64, dt, DEV); y.buf.normal_()
    act = ops.new_act(16, 256, 256, 64, dt, DEV)
    sc, sh = torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV)
    us = ev_time(lambda: ops.bn_relu_apply(y, sc, sh, act, None))
    out["bn_apply_64ch_256_us"] = round(us, 1)
    out["bn_apply_gbs"] = round(2 * y.buf.numel() * 2 / us / 1e3, 0)
    out["mfma_clock_after"] = L.mfma_clock_ghz(settle_s=0.3)
    try:
        sys.path.insert(0, _ROOT)
        import bench
        out["sensors"] = bench.device_state(0)
    except Exception as e:   # noqa: BLE001
        out["sensors"] = repr(e)[:100]
    print(json.dumps(out), flush=True)


main()
