"""Every 3x3 convolution launch of a unet step (B=16 3x256x256 bf16: forward and input-gradient shapes) through the C
ABI, kernel generations timed interleaved in one process (ablation build: UZ_TUNE selects), results compared.
   python tools/convbench.py [--tune=0,33554432] [--only=e2b,d3a] [--check] [--reps=3]
UZ_TUNE 0 = shipped plan; 0x2000000 (33554432) = without the ping-pong kernel; 0x1000000 = ping-pong wherever it applies."""
import os
import sys
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("UNET_ZOO_AMD_LIB", os.path.join(_ROOT, "unet_zoo_amd", "libunetzoo_hip_ablate.so"))
import torch
sys.path.insert(0, _ROOT)
from unet_zoo_amd import _lib as L, ops

DEV = "cuda"
dt = torch.bfloat16
B = int(os.environ.get("CB_BATCH", "16"))
# (name, H=W, Cin, Cout, bnred, count per step)
LAYERS = [
    ("e1b", 256, 64, 64, 0, 2), ("e1b.d", 256, 64, 64, 1, 2),
    ("d1a", 256, 128, 64, 0, 1), ("d1a.d", 256, 64, 128, 0, 1),
    ("e2a", 128, 64, 128, 0, 1), ("e2a.d", 128, 128, 64, 0, 1),
    ("e2b", 128, 128, 128, 0, 2), ("e2b.d", 128, 128, 128, 1, 2),
    ("d2a", 128, 256, 128, 0, 1), ("d2a.d", 128, 128, 256, 0, 1),
    ("e3a", 64, 128, 256, 0, 1), ("e3a.d", 64, 256, 128, 0, 1),
    ("e3b", 64, 256, 256, 0, 2), ("e3b.d", 64, 256, 256, 1, 2),
    ("d3a", 64, 512, 256, 0, 1), ("d3a.d", 64, 256, 512, 0, 1),
    ("e4a", 32, 256, 512, 0, 1), ("e4a.d", 32, 512, 256, 0, 1),
    ("e4b", 32, 512, 512, 0, 2), ("e4b.d", 32, 512, 512, 1, 2),
    ("d4a", 32, 1024, 512, 0, 1), ("d4a.d", 32, 512, 1024, 0, 1),
    ("e5a", 16, 512, 1024, 0, 1), ("e5a.d", 16, 1024, 512, 0, 1),
    ("e5b", 16, 1024, 1024, 0, 1), ("e5b.d", 16, 1024, 1024, 1, 1),
]


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us


def main():
    tunes, only, check, reps = ["0", str(0x2000000)], None, False, 3
    for a in sys.argv[1:]:
        if a.startswith("--tune="):
            tunes = a.split("=")[1].split(",")
        elif a.startswith("--only="):
            only = a.split("=")[1].split(",")
        elif a == "--check":
            check = True
        elif a.startswith("--reps="):
            reps = int(a.split("=")[1])
    tot = {t: 0.0 for t in tunes}
    totgf = 0.0
    for name, hw, cin, cout, bnred, cnt in LAYERS:
        if only and name not in only:
            continue
        x = ops.new_act(B, hw, hw, cin, dt, DEV); x.buf.normal_()
        w = torch.randn(cout, cin, 3, 3, device=DEV) * 0.05
        wp = ops.pack_weights(w, L.PACK_CONV_FWD, dt)
        y = ops.new_act(B, hw, hw, cout, dt, DEV)
        bias = None if bnred else torch.randn(cout, device=DEV)
        if bnred:
            bny = ops.new_act(B, hw, hw, cout, dt, DEV); bny.buf.normal_()
            vec4 = [torch.rand(cout, device=DEV) + 0.5, torch.randn(cout, device=DEV) * 0.1,
                    torch.randn(cout, device=DEV) * 0.1, torch.rand(cout, device=DEV) + 0.5]

            def run():
                return ops.conv_igemm(x, wp, None, y, ntaps=9, bnred=(bny, vec4))
        else:
            def run():
                return ops.conv_igemm(x, wp, bias, y, ntaps=9, want_stats=True)
        gflop = 2.0 * B * hw * hw * 9 * cin * cout / 1e9
        res = {t: [] for t in tunes}
        for rep in range(reps):
            for t in tunes:
                os.environ["UZ_TUNE"] = t
                res[t].append(timeit(run))
        line = f"{name:6s} {hw:3d} {cin:4d}->{cout:4d} x{cnt} {gflop:6.1f} GF |"
        for t in tunes:
            us = min(res[t])
            tot[t] += us * cnt
            line += f" [{t}] {us:7.1f} us {gflop / us * 1e3:5.0f} TF |"
        totgf += gflop * cnt
        if check and len(tunes) > 1:
            outs = []
            for t in tunes:
                os.environ["UZ_TUNE"] = t
                y.buf.zero_()
                st = run()
                torch.cuda.synchronize()
                outs.append((y.buf.float().clone(), st.double().sum(0)))
            d = (outs[0][0] - outs[1][0]).abs().max().item()
            m = outs[1][0].abs().max().item()
            ds = ((outs[0][1] - outs[1][1]).abs().max() / (outs[1][1].abs().max() + 1e-30)).item()
            nd = (outs[0][0] != outs[1][0]).float().mean().item()
            line += f" maxdiff {d:.3g} of {m:.3g}, differing {nd:.2e}, stats rel {ds:.2e}"
        print(line, flush=True)
    for t in tunes:
        print(f"total [{t}]: {tot[t]:8.1f} us per step, {totgf / tot[t] * 1e3:6.0f} TF = {totgf / tot[t] / 2.5:.3f} of 2.5 PF", flush=True)


main()
