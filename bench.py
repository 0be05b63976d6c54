"""Benchmark of the BASELINE.json metric: images/sec of the 'unet' training hot path at B=16
3x256x256 bf16 per GPU on 1..8 MI355X (weak scaling: per-GPU batch fixed).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 outside a launcher starts N fresh rank processes itself (unet_zoo_amd.launch, before this
process makes any GPU call) and relays rank 0's JSON line; with fewer than N visible GPUs it exits non-zero.
The step is `unet_zoo_amd.GraphedStep` -- the product API a training loop calls -- not a harness private to this file.

One "step" = the reference's whole training step (unet_zoo/utils/training_loop.py:112-121):
zero_grad -> forward -> BCEWithLogits -> backward (-> RCCL gradient all-reduce) ->
clip_grad_norm_(1.0) -> AdamW.  Inputs are synthetic and already resident in HBM.  `value` is
images/sec of that WHOLE step (a lower bound of the fwd+bwd rate, which is reported beside it as
`fwd_bwd_images_per_s`, timed on the forward+backward hipGraph alone).

Extra objects on the JSON line:
  roofline     — the dominant kernel (most GPU time), timed live with HIP events on the launch
                 stream during the timed steps; flops/bytes are the algorithmic counts of
                 SURVEY.md §8d for exactly the launches timed.
  cpu_baseline — the CPU oracle (oracle/torch_ref.py, a port of the reference graph to
                 torch.nn.functional, pinned to the reference by tests/golden) doing the same
                 step on the host cores, rank 0, N=1 only, on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import unet_zoo_amd  # noqa: E402
from unet_zoo_amd import _lib as L, launch, ops  # noqa: E402
from unet_zoo_amd.parallel import RcclDataParallel  # noqa: E402
from unet_zoo_amd.step import GraphedStep  # noqa: E402

ABLATION_ENV = ("UZ_TUNE", "UZ_ATTN_GX", "UZ_WG_SPLIT")
PEAK = {"mfma_bf16_tflops": 2500.0, "mfma_f32_tflops": 157.3, "hbm_gbs": 8000.0}  # MI355X_MICROARCH.md


_LOSS_IMPL = "hip"

# Kernel family (the library's own names: uz_conv_igemm_kernel_name / uz_wgrad_kernel_name) -> substrings of the kernel
# symbols rocprofv3 reports for it.  The PMC table (tools/pmc_traffic.py) is keyed by symbol; a family matches every symbol
# that contains one of its substrings (all epilogue variants of a ping-pong configuration share the `PpCfg<...>` prefix).
# tests/test_bench_contract.py checks that every entry resolves in the newest committed profiles/r*_pmc_traffic.json.
PMC_KEYS = {
    # an entry = alternatives; an alternative is a substring, or a tuple of substrings that must ALL occur in the symbol
    "conv3x3_pp512_bf16": ("PpCfg<16, 32, 4, 2, 1,",),
    "conv3x3_pp512x64_bf16": ("PpCfg<16, 32, 8, 1, 3, false>",),
    "conv3x3_pp512x64_bf16_xf": ("PpCfg<16, 32, 8, 1, 3, true>",),
    "conv3x3_pp256_bf16": ("PpCfg<8, 32, 4, 2, 3,",),
    "conv3x3_pp256w16_bf16": ("PpCfg<16, 16, 4, 2, 3,",),
    "conv3x3_pp128w16_bf16": ("PpCfg<8, 16, 4, 2, 3,",),
    "wgrad9_bf16_64x64_rowwalk": (("wgrad9_kernel<64,", "false>"),),
    "wgrad9_bf16_64x64_rowwalk_xf": (("wgrad9_kernel<64,", "true>"),),
    "wgrad3x3_bf16_128x128_3tap": ("wgrad3x3_kernel<128, 128, 1, 3,",),
    "wgrad3x3_bf16_128x128_gather4": ("wgrad3x3_kernel<128, 128, 1, 1,",),
    "wgrad_g4_bf16_128x64_4tap": ("wgrad_g4_kernel<128,",),
}


def newest_pmc_file(root: str = None):
    """profiles/rNN_pmc_traffic.json with the highest round number, or None"""
    import glob
    files = sorted(glob.glob(os.path.join(root or ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))
    return files[-1] if files else None


def pmc_rows(family: str, pmc_kernels: dict) -> list:
    keys = PMC_KEYS.get(family, ())

    def hit(sym, alt):
        return alt in sym if isinstance(alt, str) else all(a in sym for a in alt)
    return [v for k, v in pmc_kernels.items() if any(hit(k, c) for c in keys)]


def torch_criterion(out, mask):
    """nn.BCEWithLogitsLoss as the reference applies it (scripts/train.py:135; dict outputs: training_loop.py:60-64)"""
    if isinstance(out, dict):
        total = None
        for v in out.values():
            l = F.binary_cross_entropy_with_logits(v, mask)
            total = l if total is None else total + l
        return total
    return F.binary_cross_entropy_with_logits(out, mask)


def model_loss(out, mask):
    """BCEWithLogits (scripts/train.py:135); u2net's dict of seven heads: their unit-weighted sum
    (unet_zoo/utils/training_loop.py:24-32, 60-64).  --loss hip (default): loss, its gradient and the step's Dice
    metric (training_loop.py:113-124) from uz_bce_dice, device scalars; --loss torch: F.binary_cross_entropy_with_logits."""
    if _LOSS_IMPL == "hip":
        from unet_zoo_amd.loss import loss_and_dice
        return loss_and_dice(out, mask)[0]
    if isinstance(out, dict):
        total = None
        for v in out.values():
            l = F.binary_cross_entropy_with_logits(v, mask)
            total = l if total is None else total + l
        return total
    return F.binary_cross_entropy_with_logits(out, mask)


def make_model(model_name: str, hw: int):
    """the reference's create_model call for this workload; swin needs image_size and a window that
    tiles the 4x4-patch token grid (8 for 256, the reference's default 7 for 224)"""
    kw = {}
    if model_name == "swin_unet_v2":
        kw = {"image_size": hw, "window_size": 7 if (hw // 4) % 7 == 0 else 8}
    elif model_name == "uctransnet":
        kw = {"image_size": hw}
    return unet_zoo_amd.create_model(model_name, in_channels=3, num_classes=1, **kw), kw


def device_state(dev_index: int = 0) -> dict:
    """Clocks, power and temperature of the GPU this rank runs on, best effort and OUTSIDE the timed region: the sysfs
    files of the card whose PCI address matches the HIP device (current pp_dpm_* level, hwmon power / temperatures),
    else `rocm-smi --json`.  Evidence for the box-to-box spread of the MFMA-bound kernels: the
    number that matters is `mfma_clock` below (the clock held UNDER matrix load), these are its context."""
    import glob
    out = {}
    try:
        pr = torch.cuda.get_device_properties(dev_index)
        want = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", -1), getattr(pr, "pci_device_id", 0))
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
        card = next((c for c in cards if want in os.path.realpath(c)), None)
        if card is None and len(cards) == 1:
            card = cards[0]
        if card:
            out["pci"] = os.path.basename(os.path.realpath(card))
            for f in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk"):
                try:
                    cur = [l for l in open(os.path.join(card, f)).read().splitlines() if l.strip().endswith("*")]
                    if cur:
                        out[f[7:] + "_mhz"] = int("".join(ch for ch in cur[0].split(":")[1] if ch.isdigit()))
                except (OSError, ValueError, IndexError):
                    pass
            for hw in glob.glob(os.path.join(card, "hwmon", "hwmon*")):
                for f, key, div in (("power1_average", "power_w", 1e6), ("power1_input", "power_w", 1e6), ("power1_cap", "power_cap_w", 1e6),
                                    ("temp1_input", "temp_edge_c", 1e3), ("temp2_input", "temp_junction_c", 1e3),
                                    ("temp3_input", "temp_mem_c", 1e3), ("freq1_input", "sclk_hwmon_mhz", 1e6)):
                    try:
                        out.setdefault(key, round(int(open(os.path.join(hw, f)).read()) / div, 1))
                    except (OSError, ValueError):
                        pass
    except Exception as e:      # noqa: BLE001  (best effort: never fail the benchmark over a sensor)
        out["error"] = repr(e)[:120]
    if len(out) <= 1:
        import subprocess
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True,
                               text=True, timeout=20)
            js = json.loads(r.stdout)
            k = sorted(js)[dev_index] if js else None
            if k:
                out["rocm_smi"] = {a: b for a, b in js[k].items() if any(t in a.lower() for t in ("sclk", "mclk", "fclk", "power", "temperature"))}
        except Exception as e:  # noqa: BLE001
            out["rocm_smi_error"] = repr(e)[:120]
    return out


def cpu_baseline(batch: int, hw: int, steps: int, model_name: str = "unet", threads: int = 16):
    """Reference step on the host CPU through the oracle (checker code, used here only as the
    reported baseline)."""
    from oracle import torch_ref
    # the GPU box gives one GPU's job a 16-core share of the host; more threads than that
    # oversubscribe and run slower (measured: 256 threads 0.05 img/s, 128 threads 0.69 img/s)
    try:
        ncpu = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        ncpu = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(threads, ncpu)))
    torch.manual_seed(0)
    m, kw = make_model(model_name, hw)
    sd = m.state_dict()
    fkw = {"cfg": torch_ref.swin_config(sd, hw, window_size=kw["window_size"])} if model_name == "swin_unet_v2" else {}
    st = torch_ref.clone_state(sd, requires_grad=True)
    params = [v for v in st.values() if v.requires_grad]
    drops = None
    if model_name == "swin_unet_v2":
        # stochastic depth as in the reference's training mode (default rate 0.1, linspace over the blocks)
        pres = [k[:-len(".norm1.weight")] for k in sd if k.endswith(".norm1.weight") and k.startswith("layers.")]
        rates = torch.linspace(0, 0.1, len(pres)).tolist()
        pres_up = [k[:-len(".norm1.weight")] for k in sd if k.endswith(".norm1.weight") and k.startswith("layers_up.")]
        # decoder level lvl reuses the encoder rates of the same level (swin_unet_v2.py:647-649)
        rate_of = dict(zip(pres, rates))
        for pu in pres_up:
            inx, b = int(pu.split(".")[1]), int(pu.split(".")[3])
            rate_of[pu] = rate_of[f"layers.{3 - inx}.blocks.{b}"]
        drops = rate_of
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-5)
    x, mask = torch_ref.synthetic_batch(batch, 3, hw, hw, seed=1234)
    times = []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        opt.zero_grad()
        if drops is not None:
            fkw["drop_scales"] = {k: torch.empty(batch).bernoulli_(1 - r) / (1 - r) for k, r in drops.items() if r > 0}
        loss = torch_ref.model_loss(torch_ref.FORWARDS[model_name](st, x, True, **fkw), mask)
        loss.backward()
        torch.nn.utils.clip_grad_norm_([p for p in params if p.grad is not None], 1.0)
        opt.step()
        times.append(time.perf_counter() - t0)
    times = sorted(times[1:])  # drop the warm-up step
    med = times[len(times) // 2]
    return {"value": round(batch / med, 4), "unit": "images/sec", "cores": torch.get_num_threads(),
            "host_cpus": os.cpu_count(),
            "kind": "port",
            "sample": f"{model_name} train step on CPU fp32, B={batch} 3x{hw}x{hw}, 1 warm-up + {steps} timed steps, median"}


def torch_rocm_baseline(batch: int, hw: int, steps: int, model_name: str = "unet", both: bool = True):
    """The SAME training step through stock PyTorch-ROCm on the same GPU (MIOpen / rocBLAS kernels, eager launches): the
    oracle's restatement of the reference graph (torch.nn.functional calls with the reference's own parameters) moved to
    the device -- (a) as the reference runs it (fp32, NCHW; scripts/train.py has neither autocast nor channels_last) and
    (b) as a user would tune it without leaving PyTorch (bf16 autocast, channels_last).  A reported yardstick beside
    `cpu_baseline` (--torch-baseline; checker code, never the measured product path).  Images/s, median step."""
    from oracle import torch_ref
    out = {}
    dev = torch.device("cuda")
    modes = (("as_reference_fp32_nchw", False, False), ("bf16_autocast_channels_last", True, True))
    for key, amp, cl in (modes if both else modes[1:]):
        torch.manual_seed(0)
        m, kw = make_model(model_name, hw)
        sd = m.state_dict()
        with torch.device(dev):      # the oracle's factory calls (attention masks, index tables) on the device
            fkw = {"cfg": torch_ref.swin_config(sd, hw, window_size=kw["window_size"])} if model_name == "swin_unet_v2" else {}
        st = torch_ref.clone_state(sd, requires_grad=False)
        for k in list(st):
            v = st[k].detach().to(dev)
            if cl and v.dim() == 4:
                v = v.contiguous(memory_format=torch.channels_last)
            if v.is_floating_point() and not torch_ref._is_buffer(k):
                v.requires_grad_(True)
            st[k] = v
        params = [v for v in st.values() if v.requires_grad]
        opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-5, fused=True)
        x, mask = torch_ref.synthetic_batch(batch, 3, hw, hw, seed=1234)
        x, mask = x.to(dev), mask.to(dev)
        if cl:
            x = x.contiguous(memory_format=torch.channels_last)
        times = []
        t_start = time.perf_counter()
        for i in range(steps + 3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            opt.zero_grad(set_to_none=True)
            with torch.device(dev), torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                outp = torch_ref.FORWARDS[model_name](st, x, True, **fkw)
            loss = torch_ref.model_loss(_as_float(outp), mask)
            loss.backward()
            torch.nn.utils.clip_grad_norm_([p for p in params if p.grad is not None], 1.0)
            opt.step()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
            if time.perf_counter() - t_start > 240 and i >= 3:    # MIOpen's first-use search can take minutes: bounded
                break
        timed = sorted(times[3:]) if len(times) > 3 else sorted(times[-1:])
        med = timed[len(timed) // 2]
        out[key] = {"images_per_s": round(batch / med, 1), "ms_per_step": round(med * 1e3, 3), "timed_steps": len(timed),
                    "first_step_s": round(times[0], 1)}
    out["what"] = (f"{model_name} train step (zero_grad+fwd+BCE+bwd+clip+fused AdamW) B={batch} 3x{hw}x{hw} through stock "
                   f"torch {torch.__version__} ops on this GPU, eager; 3 warm-up steps, median of the rest")
    return out


def _as_float(o):
    """logits of an autocast forward as fp32 for the loss (dict / tuple outputs kept)"""
    if isinstance(o, dict):
        return {k: v.float() for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return type(o)(v.float() for v in o)
    return o.float()


def time_graphed(gs, x, mask, steps, warmup, distributed, dev):
    """W untimed + exactly K timed whole steps between barrier + synchronize; returns (max-over-ranks seconds,
    seconds of K forward+backward(+all-reduce) replays alone)"""
    for _ in range(max(warmup, 1)):      # the first call dry-runs, lays out the flat buffers and captures
        gs(x, mask)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        gs(x, mask)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(steps):
        gs.forward_backward(x, mask)
    torch.cuda.synchronize()
    return elapsed, time.perf_counter() - t1


def second_headline(dev, steps: int, warmup: int, cpu_steps: int, with_cpu: bool, loss_impl: str):
    """The north star's second model on the same GPU, same process, after the headline measurement: swin_unet_v2
    (image_size 256, window 8) B = 16, whole train step from hipGraphs, with its own CPU baseline (oracle, same batch)."""
    name, B, hw = "swin_unet_v2", 16, 256
    torch.manual_seed(0)
    m, kw = make_model(name, hw)
    m.run_dtype = torch.bfloat16
    m = m.to(dev).train()
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, 3, hw, hw, generator=g).to(dev)
    mask = (torch.rand(B, 1, hw, hw, generator=g) > 0.5).float().to(dev)
    gs = GraphedStep(m, "bce_dice" if loss_impl == "hip" else torch_criterion, lr=1e-4, weight_decay=1e-5, max_norm=1.0)
    elapsed, fb_s = time_graphed(gs, x, mask, steps, warmup, False, dev)
    out = {"metric": f"images/sec (fwd+bwd) {name} B={B} 3x{hw}x{hw} (window {kw['window_size']}), 1 MI355X",
           "value": round(B * steps / elapsed, 2), "unit": "images/sec", "ms_per_step": round(elapsed / steps * 1e3, 3),
           "steps": steps, "warmup": warmup, "dtype": "bf16", "data": "synthetic",
           "fwd_bwd_ms": round(fb_s / steps * 1e3, 3), "loss": round(float(gs.loss.item()), 5), "launch": gs.describe(),
           "config": {"workload": f"{name} train step (zero_grad+fwd+BCE+bwd+clip+AdamW), B={B} 3x{hw}x{hw}, "
                                  f"stochastic depth as in the reference's training mode, random-init weights"}}
    del gs, m
    torch.cuda.empty_cache()
    if with_cpu:
        out["cpu_baseline"] = cpu_baseline(B, hw, cpu_steps, name)
        out["vs_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node (default: WORLD_SIZE, else 1)")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--model", default="unet", choices=sorted(unet_zoo_amd.hip_models()),
                    help="unet = BASELINE configs[1] (the headline metric); attention_unet = configs[2] with --size 512; "
                         "u2net = configs[4] with --size 512 --batch 8; swin_unet_v2 = the second north-star model at "
                         "--size 256 (window 8) or configs[3] with --size 224 --batch 32 (window 7)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--torch-baseline", action="store_true",
                    help="time the same step through stock PyTorch-ROCm ops on this GPU in BOTH forms (fp32 NCHW as the reference "
                         "runs, and bf16 autocast + channels_last); the default line carries the second, faster one only "
                         "(MIOpen's first-use search: ~75 s) -> \"torch_rocm_baseline\" in the line")
    ap.add_argument("--no-torch-baseline", action="store_true")
    ap.add_argument("--torch-baseline-child", default=None, metavar="OUT.json",
                    help=argparse.SUPPRESS)    # internal: the yardstick's helper process (started before the GPU is touched)
    ap.add_argument("--cpu-batch", type=int, default=None,
                    help="batch of the CPU baseline (default: the configuration's own batch, capped at 16 -- unet B=16: "
                         "~6 s per step on the 16 host threads of a one-GPU box)")
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed CPU steps after one warm-up (median; BASELINE.md section 4: >= 3)")
    ap.add_argument("--graph", default="on", choices=["on", "off"],
                    help="on: unet_zoo_amd.GraphedStep (hipGraph replays); off: eager launches through autograd + "
                         "torch's clip_grad_norm_ / AdamW (with N > 1: the bucket reducer of RcclDataParallel)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and use the multi-GPU launch strategy even for 1 rank")
    ap.add_argument("--comm", default="auto", choices=["auto", "overlap", "tail"],
                    help="N>1 ranks: all-reduce every finished backward phase beside the next one (overlap), the whole "
                         "gradient buffer once after the backward (tail), or time both before the warm-up and keep the "
                         "faster (auto; the choice and both timings are reported in the line)")
    ap.add_argument("--comm-dtype", default="fp32", choices=["fp32", "bf16"],
                    help="N>1 ranks: fp32 all-reduce of the gradients (default, the reference's semantics) or the bf16 exchange "
                         "of GraphedStep(comm_dtype=torch.bfloat16): half the bytes on the links, two bf16 roundings")
    ap.add_argument("--phases", type=int, default=5,
                    help="N>1 ranks: number of backward phases (hipGraphs) whose gradient "
                         "all-reduce overlaps the next phase; 1 = one all-reduce after the whole backward")
    ap.add_argument("--cu-reserve", type=int, default=None,
                    help="CUs the persistent convolution / GEMM / weight-gradient grids leave free for RCCL's kernels "
                         "(uz_set_cu_reserve); default: 0 for one rank, the library default otherwise")
    ap.add_argument("--loss", default="hip", choices=["hip", "torch"],
                    help="hip = BCEWithLogits + Dice + gradient in one kernel pass inside the graph (unet_zoo_amd.loss); "
                         "torch = F.binary_cross_entropy_with_logits, evaluated eagerly between the graphs")
    ap.add_argument("--profile-steps", type=int, default=5,
                    help="eager steps with per-launch HIP events, run after the timed region")
    ap.add_argument("--second-steps", type=int, default=20,
                    help="N=1 default workload only: after the headline, time this many steps of swin_unet_v2 B=16 256x256 "
                         "(the north star's second model) in the same process -> `second_headline`; 0 = skip")
    ap.add_argument("--fp32-steps", type=int, default=10,
                    help="N=1, bf16 runs: also time this many steps of the SAME model in the fp32 run mode (the "
                         "mode that meets the 1e-3 parity bound) and report fp32_images_per_s; 0 = skip")
    args = ap.parse_args()
    global _LOSS_IMPL
    _LOSS_IMPL = args.loss

    # a measurement must not depend on the environment: the shipped library ignores these switches, and an
    # ablation build (make ABLATE=1) is not what this file times
    stray = [k for k in ABLATION_ENV if os.environ.get(k)]
    if stray:
        print(f"bench.py: refusing to run with ablation switches set in the environment: {stray}", file=sys.stderr)
        sys.exit(2)

    # ---- the stock-PyTorch yardstick runs in a HELPER process: MIOpen's first-use search takes 60-80 s, and a hang or an abort
    # in a vendor library must cost the yardstick, not the line.  The helper is started here, BEFORE this process touches the
    # GPU (a GPU process must not fork + exec on these boxes), waits on its stdin, and is told to go after the timed work.
    if args.torch_baseline_child:
        if not sys.stdin.readline().strip():     # "go"; EOF = the parent is gone or skipped the yardstick: do nothing
            return
        out = {"error": "no go"}
        try:
            out = torch_rocm_baseline(args.batch, args.size, 10, args.model, both=args.torch_baseline)
            if (args.model, args.size, args.batch) == ("unet", 256, 16) and args.second_steps > 0:
                # the second headline's model through the same stock ops (no MIOpen search to speak of: ~10 s)
                try:
                    out["second_headline_swin_unet_v2"] = torch_rocm_baseline(16, 256, 10, "swin_unet_v2", both=False)[
                        "bf16_autocast_channels_last"]
                except Exception as e:      # noqa: BLE001
                    out["second_headline_swin_unet_v2"] = {"error": repr(e)[:200]}
        except Exception as e:      # noqa: BLE001
            out = {"error": repr(e)[:200]}
        with open(args.torch_baseline_child, "w") as f:
            json.dump(out, f)
        return
    helper = None
    want_yardstick = args.torch_baseline or not (args.no_torch_baseline or args.no_cpu_baseline)
    if (want_yardstick and not launch.under_launcher() and (args.gpus in (None, 1)) and not args.force_dist):
        import subprocess
        import tempfile
        helper_out = os.path.join(tempfile.mkdtemp(prefix="uz_bench_"), "torch_baseline.json")
        helper = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--torch-baseline-child", helper_out, "--model",
                                   args.model, "--batch", str(args.batch), "--size", str(args.size), "--second-steps",
                                   str(args.second_steps)]
                                  + (["--torch-baseline"] if args.torch_baseline else []),
                                  stdin=subprocess.PIPE, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    try:
        _bench_main(args, helper, helper_out if helper is not None else None)
    finally:
        if helper is not None and helper.poll() is None:      # not told to go, or still running past its time: this exact PID
            helper.kill()
            helper.wait()


def _yardstick_from_helper(helper, helper_out, limit_s: float):
    """tell the helper to go, wait at most limit_s, read what it wrote"""
    import subprocess
    try:
        helper.stdin.write(b"go\n")
        helper.stdin.flush()
        helper.wait(timeout=limit_s)
        return json.load(open(helper_out))
    except subprocess.TimeoutExpired:
        helper.kill()
        helper.wait()
        return {"error": f"the stock-PyTorch yardstick did not finish within {limit_s:.0f} s (MIOpen's search); killed"}
    except Exception as e:      # noqa: BLE001
        return {"error": repr(e)[:200]}


def _bench_main(args, helper, helper_out):
    global _LOSS_IMPL
    _LOSS_IMPL = args.loss
    # ---- ranks: one process per GPU, started BEFORE this process touches the GPU ---------------------------
    if not launch.under_launcher():
        want = 1 if args.gpus is None else args.gpus
        if want > 1:
            sys.exit(launch.spawn_ranks(want, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
        if want < 1:
            print(f"bench.py: --gpus must be >= 1, got {want}", file=sys.stderr)
            sys.exit(2)
    rank, local_rank, world = launch.rank_info()
    if args.gpus is not None and args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch exactly one rank per GPU "
              f"(python bench.py --gpus N does that itself)", file=sys.stderr)
        sys.exit(2)
    if unet_zoo_amd._lib.load().uz_build_ablate():
        print("bench.py: libunetzoo_hip.so is an ablation build (make ABLATE=1); rebuild without it", file=sys.stderr)
        sys.exit(2)
    distributed = world > 1 or args.force_dist
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    run_dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(0)
    model, _ = make_model(args.model, args.size)
    model.run_dtype = run_dtype
    model = model.to(dev).train()
    params = list(model.parameters())

    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.randn(args.batch, 3, args.size, args.size, generator=g).to(dev)
    mask = (torch.rand(args.batch, 1, args.size, args.size, generator=g) > 0.5).float().to(dev)

    fb_events = []
    net = model
    opt = None

    def eager_step(timed: bool):
        """the reference's loop body, launched eagerly (training_loop.py:112-121)"""
        opt.zero_grad(set_to_none=True)
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        loss = model_loss(net(x), mask)
        loss.backward()
        if timed:
            e1.record()
            fb_events.append((e0, e1))
        torch.nn.utils.clip_grad_norm_([p for p in params if p.grad is not None], 1.0, foreach=True)
        opt.step()
        return loss

    gs = None
    comm_tuning = None
    fb_graph_ms = None
    loss_recheck = None
    dev_before = device_state(dev.index or 0) if rank == 0 else None
    if args.graph == "on":
        gs = GraphedStep(model, "bce_dice" if args.loss == "hip" else torch_criterion, lr=1e-4, weight_decay=1e-5,
                         max_norm=1.0, phases=args.phases, data_parallel=distributed, cu_reserve=args.cu_reserve,
                         comm="overlap" if args.comm == "auto" else args.comm,
                         comm_dtype=torch.bfloat16 if args.comm_dtype == "bf16" else None)
        comm_tuning = None
        if distributed and args.comm == "auto":
            # before the warm-up and the timed steps: both collective schedules on this job's own ranks, keep the faster
            comm_tuning = gs.autotune_comm(x, mask)
            comm_tuning = {k: round(v, 3) for k, v in comm_tuning.items()}
            comm_tuning["chosen"] = gs.comm
        elapsed, fb_s = time_graphed(gs, x, mask, args.steps, args.warmup, distributed, dev)
        fb_graph_ms = fb_s / args.steps * 1e3
        launch_mode = gs.describe()
        # the loss comes straight out of the replayed graph (uz_bce_dice: no library reduction, no memset node);
        # re-evaluated eagerly from the same replay's logits it must be the same number
        final_loss = float(gs.loss.item())
        with torch.no_grad():
            loss_recheck = float(torch_criterion(gs.outputs, mask).item())
    else:
        if distributed:
            net = RcclDataParallel(model)
        opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-5, fused=True)
        for _ in range(max(args.warmup, 1)):
            eager_step(False)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = eager_step(False)
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = t.item()
        final_loss = float(loss.item())
        launch_mode = "eager launches (autograd node + torch clip_grad_norm_ + fused AdamW)"

    dev_after = device_state(dev.index or 0) if rank == 0 else None
    # per-launch HIP events (same process, same shapes, eager launches right after the timed steps)
    net = model
    if opt is None:
        opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-5, fused=True)
    model._grad_sink = None
    model._grad_sink_done = None
    for p in params:          # the eager profiling steps own their gradients
        p.grad = None
    ops.profile_begin()
    for _ in range(args.profile_steps):
        eager_step(True)
    prof = ops.profile_end()
    nprof = max(args.profile_steps, 1)

    # the same model in the fp32 run mode (exact-fp32 MFMA, the mode the parity tests bound at 1e-3)
    fp32 = None
    if world == 1 and not distributed and args.dtype == "bf16" and args.fp32_steps > 0 and gs is not None:
        del opt
        for p in params:
            p.grad = None
        torch.manual_seed(0)
        m32, _ = make_model(args.model, args.size)
        m32.run_dtype = torch.float32
        m32 = m32.to(dev).train()
        gs32 = GraphedStep(m32, "bce_dice" if args.loss == "hip" else torch_criterion, lr=1e-4, weight_decay=1e-5)
        el32, _ = time_graphed(gs32, x, mask, args.fp32_steps, 2, False, dev)
        fp32 = {"fp32_images_per_s": round(args.batch * args.fp32_steps / el32, 2),
                "fp32_ms_per_step": round(el32 / args.fp32_steps * 1e3, 3), "fp32_steps": args.fp32_steps,
                "fp32_loss": round(float(gs32.loss.item()), 5)}
        del gs32, m32

    mfma_clock = None
    if rank == 0 and world == 1:
        try:
            mfma_clock = L.mfma_clock_ghz()      # ~1 s of dense MFMAs, after everything that is timed
        except Exception as e:                   # noqa: BLE001
            mfma_clock = {"error": repr(e)[:120]}
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * args.batch * args.steps / elapsed
        if fb_graph_ms is not None:
            fb_ms = fb_graph_ms
        else:
            fb_ms = sorted(a.elapsed_time(b) for a, b in fb_events)[len(fb_events) // 2] if fb_events else None
        # dominant kernel = the kernel with the largest summed duration.  A kernel = one tile configuration of one
        # template: the `_bnred` family of a convolution (the same main loop and tiles with the BatchNorm-backward sums in
        # the epilogue, uz_conv_igemm_bnred) is counted with its base family -- round 3's convolution and the three-tap
        # weight gradient are within 3 % of each other per FAMILY and swapped places from run to run
        groups: dict = {}
        for name, v in prof.items():
            gname = name[:-len("_bnred")] if name.endswith("_bnred") else name
            gd = groups.setdefault(gname, {"ms": 0.0, "launches": 0, "flops": 0.0, "bytes": 0.0, "families": []})
            for k in ("ms", "launches", "flops", "bytes"):
                gd[k] += v[k]
            gd["families"].append(name)
        dom_name, dom = max(groups.items(), key=lambda kv: kv[1]["ms"]) if groups else (None, None)
        roofline = None
        # HBM traffic of the dominant kernel: from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE in separate runs of this same command, FETCH_SIZE doubled as MI355X_MICROARCH.md
        # prescribes for gfx950) -- counters cannot be read from inside this process
        traffic = None
        try:
            pmc_path = newest_pmc_file()
            with open(pmc_path) as f:
                pmc = json.load(f)["kernels"]
            # (both epilogue variants of a ping-pong configuration belong to the kernel: their symbols share the prefix)
            rows = pmc_rows(dom_name, pmc)
            if rows and args.model == "unet" and args.size == 256 and args.batch == 16:
                n = sum(r["launches"] for r in rows)
                traffic = {"hbm_read_mb_per_launch": round(sum(r["hbm_read_mb_per_launch_corrected"] * r["launches"] for r in rows) / n, 2),
                           "hbm_write_mb_per_launch": round(sum(r["hbm_write_mb_per_launch"] * r["launches"] for r in rows) / n, 2),
                           "algorithmic_mb_per_launch": round(dom["bytes"] / dom["launches"] / 2 ** 20, 2),
                           "source": "profiles/" + os.path.basename(pmc_path)}
        except (OSError, KeyError, ValueError, TypeError):
            pass
        if dom is not None:
            flops_per_launch = dom["flops"] / dom["launches"]
            sec_per_launch = dom["ms"] * 1e-3 / dom["launches"]
            ach = flops_per_launch / sec_per_launch / 1e12
            peak = PEAK["mfma_bf16_tflops"] if run_dtype == torch.bfloat16 else PEAK["mfma_f32_tflops"]
            gbs = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
            # which roof bounds the kernel: its algorithmic intensity against the ridge point peak_flops / peak_bytes
            ridge = peak * 1e12 / (PEAK["hbm_gbs"] * 1e9)
            if dom["bytes"] > 0 and dom["flops"] / dom["bytes"] < ridge:
                roofline = {"kernel": dom_name, "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK["hbm_gbs"],
                            "unit": "GB/s", "frac": round(gbs / PEAK["hbm_gbs"], 4), "traffic": traffic,
                            "algorithmic_tflops": round(ach, 2)}
            else:
                roofline = {"kernel": dom_name, "bound": "mfma", "achieved": round(ach, 2), "peak": peak,
                            "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
                            "algorithmic_gbytes_per_s": round(gbs, 1)}
            roofline.update({"families": sorted(dom["families"]), "launches_per_step": dom["launches"] // nprof,
                             "avg_launch_us": round(sec_per_launch * 1e6, 2),
                             "share_of_step": round(dom["ms"] / nprof / ms, 4)})
        line = {
            "metric": "images/sec (fwd+bwd) at B=16 3x256x256, 1/2/4/8 MI355X" if (args.model, args.size, args.batch) == ("unet", 256, 16)
                      else f"images/sec (fwd+bwd) {args.model} B={args.batch} 3x{args.size}x{args.size}",
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.model} train step (zero_grad+fwd+BCE+bwd+clip+AdamW), B={args.batch}/GPU "
                                   f"3x{args.size}x{args.size}, random-init weights",
                       "global_batch": world * args.batch, "parallelism": f"dp{world}"},
            "fwd_bwd_ms": round(fb_ms, 3) if fb_ms else None,
            "fwd_bwd_images_per_s": round(args.batch * world / (fb_ms * 1e-3), 2) if fb_ms else None,
            "loss": round(final_loss, 5),
            "loss_eager_recheck": round(loss_recheck, 5) if loss_recheck is not None else None,
            "roofline": roofline,
            "launch": launch_mode,
            "cu_reserve": L.get_cu_reserve(),
            "comm_tuning_ms": comm_tuning if args.graph == "on" else None,
            "comm_dtype": args.comm_dtype if distributed else None,
            "device": {"before_timed_steps": dev_before, "after_timed_steps": dev_after, "mfma_clock": mfma_clock},
            "kernel_ms_per_step": {k: round(v["ms"] / nprof, 3) for k, v in sorted(prof.items())},
        }
        # the north star's block-level line: unet level 1 DoubleConv forward (im2col + both convolutions + BatchNorm
        # finalize / apply + ReLU + pool) as the eager profile steps timed it, against SURVEY 8d's floor
        # 4 * P * Cin + s * P * 3 * Cout + weights (415 MB at B = 16 256 x 256 bf16) and 8 TB/s
        sc = ops.profile_scopes().get("doubleconv_l1")
        if sc and args.model == "unet" and args.profile_steps > 0:
            P_, es_ = args.batch * args.size * args.size, (2 if run_dtype == torch.bfloat16 else 4)
            # the network input is fp32 NCHW (4 bytes per element), the activations are in the run dtype
            alg = 4 * P_ * 3 + es_ * P_ * (3 * 64) + 4 * (9 * 3 * 64 + 64 + 9 * 64 * 64 + 64)
            us = sc["ms"] / nprof * 1e3
            line["doubleconv_l1"] = {
                "block": "unet down_convolution_1: DoubleConv(3->64->64, train BN, ReLU) + MaxPool, forward",
                "algorithmic_mb": round(alg / 1e6, 1), "us": round(us, 1), "launches": sc["launches"] // nprof,
                "achieved_gbs": round(alg / us / 1e3, 1), "peak_gbs": PEAK["hbm_gbs"],
                "frac": round(alg / us / 1e3 / PEAK["hbm_gbs"], 4),
                "kernels_us": {k: round(v / nprof * 1e3, 1) for k, v in sorted(sc["kernels"].items())}}
        if fp32 is not None:
            line.update(fp32)
        if (world == 1 and not distributed and args.second_steps > 0 and args.graph == "on" and args.dtype == "bf16"
                and (args.model, args.size, args.batch) == ("unet", 256, 16)):
            try:
                line["second_headline"] = second_headline(dev, args.second_steps, args.warmup, args.cpu_steps,
                                                          not args.no_cpu_baseline, args.loss)
            except Exception as e:      # noqa: BLE001  (the headline line must not be lost over the second one)
                line["second_headline"] = {"error": repr(e)[:200]}
        if world == 1 and not args.no_cpu_baseline:
            cpu_b = args.cpu_batch if args.cpu_batch else min(args.batch, 16)
            cb = cpu_baseline(cpu_b, args.size, args.cpu_steps, args.model)
            # (8 threads, to compare with the survey's anchors: a quarter of the batch keeps the default run short)
            cb["value_8_threads"] = cpu_baseline(max(cpu_b // 4, 1), args.size, 1, args.model, threads=8)["value"]
            line["cpu_baseline"] = cb
        if helper is not None:
            # 75-80 s of search + steps for the bf16 form; both forms (--torch-baseline) twice that and more at 512 x 512
            line["torch_rocm_baseline"] = _yardstick_from_helper(helper, helper_out, 900.0 if args.torch_baseline else 300.0)
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
