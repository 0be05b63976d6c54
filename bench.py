"""Benchmark of the BASELINE.json metric: images/sec of the 'unet' training hot path at B=16
3x256x256 bf16 per GPU on 1..8 MI355X (weak scaling: per-GPU batch fixed).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

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
from unet_zoo_amd import ops  # noqa: E402
from unet_zoo_amd.graph import PhasedStep
from unet_zoo_amd.optim import FlatClipAdamW
from unet_zoo_amd.parallel import RcclDataParallel  # noqa: E402

# hipGraph capture checks only THIS thread's calls: the process-group watchdog thread polls its events
# concurrently (legal for it, but fatal to a capture in the default "global" mode)
CAPTURE_MODE = "thread_local"
PEAK = {"mfma_bf16_tflops": 2500.0, "mfma_f32_tflops": 157.3, "hbm_gbs": 8000.0}  # MI355X_MICROARCH.md


_LOSS_IMPL = "hip"


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
    return unet_zoo_amd.create_model(model_name, in_channels=3, num_classes=1, **kw), kw


def cpu_baseline(batch: int, hw: int, steps: int, model_name: str = "unet"):
    """Reference step on the host CPU through the oracle (checker code, used here only as the
    reported baseline)."""
    from oracle import torch_ref
    # the GPU box gives one GPU's job a 16-core share of the host; more threads than that
    # oversubscribe and run slower (measured: 256 threads 0.05 img/s, 128 threads 0.69 img/s)
    try:
        ncpu = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        ncpu = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(16, ncpu)))
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--model", default="unet", choices=["unet", "attention_unet", "u2net", "swin_unet_v2", "nested_unet", "resunet", "missformer"],
                    help="unet = BASELINE configs[1] (the headline metric); attention_unet = configs[2] with --size 512; "
                         "u2net = configs[4] with --size 512 --batch 8; swin_unet_v2 = the second north-star model at "
                         "--size 256 (window 8) or configs[3] with --size 224 --batch 32 (window 7)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=4)
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="replay the whole step from one hipGraph (auto: try, fall back to eager)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and use the multi-GPU launch strategy even for 1 rank")
    ap.add_argument("--optimizer", default="flat", choices=["flat", "torch"],
                    help="graph mode: flat = clip + AdamW as three launches on flat buffers (unet_zoo_amd.optim), "
                         "torch = torch.nn.utils.clip_grad_norm_ + fused torch.optim.AdamW")
    ap.add_argument("--phases", type=int, default=5,
                    help="N>1 ranks, graph mode: number of backward phases (hipGraphs) whose gradient "
                         "all-reduce overlaps the next phase; 1 = one all-reduce after the whole backward")
    ap.add_argument("--loss", default="hip", choices=["hip", "torch"],
                    help="hip = BCEWithLogits + Dice + gradient in one kernel pass (unet_zoo_amd.loss); torch = F.binary_cross_entropy_with_logits")
    ap.add_argument("--profile-steps", type=int, default=5,
                    help="eager steps with per-launch HIP events, run after the timed region")
    args = ap.parse_args()
    global _LOSS_IMPL
    _LOSS_IMPL = args.loss

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or args.force_dist:
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
    net = RcclDataParallel(model) if (world > 1 or args.force_dist) else model
    params = list(model.parameters())
    use_graph = args.graph != "off"
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-5, fused=True, capturable=use_graph)

    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.randn(args.batch, 3, args.size, args.size, generator=g).to(dev)
    mask = (torch.rand(args.batch, 1, args.size, args.size, generator=g) > 0.5).float().to(dev)

    fb_events = []

    def fwd_bwd(timed: bool = False):
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        out = net(x)
        loss = model_loss(out, mask)
        loss.backward()
        if timed:
            e1.record()
            fb_events.append((e0, e1))
        return loss

    def opt_step():
        torch.nn.utils.clip_grad_norm_(params, 1.0, foreach=True)
        opt.step()

    def step(timed: bool):
        opt.zero_grad(set_to_none=True)
        loss = fwd_bwd(timed)
        opt_step()
        return loss

    # warm-up (eager, on a side stream so that a graph can be captured afterwards)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(max(args.warmup, 1)):
            loss = step(False)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()

    # ---- launch strategy ---------------------------------------------------------------------
    #  The step is two hipGraphs: (zero + fwd + loss + bwd, gradients accumulated into one flat fp32
    #  buffer) and (clip + AdamW).  With N > 1 ranks ONE eager RCCL all-reduce (AVG) of the flat
    #  buffer runs between them: collectives stay out of graph capture.  (The bucket reducer that
    #  overlaps all-reduce with backward, parallel.RcclDataParallel, is the eager path: --graph off.)
    run_one = None
    launch_mode = "eager"
    distributed = world > 1 or args.force_dist
    if use_graph:
        try:
            inner = net.module if isinstance(net, RcclDataParallel) else net
            inner._grad_sink = None          # gradients are reduced from the flat buffer instead
            inner._grad_sink_done = None
            # parameters the graph never reaches (swin's mlp / norm2, swin_unet_v2.py:264-267) keep
            # .grad = None exactly as in the reference, so clip and AdamW (incl. weight decay) skip them
            used = [p for p in params if p.grad is not None]
            for p in params:
                p.grad = None
            inner.grads_in_place = True      # kernels write straight into the views of `flat`
            g_opt = torch.cuda.CUDAGraph()
            optname = "clip+AdamW on flat buffers, 3 launches" if args.optimizer == "flat" else "torch clip_grad_norm_ + fused AdamW"

            def lay_out(ordered):
                """one flat gradient buffer in the given parameter order (+ the flat optimizer on it)"""
                if args.optimizer == "flat":
                    fo = FlatClipAdamW(ordered, lr=1e-4, weight_decay=1e-5, max_norm=1.0)
                    inner._pack_cache.repoint()      # parameters moved into the flat buffer:
                    inner._pack_cache.refresh(inner.run_dtype)   # new pointer tables, built outside any capture
                    return fo.flat_g[:fo.n], fo.step
                A = FlatClipAdamW.ALIGN
                fl = torch.zeros(sum((p.numel() + A - 1) // A * A for p in ordered), dtype=torch.float32, device=dev)
                o = 0
                for p in ordered:
                    p.grad = fl[o:o + p.numel()].view_as(p)
                    o += (p.numel() + A - 1) // A * A
                return fl, opt_step
            if distributed and args.phases > 1:
                # backward cut into phases, one hipGraph each; the gradients a phase completed are
                # all-reduced (async RCCL) while the next phase's graph runs
                ps = PhasedStep(inner, model_loss)
                for p in used:
                    p.grad = torch.zeros_like(p)
                ps.forward(x, mask)          # eager dry run: which tape entry completes which parameter
                ps.backward(ps.n_entries, 0, True)
                # cut where the cumulative gradient bytes cross k/(K-1) * 85 %: the last phase (the
                # high-resolution encoder layers: few parameters, long compute) hides the exchange
                # of everything before it and leaves ~15 % of the bytes exposed
                cuts, groups = ps.plan([0.85 * (i + 1) / (args.phases - 1) for i in range(args.phases - 1)])
                ps.finish()
                flat, do_opt = lay_out([p for grp in groups for p in grp])   # ordered by phase: one collective each
                off, spans, A = 0, [], FlatClipAdamW.ALIGN
                for grp in groups:
                    k = sum((p.numel() + A - 1) // A * A for p in grp)
                    spans.append((off, off + k))
                    off += k
                assert off == flat.numel()
                graphs, pool = [], None
                for k in range(len(groups)):
                    gk = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gk, pool=pool, capture_error_mode=CAPTURE_MODE):
                        if k == 0:
                            static_loss = ps.forward(x, mask)
                            out = ps.outputs
                        ps.backward(cuts[k], cuts[k + 1], k == 0)
                    pool = gk.pool()
                    graphs.append(gk)
                ps.finish()

                def fb_replay():
                    for gk in graphs:
                        gk.replay()
                with torch.cuda.graph(g_opt, capture_error_mode=CAPTURE_MODE):
                    do_opt()

                def run_one():
                    works = []
                    for gk, (a0, a1) in zip(graphs, spans):
                        gk.replay()
                        works.append(dist.all_reduce(flat[a0:a1], op=dist.ReduceOp.AVG, async_op=True))
                    for w in works:
                        w.wait()
                    g_opt.replay()
                mb = [round((a1 - a0) * 4 / 2 ** 20, 1) for a0, a1 in spans]
                launch_mode = (f"{len(graphs)} hipGraphs (fwd + backward phases) with async RCCL all-reduce of "
                               f"{mb} MB overlapped with the next phase + hipGraph({optname})")
            else:
                flat, do_opt = lay_out(used)  # .grad = views of one buffer -> one collective
                g_fb = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_fb, capture_error_mode=CAPTURE_MODE):
                    out = inner(x)
                    static_loss = model_loss(out, mask)
                    static_loss.backward()   # every parameter gradient overwritten in place
                fb_replay = g_fb.replay
                with torch.cuda.graph(g_opt, capture_error_mode=CAPTURE_MODE):
                    do_opt()
                if distributed:
                    def run_one():
                        g_fb.replay()
                        dist.all_reduce(flat, op=dist.ReduceOp.AVG)
                        g_opt.replay()
                    launch_mode = f"hipGraph(fwd+bwd) + eager RCCL all-reduce + hipGraph({optname})"
                else:
                    def run_one():
                        g_fb.replay()
                        g_opt.replay()
                    launch_mode = f"hipGraph(fwd+bwd) + hipGraph({optname})"
            run_one()                 # one untimed replay
            torch.cuda.synchronize()
            if os.environ.get("UZ_BENCH_DEBUG"):
                for _ in range(4):
                    run_one()
                    torch.cuda.synchronize()
                    print("# debug loss", float(static_loss.item()), float(model_loss(out, mask).item()),
                          "grad norm", float(flat.norm().item()),
                          file=sys.stderr, flush=True)
        except Exception as e:  # noqa: BLE001
            if args.graph == "on":
                raise
            run_one = None
            launch_mode = "eager"
            torch.cuda.synchronize()
            for p in params:
                p.grad = None
            (net.module if isinstance(net, RcclDataParallel) else net).grads_in_place = False
            if isinstance(net, RcclDataParallel):
                net.module._grad_sink = net.reducer.push
                net.module._grad_sink_done = net.reducer.finish
            if rank == 0:
                print(f"# hipGraph capture failed ({type(e).__name__}: {e}); timing eager launches",
                      file=sys.stderr, flush=True)

    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if run_one is not None:
        for _ in range(args.steps):
            run_one()
        # the loss is re-evaluated eagerly from the last replay's logits: the library's multi-block mean
        # reduction inside a replayed hipGraph intermittently returned 0 on this stack (gradients are
        # unaffected: they do not depend on the reduced value)
        with torch.no_grad():
            loss = model_loss(out, mask)
    else:
        for _ in range(args.steps):
            loss = step(False)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    final_loss = float(loss.item())
    fb_graph_ms = None
    if run_one is not None:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            fb_replay()
        torch.cuda.synchronize()
        fb_graph_ms = (time.perf_counter() - t1) / args.steps * 1e3

    # per-launch HIP events (same process, same shapes, eager launches right after the timed steps)
    for p in params:          # the eager profiling steps own their gradients again
        p.grad = None
    (net.module if isinstance(net, RcclDataParallel) else net).grads_in_place = False
    ops.profile_begin()
    for _ in range(args.profile_steps):
        step(True)
    prof = ops.profile_end()
    nprof = max(args.profile_steps, 1)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * args.batch * args.steps / elapsed
        if fb_graph_ms is not None:
            fb_ms = fb_graph_ms
        else:
            fb_ms = sorted(a.elapsed_time(b) for a, b in fb_events)[len(fb_events) // 2] if fb_events else None
        # dominant kernel = the family with the largest summed duration
        dom_name, dom = max(prof.items(), key=lambda kv: kv[1]["ms"]) if prof else (None, None)
        roofline = None
        # HBM traffic of the dominant kernel: from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE in separate runs of this same command, FETCH_SIZE doubled as MI355X_MICROARCH.md
        # prescribes for gfx950) -- counters cannot be read from inside this process
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
                pmc = json.load(f)["kernels"]
            key = {"conv3x3_direct_bf16_bn128": "21conv3x3_direct_kernelIDF16bLi32ELi128ELb0EEEvNS_10DirectArgsE",
                   "wgrad3x3_bf16_128x128_3tap": "wgrad3x3_kernel<128, 128, 1, 3, 2, 4, 1>"}.get(dom_name)
            full = next((k for k in pmc if key and key in k), None)   # the table is keyed by the full kernel name
            if full is not None and args.model == "unet" and args.size == 256 and args.batch == 16:
                key = full
                traffic = {"hbm_read_mb_per_launch": pmc[key]["hbm_read_mb_per_launch_corrected"],
                           "hbm_write_mb_per_launch": pmc[key]["hbm_write_mb_per_launch"],
                           "algorithmic_mb_per_launch": round(dom["bytes"] / dom["launches"] / 2 ** 20, 2),
                           "source": "profiles/r01_pmc_traffic.json"}
        except (OSError, KeyError, ValueError):
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
            roofline.update({"launches_per_step": dom["launches"] // nprof,
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
            "roofline": roofline,
            "launch": launch_mode,
            "kernel_ms_per_step": {k: round(v["ms"] / nprof, 3) for k, v in sorted(prof.items())},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_batch, args.size, args.cpu_steps, args.model)
        print(json.dumps(line), flush=True)
    if world > 1 or args.force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
