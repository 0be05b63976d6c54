"""The reference's training step as hipGraph replays: ``GraphedStep``.

The reference's hot loop (unet_zoo/utils/training_loop.py:108-124) is, per batch,

    optimizer.zero_grad(); outputs = model(img); loss, dice = criterion(...); loss.backward()
    clip_grad_norm_(model.parameters(), 1.0); optimizer.step()

Launched eagerly from Python this engine is CPU-bound (17.7 ms of launches for 10 ms of GPU work on the UNet of
BASELINE configs[1]).  ``GraphedStep`` is the same step captured once and replayed:

    step = unet_zoo_amd.GraphedStep(model, lr=1e-4, weight_decay=1e-5)     # instead of optim.AdamW(...)
    loss = step(img, mask)                                                 # instead of lines 112-121
    outputs, dice = step.outputs, step.dice                                # device tensors, no host sync

* hipGraph 1..K: forward + loss + backward, the backward cut into K phases at tape positions; every kernel writes
  its parameter gradient in place into ONE flat fp32 buffer laid out phase by phase;
* with more than one rank (one process per GPU, ``torch.distributed`` over RCCL): after phase k's graph its span of
  the flat buffer is all-reduced (AVG, asynchronous) while phase k+1's graph runs — collectives stay outside graph
  capture and still overlap the backward (replaces ``nn.DataParallel``, multi_gpu.py:20-31);
* last hipGraph: ``clip_grad_norm_(max_norm)`` + AdamW on the flat parameter / gradient / moment buffers (three
  launches, ``optim.FlatClipAdamW``).

``step.loss``, ``step.dice`` and ``step.outputs`` are STATIC tensors of the captured graphs: the next call overwrites
them in place, so read (``.item()``) or ``.clone()`` what must outlive the step.

Loss: the default ``criterion="bce_dice"`` is the fused kernel of ``loss.py`` (BCEWithLogits + its gradient + the
Dice metric, one pass) inside the first graph.  Any callable ``criterion(outputs, target) -> loss`` works too; it is
evaluated EAGERLY between the forward graph and the backward graphs, because library reductions must not be captured
on this stack: a memset node of a replayed hipGraph writes its value only in the first replay, and torch's multi-block
reductions reset their semaphores with exactly such a node (tools/graph_canary.py).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple, Union

import torch
import torch.distributed as dist
import torch.nn as nn

from .graph import HipModule, PhasedStep
from .loss import loss_and_dice, loss_and_dice_direct
from .optim import FlatClipAdamW

# hipGraph capture checks only THIS thread's calls: the process-group watchdog thread polls its events concurrently
# (legal for it, but fatal to a capture in the default "global" mode)
CAPTURE_MODE = "thread_local"


_HIP = None


def _memset_nodes(graph: "torch.cuda.CUDAGraph") -> Optional[int]:
    """number of memset nodes in a captured graph (None when the runtime handle cannot be inspected).  On this stack a
    memset node of a replayed hipGraph writes its value in the FIRST replay only (tools/graph_canary.py):
    a library reduction that resets its semaphores with hipMemsetAsync gives right numbers once and silently wrong
    gradients afterwards -- so a capture that contains one is refused instead of trusted by convention."""
    global _HIP
    import ctypes
    try:
        raw = graph.raw_cuda_graph()
        if _HIP is None:
            _HIP = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_size_t(0)
        if _HIP.hipGraphGetNodes(ctypes.c_void_p(raw), None, ctypes.byref(n)) != 0:
            return None
        if n.value == 0:
            return 0
        nodes = (ctypes.c_void_p * n.value)()
        if _HIP.hipGraphGetNodes(ctypes.c_void_p(raw), nodes, ctypes.byref(n)) != 0:
            return None
        count = 0
        for i in range(n.value):
            t = ctypes.c_int(-1)
            if _HIP.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(t)) != 0:
                return None
            if t.value == 2:          # hipGraphNodeTypeMemset
                count += 1
        return count
    except Exception:                 # noqa: BLE001  (older torch without raw_cuda_graph, no libamdhip64: cannot inspect)
        return None


def _new_graph() -> "torch.cuda.CUDAGraph":
    try:
        return torch.cuda.CUDAGraph(keep_graph=True)     # keeps the hipGraph_t so that its nodes can be listed
    except TypeError:
        return torch.cuda.CUDAGraph()


def _check_capture(graph: "torch.cuda.CUDAGraph", what: str) -> None:
    n = _memset_nodes(graph)
    if n:
        raise RuntimeError(f"GraphedStep: the captured {what} contains {n} memset node(s); on this stack a replayed "
                           f"memset node acts only once (DESIGN.md 5a) -- a library reduction / zero-fill by "
                           f"hipMemsetAsync was captured.  Keep it out of the graph (criterion as a callable runs "
                           f"eagerly) or replace it with a kernel.")


def _unwrap(model: nn.Module) -> HipModule:
    inner = model.module if hasattr(model, "module") and isinstance(model.module, HipModule) else model
    if not isinstance(inner, HipModule):
        raise TypeError(f"GraphedStep needs a unet_zoo_amd model (HipModule), got {type(model).__name__}")
    return inner


class _ShapeGraphs:
    """everything captured for one (input shape, target shape)"""
    __slots__ = ("x", "t", "fwd", "phases", "gouts", "loss", "dice", "outputs", "pool", "ps")


class GraphedStep:
    def __init__(self, model: nn.Module, criterion: Union[str, Callable] = "bce_dice", *, lr: float = 1e-4,
                 betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-5,
                 max_norm: float = 1.0, phases: int = 5, process_group=None, data_parallel: Optional[bool] = None,
                 cu_reserve: Optional[int] = None, comm: str = "overlap", comm_dtype: Optional[torch.dtype] = None):
        """comm: when the gradient exchange of a data-parallel step runs -- "overlap": the span of every finished backward
        phase is all-reduced while the next phase's graph runs; "tail": ONE all-reduce of the whole flat buffer after the
        last phase.  The same graphs serve both; autotune_comm() times them on the job's own hardware and keeps the faster
        one (the persistent one-workgroup-per-CU grids and a collective's kernels compete for CUs: which schedule wins
        depends on the rank count and the fabric).
        comm_dtype: None / torch.float32 -- gradients are all-reduced in fp32 (the reference's DataParallel sums fp32 replicas'
        gradients, multi_gpu.py:28-31); torch.bfloat16 -- an OPTION that halves the bytes on the links: every rank rounds its
        span to bf16, a reduce-scatter by all-to-all, the shards are summed in fp32 ON ARRIVAL (fixed rank order) and averaged,
        rounded once to bf16 and all-gathered: every rank ends with the same bf16-representable mean; 2 (N-1)/N of the bf16
        span crosses each rank's links instead of 2 (N-1)/N of the fp32 one, and the direct all-to-all / all-gather pair uses
        all of a GPU's xGMI links at once.  Not the default: it changes the gradients by two bf16
        roundings.
        cu_reserve: CUs the library's persistent grids leave free (uz_set_cu_reserve; process-wide, applied here,
        before anything is planned or captured) so that the all-reduce of a finished gradient span can start while the
        next backward phase runs -- convolution / GEMM / weight-gradient kernels otherwise hold every CU with one
        160 KB workgroup until a kernel boundary.  None leaves the library's setting alone (default 0)."""
        self.model = _unwrap(model)
        if cu_reserve is not None:
            from . import _lib
            _lib.set_cu_reserve(cu_reserve)
        if isinstance(criterion, str):
            if criterion != "bce_dice":
                raise ValueError(f"unknown built-in criterion {criterion!r}; pass 'bce_dice' or a callable")
            self._fused_loss = True
            self._loss_fn = lambda out, t: loss_and_dice(out, t)[0]
        else:
            self._fused_loss = False
            self._loss_fn = criterion
        self.lr, self.betas, self.eps, self.weight_decay, self.max_norm = lr, betas, eps, weight_decay, max_norm
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # data_parallel=True with one rank keeps the multi-rank launch strategy (phases + collectives): a rehearsal
        self.distributed = (self.world > 1) if data_parallel is None else bool(data_parallel)
        if self.distributed and not dist.is_initialized():
            raise RuntimeError("data_parallel=True needs an initialised torch.distributed process group")
        self._nccl = self.distributed and dist.get_backend(process_group) == "nccl"
        self.n_phases = max(1, int(phases)) if self.distributed else 1
        if comm not in ("overlap", "tail"):
            raise ValueError(f"comm must be 'overlap' or 'tail', got {comm!r}")
        self.comm = comm
        if comm_dtype not in (None, torch.float32, torch.bfloat16):
            raise ValueError(f"comm_dtype must be None, torch.float32 or torch.bfloat16, got {comm_dtype!r}")
        self.comm_dtype = torch.bfloat16 if comm_dtype == torch.bfloat16 else torch.float32
        self._comm_stream: Optional[torch.cuda.Stream] = None
        self._comm_bufs: Dict[tuple, tuple] = {}
        self.opt: Optional[FlatClipAdamW] = None
        self._g_opt: Optional[torch.cuda.CUDAGraph] = None
        self._cuts: Optional[List[int]] = None
        self._spans: List[Tuple[int, int]] = []
        self._graphs: Dict[tuple, _ShapeGraphs] = {}
        self._cur: Optional[_ShapeGraphs] = None
        self.loss: Optional[torch.Tensor] = None
        self.dice: Optional[torch.Tensor] = None
        self.outputs = None
        self.steps_done = 0

    # ------------------------------------------------------------------ set-up (first call)
    def _dry_run(self, x: torch.Tensor, t: torch.Tensor):
        """One eager forward + backward that changes nothing: finds the parameters the graph reaches, the tape
        position that completes each of them, and warms the weight-layout cache."""
        m = self.model
        saved = [b.detach().clone() for b in m.buffers()]      # BatchNorm running statistics, counters
        rng = torch.cuda.get_rng_state(x.device)
        for p in m.parameters():
            p.grad = None
        ps = PhasedStep(m, self._loss_fn)
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ps.forward(x, t)
            ps.backward(ps.n_entries, 0, True)
            K = self.n_phases
            # cut where the cumulative gradient bytes cross k/(K-1) * 85 %: the last phase (the high-resolution
            # layers: few parameters, long compute) hides the exchange of everything before it
            fr = [0.85 * (i + 1) / (K - 1) for i in range(K - 1)] if K > 1 else []
            cuts, groups = ps.plan(fr)
            ps.finish()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize(x.device)
        for b, s in zip(m.buffers(), saved):
            b.copy_(s)
        torch.cuda.set_rng_state(rng, x.device)
        return cuts, groups

    def _setup(self, x: torch.Tensor, t: torch.Tensor) -> None:
        m = self.model
        if not m.training:
            raise RuntimeError("GraphedStep is the TRAINING step: call model.train() first "
                               "(evaluate with model.eval() and torch.no_grad() as the reference does)")
        cuts, groups = self._dry_run(x, t)
        ordered = [p for grp in groups for p in grp]
        if not ordered:
            raise RuntimeError("no parameter received a gradient")
        self.opt = FlatClipAdamW(ordered, lr=self.lr, betas=self.betas, eps=self.eps,
                                 weight_decay=self.weight_decay, max_norm=self.max_norm)
        # the parameters moved into the flat buffer: new pointer tables, built outside any capture
        m._pack_cache.repoint()
        m._pack_cache.refresh(m.run_dtype)
        A, off = FlatClipAdamW.ALIGN, 0
        self._spans = []
        for grp in groups:
            k = sum((p.numel() + A - 1) // A * A for p in grp)
            self._spans.append((off, off + k))
            off += k
        assert off == self.opt.n
        self._cuts = cuts
        if self.world > 1:
            # every rank starts from rank 0's parameters and buffers (one collective for all parameters); BatchNorm
            # running statistics then follow each rank's own shard, as the reference's DataParallel replicas do
            self._broadcast(self.opt.flat_p)
            for b in m.buffers():
                self._broadcast(b)
        self._capture_opt()

    def _capture_opt(self) -> None:
        self._g_opt = _new_graph()
        with torch.cuda.graph(self._g_opt, capture_error_mode=CAPTURE_MODE):
            self.opt.step()
        _check_capture(self._g_opt, "optimizer graph")

    def set_lr(self, lr: float) -> None:
        """Learning-rate change (the reference's DiceScheduler, utils/lr_scheduler.py:70-81): the rate is a kernel
        argument, so the three-launch optimizer graph is captured again."""
        self.lr = lr
        if self.opt is not None:
            self.opt.lr = lr
            torch.cuda.synchronize()
            self._capture_opt()

    def _capture(self, x: torch.Tensor, t: torch.Tensor) -> _ShapeGraphs:
        g = _ShapeGraphs()
        g.x, g.t = x, t
        ps = PhasedStep(self.model, self._loss_fn)
        cuts = self._cuts
        g.phases, g.fwd, g.gouts, g.dice, pool = [], None, None, None, None
        torch.cuda.synchronize()
        if self._fused_loss:
            dice_box = []

            def fused(out, tt):
                l, d = loss_and_dice(out, tt)
                dice_box.append(d)
                return l

            def fused_direct(out, tt):   # the same numbers and d(loss)/d(outputs) without autograd's three extra launches
                l, d, gouts = loss_and_dice_direct(out, tt)
                dice_box.append(d)
                return l, gouts
            fused.direct = fused_direct
            ps.loss_fn = fused
        else:
            # forward graph; the criterion runs eagerly on its static outputs; the backward graphs read static
            # d(loss)/d(output) buffers
            g.fwd = _new_graph()
            with torch.cuda.graph(g.fwd, capture_error_mode=CAPTURE_MODE):
                ps.emit(g.x)
            _check_capture(g.fwd, "forward graph")
            pool = g.fwd.pool()
            g.outputs = ps.outputs
            g.gouts = [torch.zeros_like(o) for o in ps._outs]
            ps.set_output_grads(g.gouts)
        for k in range(len(cuts) - 1):
            gk = _new_graph()
            with torch.cuda.graph(gk, pool=pool, capture_error_mode=CAPTURE_MODE):
                if k == 0 and self._fused_loss:
                    g.loss = ps.forward(g.x, g.t)
                    g.outputs = ps.outputs
                    g.dice = dice_box[-1].detach()
                ps.backward(cuts[k], cuts[k + 1], k == 0)
            _check_capture(gk, f"backward phase {k}")
            pool = gk.pool()
            g.phases.append(gk)
        g.ps = ps            # keeps the loss function / output leaves for the eager criterion
        if self._fused_loss:
            ps.finish()
        else:
            ps.eng.tape.clear()
            ps.eng = None
        g.pool = pool
        return g

    # ------------------------------------------------------------------ collectives
    def _broadcast(self, t: torch.Tensor) -> None:
        if self._nccl:
            dist.broadcast(t, src=0, group=self.pg)
        else:
            h = t.detach().cpu()
            dist.broadcast(h, src=0, group=self.pg)
            t.copy_(h)

    class _EventWait:
        """what the bf16 exchange hands back in place of a collective's work handle"""

        def __init__(self, ev):
            self.ev = ev

        def wait(self):
            torch.cuda.current_stream().wait_event(self.ev)

    def _exchange_bf16(self, buf: torch.Tensor):
        """buf (a span of the flat fp32 gradient buffer) <- bf16(mean over ranks of bf16(buf)): all-to-all of bf16 shards,
        fp32 sum on arrival in rank order, one rounding, all-gather -- on a side stream, so that the next backward phase
        keeps running; returns an object whose wait() orders the caller's stream behind it"""
        n, W = buf.numel(), self.world
        shard = (n + W - 1) // W
        key = (buf.data_ptr(), n)
        if key not in self._comm_bufs:
            dev = buf.device
            self._comm_bufs[key] = (torch.zeros(W * shard, dtype=torch.bfloat16, device=dev),
                                    torch.empty(W * shard, dtype=torch.bfloat16, device=dev),
                                    torch.empty(shard, dtype=torch.bfloat16, device=dev))
        send, recv, mine = self._comm_bufs[key]
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=buf.device)
        ready = torch.cuda.Event()
        ready.record()
        cs = self._comm_stream
        with torch.cuda.stream(cs):
            cs.wait_event(ready)
            send[:n].copy_(buf)                                   # fp32 -> bf16 (round to nearest even); the tail stays zero
            dist.all_to_all_single(recv, send, group=self.pg)     # shard r of every rank's span arrives at rank r
            mine.copy_(recv.view(W, shard).float().sum(0).mul_(1.0 / W))   # fp32 accumulate on arrival, fixed order, one rounding
            dist.all_gather_into_tensor(send, mine, group=self.pg)
            buf.copy_(send[:n])
            done = torch.cuda.Event()
            done.record(cs)
        return GraphedStep._EventWait(done)

    def _all_reduce_avg(self, buf: torch.Tensor):
        if self.comm_dtype == torch.bfloat16:
            if self._nccl:
                return self._exchange_bf16(buf)
            # other backends (gloo rehearsals): the same arithmetic staged through the host -- every rank's span rounded to
            # bf16, summed in fp32, averaged, rounded once
            h = buf.detach().to(torch.bfloat16).float().cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.pg)
            buf.copy_(h.mul_(1.0 / self.world).to(torch.bfloat16).float().to(buf.device))
            return None
        if self._nccl:
            return dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.pg, async_op=True)
        # other backends (gloo: CPU rehearsals and the single-GPU two-rank test): staged through the host
        h = buf.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.pg)
        buf.copy_(h.mul_(1.0 / self.world).to(buf.device))
        return None

    # ------------------------------------------------------------------ the step
    def forward_backward(self, x: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        """zero_grad + forward + loss + backward (+ gradient all-reduce): afterwards every ``p.grad`` (views of the flat
        buffer) holds this step's (rank-averaged) gradient.  Returns the loss as a device scalar."""
        key = (tuple(x.shape), tuple(target.shape))
        g = self._graphs.get(key)
        if g is None:
            # static input buffers of this shape (fp32 on the model's device, what `.float().to(device)` of
            # training_loop.py:109-110 produces); later calls copy into them -- host tensors included
            dev = next(self.model.parameters()).device
            sx = x.detach().to(device=dev, dtype=torch.float32, copy=True)
            st = target.detach().to(device=dev, dtype=torch.float32, copy=True)
            if self.opt is None:
                self._setup(sx, st)
            g = self._graphs[key] = self._capture(sx, st)
        self._cur = g
        if x.data_ptr() != g.x.data_ptr():
            g.x.copy_(x)
        if target.data_ptr() != g.t.data_ptr():
            g.t.copy_(target)
        if g.fwd is not None:
            g.fwd.replay()
            loss = g.ps.loss(g.t)                    # eager criterion; leaves d(loss)/d(output)
            for dst, src in zip(g.gouts, g.ps._gouts):
                if src is None:
                    dst.zero_()
                else:
                    dst.copy_(src)
            self.loss = loss
        else:
            self.loss = g.loss
        self.dice, self.outputs = g.dice, g.outputs
        if self.distributed and self.comm == "tail":
            for gk in g.phases:
                gk.replay()
            w = self._all_reduce_avg(self.opt.flat_g[:self._spans[-1][1]])
            if w is not None:
                w.wait()
        elif self.distributed:
            works = []
            for gk, (a0, a1) in zip(g.phases, self._spans):
                gk.replay()
                works.append(self._all_reduce_avg(self.opt.flat_g[a0:a1]))
            for w in works:
                if w is not None:
                    w.wait()
        else:
            for gk in g.phases:
                gk.replay()
        return self.loss

    def autotune_comm(self, x: torch.Tensor, target: torch.Tensor, steps: int = 6) -> Dict[str, float]:
        """Time `steps` whole training steps under each collective schedule ("overlap", "tail") on this job's ranks and
        keep the faster one; every rank takes the same decision (the slowest rank's time counts).  These are real
        optimizer steps on (x, target).  Returns {"overlap": ms, "tail": ms}."""
        if not self.distributed:
            return {}
        dev = next(self.model.parameters()).device
        out: Dict[str, float] = {}
        for mode in ("overlap", "tail"):
            self.comm = mode
            for _ in range(2):
                self(x, target)
            torch.cuda.synchronize(dev)
            if self.world > 1:
                dist.barrier(group=self.pg)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                self(x, target)
            e1.record()
            torch.cuda.synchronize(dev)
            ms = torch.tensor([e0.elapsed_time(e1) / steps], dtype=torch.float64, device=dev if self._nccl else "cpu")
            if self.world > 1:
                dist.all_reduce(ms, op=dist.ReduceOp.MAX, group=self.pg)
            out[mode] = float(ms.item())
        self.comm = "tail" if out["tail"] < 0.99 * out["overlap"] else "overlap"      # overlap unless the tail schedule wins by > 1 %
        return out

    def optimizer_step(self) -> None:
        """clip_grad_norm_(max_norm) + AdamW on the flat buffers (training_loop.py:120-121)"""
        self._g_opt.replay()
        self.steps_done += 1

    def __call__(self, x: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        loss = self.forward_backward(x, target)
        self.optimizer_step()
        return loss

    # ------------------------------------------------------------------ introspection
    @property
    def grad_norm(self) -> torch.Tensor:
        """total gradient norm the last optimizer step saw, before clipping (device scalar)"""
        return self.opt.last_grad_norm()

    @property
    def flat_grad(self) -> torch.Tensor:
        return self.opt.flat_g[:self.opt.n]

    def describe(self) -> str:
        """launch strategy in words (bench.py's `launch` field)"""
        k = len(self._cuts) - 1 if self._cuts else self.n_phases
        opt = "hipGraph(clip+AdamW on flat buffers, 3 launches)"
        crit = "" if self._fused_loss else "hipGraph(fwd) + eager criterion + "
        if not self.distributed:
            return f"{crit}hipGraph({'fwd+' if self._fused_loss else ''}bwd) + {opt}"
        mb = [round((a1 - a0) * 4 / 2 ** 20, 1) for a0, a1 in self._spans]
        if self.comm_dtype == torch.bfloat16:
            opt = "gradient exchange in bf16 (all-to-all, fp32 sum on arrival, all-gather) + " + opt
        if self.comm == "tail":
            return (f"{crit}{k} hipGraphs ({'fwd + ' if self._fused_loss else ''}backward phases), then ONE RCCL all-reduce of "
                    f"{round(sum(mb), 1)} MB + {opt}")
        return (f"{crit}{k} hipGraphs ({'fwd + ' if self._fused_loss else ''}backward phases) with async RCCL "
                f"all-reduce of {mb} MB overlapped with the next phase + {opt}")
