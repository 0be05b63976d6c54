"""Kernel-level execution engine: a model's ``emit`` method calls the block functions below, which
launch HIP kernels immediately (forward) and record a closure per block on a tape; ``backward``
replays the tape in reverse.  No tracing compiler, no per-op autograd graph: the whole network is
ONE ``torch.autograd.Function`` node (see ``graph_fn.py``), PyTorch only owns parameters, the
optimizer and the loss.

Step semantics follow the reference's training loop (unet_zoo/utils/training_loop.py:108-121):
train-mode BatchNorm uses batch statistics (biased variance) and updates running statistics with
momentum and the unbiased variance; eval mode uses the running statistics.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .ops import Act


def _round_up(a: int, b: int) -> int:
    return (a + b - 1) // b * b


class PackCache:
    """Kernel-layout copies of a model's weights, kept across steps and refreshed by ONE batched
    launch per forward (uz_pack_weights_batched) instead of one launch per tensor."""

    def __init__(self):
        self.entries: Dict[tuple, tuple] = {}   # key -> (param, mode, kpad, dtype, dst)
        self._table: Optional[torch.Tensor] = None
        self._total = 0
        self._n = 0
        self._table_dtype = None
        self._table3: Optional[torch.Tensor] = None
        self._n3 = 0
        self._tiles3 = 0

    def invalidate(self) -> None:
        self.entries.clear()
        self._table = None

    def repoint(self) -> None:
        """The parameters kept their identity but their storage moved (e.g. into a flat optimizer
        buffer): keep the packed destinations, rebuild only the pointer tables at the next refresh."""
        self._table = None

    def get(self, p: nn.Parameter, mode: int, kpad: int, dtype: torch.dtype) -> torch.Tensor:
        key = (id(p), mode, kpad, dtype)
        e = self.entries.get(key)
        if e is None:
            if mode == L.PACK_VEC_REPEAT:     # an fp32 vector kpad times over (ConvTranspose2d's bias per sub-pixel)
                dst = p.detach().float().repeat(kpad).contiguous()
            else:
                dst = ops.pack_weights(p.detach(), mode, dtype, kpad)   # packed now, batched from the next step on
            self.entries[key] = (p, mode, kpad, dtype, dst)
            self._table = None
            return dst
        return e[4]

    def refresh(self, dtype: torch.dtype) -> None:
        """Re-pack every registered tensor of `dtype` (one launch).  Called at the start of every
        forward: version counters are not a reliable change signal (fused / captured optimizer
        steps), and the launch costs less than 0.1 ms."""
        ents = [e for e in self.entries.values() if e[3] == dtype]
        if not ents:
            return
        if self._table is None or self._table_dtype != dtype:
            self._build_tables(ents, dtype)
        lib = L.load()
        if self._n3 > 0:
            L.check(lib.uz_pack_conv3x3_batched(L.dtype_code(dtype), self._table3.data_ptr(), self._n3,
                                                self._tiles3, L.stream_ptr()), "uz_pack_conv3x3_batched")
        if self._n > 0:
            L.check(lib.uz_pack_weights_batched(L.dtype_code(dtype), self._table.data_ptr(), self._n,
                                                self._total, L.stream_ptr()), "uz_pack_weights_batched")

    def _build_tables(self, ents, dtype) -> None:
        dev = ents[0][4].device
        # 3x3 conv weights with 32-divisible channel counts: one tiled item per parameter
        tiled = {}
        generic = []
        for (p, mode, kpad, _, dst) in ents:
            ok = (mode in (L.PACK_CONV_FWD, L.PACK_CONV_DGRAD) and p.dim() == 4 and tuple(p.shape[2:]) == (3, 3)
                  and p.shape[0] % 32 == 0 and p.shape[1] % 32 == 0)
            if ok:
                it = tiled.setdefault(id(p), [p, None, None])
                it[1 if mode == L.PACK_CONV_FWD else 2] = dst
            else:
                generic.append((p, mode, kpad, dst))
        arr3 = (L.Pack3x3Item * max(len(tiled), 1))()
        tb = 0
        for i, (p, df, dd) in enumerate(tiled.values()):
            arr3[i] = L.Pack3x3Item(p.data_ptr(), df.data_ptr() if df is not None else None,
                                    dd.data_ptr() if dd is not None else None, p.shape[0], p.shape[1], tb, 0)
            tb += (p.shape[0] // 32) * (p.shape[1] // 32)
        self._n3, self._tiles3 = len(tiled), tb
        self._table3 = torch.frombuffer(bytearray(bytes(arr3)), dtype=torch.uint8).to(dev)
        arr = (L.PackItem * max(len(generic), 1))()
        begin = 0
        for i, (p, mode, kpad, dst) in enumerate(generic):
            if mode == L.PACK_VEC_REPEAT:
                arr[i] = L.PackItem(p.data_ptr(), dst.data_ptr(), begin, mode, p.numel(), 1, kpad, 0, 0)
                begin += dst.numel()
                continue
            d0, d1 = p.shape[0], p.shape[1]
            T = p.numel() // (d0 * d1)
            co, ci = (d0, d1) if mode in (L.PACK_CONV_FWD, L.PACK_CONV_DGRAD, L.PACK_IM2COL) else (d1, d0)
            arr[i] = L.PackItem(p.data_ptr(), dst.data_ptr(), begin, mode, co, ci, T, kpad, 0)
            begin += dst.numel()
        self._table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        self._total, self._n, self._table_dtype = begin, len(generic), dtype

class ImageInput:
    """The network input for its first convolution: the fp32 NCHW image and -- only if somebody asks -- its im2col'd 3x3
    patches (an Act of [P][Kpad] in the run dtype: rounds 1-3's form of the first layer, still the fp32 run mode's)."""

    __slots__ = ("nchw", "kpad", "dtype", "_patches", "N", "H", "W")

    def __init__(self, nchw: torch.Tensor, kpad: int, dtype: torch.dtype):
        self.nchw, self.kpad, self.dtype, self._patches = nchw, kpad, dtype, None
        self.N, _, self.H, self.W = nchw.shape

    def patches(self) -> Act:
        if self._patches is None:
            self._patches = ops.im2col3x3_nchw(self.nchw, self.kpad, self.dtype)
        return self._patches


class Engine:
    # the first convolution of the bf16 run mode as direct kernels on the fp32 NCHW image (uz_conv_first.hip) instead of
    # im2col + GEMM + one-tap weight gradient (class-level: tools/ab_step.py times both ways in one process)
    direct_first_conv = True
    # nn.Linear weight gradients of a backward range: nothing reads them before the optimizer, so they are issued together
    # when the range ends (uz_wgrad_multi) instead of as ~100 small launches between the GEMMs of a swin_unet_v2 step
    defer_linear_wgrads = True
    # the BatchNorm-backward reduction of a sole-reader activation rides in the epilogue of the 3x3 input-gradient
    # convolution / of the ConvTranspose input-gradient GEMM that produces its gradient (class-level so that
    # tools/ab_step.py can time both ways in one process; no environment switch)
    fuse_bn_reduce = True
    fuse_bn_reduce_convt = True
    # the BatchNorm + ReLU between the two convolutions of a DoubleConv is not a pass of its own: the second convolution and
    # its weight gradient read the first one's raw output through it (uz_conv_igemm_xf / uz_wgrad_xf), where both kernels
    # take the shape; the normalised middle tensor then never exists (class-level: tools/ab_step.py times both ways)
    fold_bn_apply = True
    # ... and the same for the LAST decoder block in front of the 1x1 head (OutConv): the head's forward reads the raw tensor
    # through the map (uz_outconv_fwd_xf), its backward needs the raw tensor only (uz_outconv_bwd_bnred with x = NULL) --
    # one full-resolution apply pass and one full-resolution read fewer per step
    fold_bn_apply_head = True
    # the BatchNorm finalize launches (forward: statistics -> scale / shift; backward: partial rows -> totals) riding in the
    # launch of the element pass that consumes them (uz_bn_relu_add_apply_fin, uz_bn_relu_bwd_apply_fin; 36 launches of
    # ~5 us per unet step, 448 per u2net step).  OFF: measured slower (round 5, tools/ab_step.py, same box: unet 6.654 vs
    # 6.586 ms, u2net 16.05 vs 15.84 ms) -- the finalize stays on the critical path of its consumer either way, and the
    # hand-over inside one launch (coherent stores, flag add, poll, coherent table loads: four extra memory round trips)
    # costs more than the ~2 us of launch gap it removes.  Kept as a tested switch (tests/test_bn_fin_gpu.py).
    fuse_bn_finalize = False
    FIN_FLAGS = 4096
    # the BatchNorm element passes walk their tensors from the END: the convolution in front wrote its output ascending, its
    # last ~100 MB are still in the Infinity Cache; the pass then ends at the low addresses, where the next (ascending)
    # convolution starts reading (class-level: tools/ab_step.py times both ways)
    reverse_element_passes = True

    def __init__(self, dtype: torch.dtype, device: torch.device, training: bool, record: bool,
                 grad_sink: Optional[Callable[[nn.Parameter, torch.Tensor], None]] = None,
                 pack_cache: Optional[PackCache] = None, grads_in_place: bool = False):
        self.dtype = dtype
        self.device = device
        self.training = training
        self.record = record          # build a tape for backward
        self.tape: List[Callable[[], None]] = []
        self.param_grads: Dict[nn.Parameter, torch.Tensor] = {}
        self.grad_sink = grad_sink    # called as soon as a parameter gradient has been launched
        self.bk = 64 if dtype == torch.bfloat16 else 32  # K-slab of the igemm kernel (128 bytes)
        self._bn_channels = 0
        self._sums: Optional[torch.Tensor] = None
        self._sums_used = 0
        self._cache = pack_cache if pack_cache is not None else PackCache()
        # write every parameter gradient straight into the existing p.grad storage (overwrite, no
        # accumulation, no temporaries): used with pre-allocated flat gradient buffers / hipGraphs
        self.grads_in_place = grads_in_place
        self._heads: List[Tuple[Callable[..., None], int]] = []   # (backward fn, number of outputs)
        # (tape position, parameter) in the order gradients are produced; position len(tape) = heads
        self.grad_log: List[Tuple[int, nn.Parameter]] = []
        self._pending_colsums: List[Tuple[int, nn.Parameter, Act]] = []   # bias gradients of the running backward range
        self._pending_rowsums: List[Tuple[int, Callable[[], None]]] = []
        self._pending_wgrads: List[Tuple[int, nn.Parameter, Act, Act]] = []
        self._rowsum_items: list = []
        self._cpb: Dict[nn.Module, dict] = {}     # position_biases(): WindowAttention module -> batched entry
        self._bn_counters: List[torch.Tensor] = []  # num_batches_tracked of the train-mode BatchNorms seen
        self._cur_entry = -1
        # one int32 per fused finalize of this forward + backward, zero when its launch starts: cleared by ONE fill kernel
        # per step (created with the first request; a kernel, not a memset node -- DESIGN 5a)
        self._fin_flags: Optional[torch.Tensor] = None
        self._fin_used = 0

    def _fin_flag(self) -> Optional[torch.Tensor]:
        """the next zeroed flag of this step's arena, or None when the finalize launches are to stay launches of their own"""
        if not self.fuse_bn_finalize:
            return None
        if self._fin_flags is None:
            self._fin_flags = torch.zeros(self.FIN_FLAGS, dtype=torch.int32, device=self.device)
        if self._fin_used >= self.FIN_FLAGS:
            return None
        self._fin_used += 1
        return self._fin_flags[self._fin_used - 1:self._fin_used]

    # ------------------------------------------------------------------ buffers
    def new_act(self, N, H, W, C, needs_grad=True) -> Act:
        return ops.new_act(N, H, W, C, self.dtype, self.device, needs_grad)

    def new_cat(self, N, H, W, channels: Sequence[int]) -> Tuple[Act, List[Act]]:
        """One buffer holding a channel-concat; producers write their part in place."""
        full = self.new_act(N, H, W, sum(channels))
        parts, o = [], 0
        for c in channels:
            parts.append(full.window(o, c))
            o += c
        full.parts = parts
        return full, parts

    def _dst(self, p: nn.Parameter) -> Optional[torch.Tensor]:
        """Destination tensor for p's gradient in in-place mode (None: allocate a fresh one)."""
        if self.grads_in_place and p.grad is not None and p.grad.is_contiguous() and p.grad.dtype == torch.float32:
            return p.grad
        return None

    def _give_grad(self, p: nn.Parameter, g: Optional[torch.Tensor]) -> None:
        if not isinstance(p, nn.Parameter):
            return      # constant stand-ins (gamma = 1 / beta = 0 of an affine-free BatchNorm): nothing to train
        self.grad_log.append((self._cur_entry, p))
        if self.grads_in_place and p.grad is not None:
            if g is None:
                return                               # analytically zero and p.grad was never touched
            if p in self.param_grads:
                # a parameter used by two layers of the graph: its second contribution.  Kernels that write into
                # p.grad itself (uz_wgrad with out=p.grad, the batched column sums) would have overwritten the first
                # one -- there is no accumulate form of them -- so that case is refused rather than silently halved;
                # a contribution that arrives in its own tensor is added (eager mode sums them the same way).
                if g.data_ptr() == p.grad.data_ptr():
                    raise RuntimeError("a parameter shared by two layers received its second gradient in place: "
                                       "shared parameters are not supported by the in-place (graphed) backward")
                p.grad.add_(g.reshape(p.grad.shape))
            elif g.data_ptr() != p.grad.data_ptr():
                p.grad.copy_(g.reshape(p.grad.shape))
            self.param_grads[p] = None               # autograd must not add it a second time
            if self.grad_sink is not None:
                self.grad_sink(p, p.grad)
            return
        if g is None:
            g = torch.zeros(p.shape, dtype=torch.float32, device=self.device)
        if p in self.param_grads:
            self.param_grads[p] = self.param_grads[p] + g
        else:
            self.param_grads[p] = g
        if self.grad_sink is not None:
            self.grad_sink(p, self.param_grads[p])

    def _bias_grad(self, p: nn.Parameter, g: Act) -> None:
        """d(loss)/d(bias) = column sums of the output gradient g: collected and computed for the whole backward
        range in two launches (uz_colsum_batched) when backward_range() ends; g stays alive until then."""
        self._pending_colsums.append((self._cur_entry, p, g))

    def _linear_wgrad(self, p: nn.Parameter, g: Act, x: Act) -> None:
        """d(loss)/d(weight) of y = W x from the output gradient g and the input x, now or with the rest of the range"""
        if self.defer_linear_wgrads and self.dtype == torch.bfloat16:
            self._pending_wgrads.append((self._cur_entry, p, g, x))
        else:
            self._give_grad(p, ops.wgrad(g, x, tuple(p.shape), ntaps=1, out=self._dst(p)))

    def _after_rowsums(self, fn: Callable[[], None]) -> None:
        """run fn (the _give_grad calls of parameter gradients left as per-workgroup partial rows) once the batched row
        sums of this backward range have been enqueued; ops take `defer=self._rowsum_items`"""
        self._pending_rowsums.append((self._cur_entry, fn))

    def _flush_rowsums(self) -> None:
        if self._pending_rowsums:
            items, self._rowsum_items = self._rowsum_items, []
            after, self._pending_rowsums = self._pending_rowsums, []
            ops.sum_rows_f32_batched(items)
            cur = self._cur_entry
            for entry, fn in after:
                self._cur_entry = entry
                fn()
            self._cur_entry = cur

    def _flush_colsums(self) -> None:
        self._flush_rowsums()
        wpend, self._pending_wgrads = self._pending_wgrads, []
        if wpend:
            wouts = []
            for _, p, _, _ in wpend:
                dst = self._dst(p)
                wouts.append(dst if dst is not None else torch.empty(tuple(p.shape), dtype=torch.float32, device=self.device))
            ops.wgrad_multi([(g, x, o, 1, L.TAPS_CONV, 1) for (_, _, g, x), o in zip(wpend, wouts)])
            for (entry, p, _, _), o in zip(wpend, wouts):
                self._cur_entry = entry
                self._give_grad(p, o)
        pend, self._pending_colsums = self._pending_colsums, []
        if not pend:
            return
        outs = []
        for _, p, g in pend:
            dst = self._dst(p)
            outs.append(dst if dst is not None else torch.empty(g.C, dtype=torch.float32, device=self.device))
        ops.colsum_batched([(g, o) for (_, _, g), o in zip(pend, outs)])
        for (entry, p, _), o in zip(pend, outs):
            self._cur_entry = entry          # the gradient belongs to the tape entry that produced g (PhasedStep.plan)
            self._give_grad(p, o)

    def _pack(self, p: nn.Parameter, mode: int, kpad: int = 0) -> torch.Tensor:
        return self._cache.get(p, mode, kpad, self.dtype)

    def _bn_sums(self, C: int) -> torch.Tensor:
        if self._sums is None:
            self._sums = torch.empty(2 * self._bn_channels, dtype=torch.float64, device=self.device)
            self._sums_used = 0
        s = self._sums[self._sums_used:self._sums_used + 2 * C].view(2, C)
        self._sums_used += 2 * C
        return s

    def _sum_grads(self, a: Act, limit: int) -> List[Act]:
        """Return at most `limit` same-resolution gradient sources (pre-adding the rest: a tensor read by many
        concats, UNet++'s dense skips, collects one gradient per reader)."""
        gs = list(a.grads)
        while len(gs) > limit:
            x, y = gs.pop(), gs.pop()
            t = self.new_act(x.N, x.H, x.W, x.C)
            ops.add_acts(x, y, t)
            gs.append(t)
        return gs

    # ------------------------------------------------------------------ blocks
    def input_im2col(self, x: torch.Tensor) -> "ImageInput":
        """Network input (N,C,H,W) fp32 for the first convolution: the image itself (the direct first-convolution kernels of
        the bf16 run mode read it as it is) and, on demand, its 3x3 patches [P][Kpad] (`ImageInput.patches()`: fp32 run
        mode, channel counts the direct kernels do not take, consumers other than Conv -> BN -> ReLU)."""
        L.require_cuda(x)
        if x.dim() != 4:
            raise ValueError(f"expected a (N, C, H, W) input, got shape {tuple(x.shape)}")
        return ImageInput(x.contiguous().float(), _round_up(9 * x.shape[1], self.bk), self.dtype)

    def conv_bn_relu(self, x: Act, conv: nn.Conv2d, bn: nn.BatchNorm2d, *, out: Optional[Act] = None,
                     pool: bool = False, im2col: bool = False, upsample: bool = False,
                     residual: Optional[Act] = None, pool_ceil: bool = False, relu: bool = True,
                     stat_repeat: int = 1, sole_reader: bool = False,
                     defer_apply: Optional[nn.Conv2d] = None) -> Tuple[Act, Optional[Act]]:
        """[nearest x2 upsample ->] Conv3x3(+bias) -> BatchNorm2d -> ReLU [-> MaxPool2d(2,2)].

        Reference: DoubleConv / ConvBlock / REBNCONV halves (common_layers.py:28-33, 47-56;
        u2net.py:10-17), DownSample's pool (common_layers.py:90-95) and UpConvBlock
        (common_layers.py:69-76: the upsampled tensor is never materialised, the convolution reads
        the half-resolution input at (h>>1, w>>1)).  `residual` is added AFTER the ReLU and before
        the pool (the RSU tail `hx1d + hxin`, u2net.py:74).  relu=False: Conv -> BatchNorm only
        (Conv2d_batchnorm(activation='None'), multiresunet.py:26-31).  stat_repeat = k: the reference applies this
        layer to a tensor in which every pixel of x occurs k times (a nearest-neighbour upsampling in front of a 1x1
        convolution, uctransnet.py:75-80); batch mean, biased variance and all gradients are those of x, only the
        sample count of the running variance's unbiased factor is k times larger.  sole_reader: the caller states
        that nothing but this convolution reads x (the middle tensor of a DoubleConv): if x is itself the output of a
        Conv -> BN -> ReLU, the first pass of ITS BatchNorm backward then rides in the epilogue of this layer's
        input-gradient convolution (uz_conv_igemm_bnred) instead of re-reading the gradient.  defer_apply = the 3x3
        convolution that is the ONLY reader of this layer's output (DoubleConv's second): if that convolution and its
        weight gradient can read the raw output through BatchNorm + ReLU (Engine.fold_bn_apply), the apply pass is not run
        and the returned activation is LAZY (Act.lazy = (scale, shift) over the raw buffer); the caller hands it to
        conv_bn_relu(..., sole_reader=True) and to nothing else.  Returns (act, pooled)."""
        N, H, W = x.N, x.H, x.W
        if upsample:
            H, W = 2 * H, 2 * W
        tmode = L.TAPS_CONV_UP2 if upsample else L.TAPS_CONV
        Cout = conv.out_channels
        xf = getattr(x, "lazy", None)      # x is the raw output of a convolution, to be read through its BatchNorm + ReLU
        assert xf is None or (sole_reader and not im2col and conv.kernel_size == (3, 3) and conv.dilation == (1, 1)), \
            "a lazy activation goes to the 3x3 convolution it was deferred for"
        dil = conv.dilation[0]
        image = None       # the fp32 NCHW input when the direct first-convolution kernels take this layer
        if im2col and isinstance(x, ImageInput):
            if (self.direct_first_conv and conv.kernel_size == (3, 3) and dil == 1 and conv.padding == (1, 1)
                    and conv.stride == (1, 1) and ops.conv_first_supported(self.dtype, x.nchw.shape[1], Cout)):
                image = x.nchw
                # the backward's weight gradient reads the caller's OWN tensor (contiguous fp32 input: no copy was made): an
                # in-place change of the input between forward and backward is caught there instead of giving a wrong dW
                image_version = image._version
            else:
                x = x.patches()
        if image is not None:
            wp, ntaps = None, 1
        elif im2col:
            wp = self._pack(conv.weight, L.PACK_IM2COL, x.C)
            ntaps = 1
        else:
            assert conv.kernel_size in ((3, 3), (1, 1)) and conv.in_channels == x.C
            wp = self._pack(conv.weight, L.PACK_CONV_FWD)
            ntaps = 9 if conv.kernel_size == (3, 3) else 1     # 1x1 + BN + ReLU: unet_transformer.py:150-177
            assert ntaps == 9 or not upsample
        y = self.new_act(N, H, W, Cout)
        bias = conv.bias.detach() if conv.bias is not None else None
        if image is not None:
            stats = ops.conv_first_fwd(image, conv.weight.detach(), bias, y, self.training)
        else:
            stats = ops.conv_igemm(x, wp, bias, y, ntaps=ntaps, dil=dil, taps_mode=tmode,
                                   want_stats=self.training, xform=xf)
        lazy = False
        if (defer_apply is not None and self.fold_bn_apply and out is None and not pool and residual is None and relu
                and stat_repeat == 1 and self.dtype == torch.bfloat16 and defer_apply.kernel_size == (3, 3)
                and defer_apply.dilation == (1, 1) and defer_apply.stride == (1, 1) and defer_apply.in_channels == Cout):
            # both kernels of the reader must take the map (the weight gradient's L operand has the reader's channels)
            c2 = defer_apply.out_channels
            lazy = (ops.conv_xform_supported(y, c2, c2)
                    and ops.wgrad_xform_shapes_supported(N, H, W, c2, c2, Cout, y.ld, self.dtype))
        elif (defer_apply is not None and self.fold_bn_apply and self.fold_bn_apply_head and out is None and not pool
                and residual is None and relu and stat_repeat == 1 and defer_apply.kernel_size == (1, 1)
                and defer_apply.in_channels == Cout and (self.training or not self.record)):
            # the reader is the 1x1 head (out_conv): its forward reads the raw tensor through the map, its backward takes the
            # gradient AND this BatchNorm's first backward pass from the raw tensor alone (uz_outconv_bwd_bnred, x = NULL) --
            # which needs batch statistics on the tape (training), or no tape at all
            lazy = self.fuse_bn_reduce_convt and ops.outconv_xform_supported(y, defer_apply.out_channels)
        # the finalize inside the apply pass's launch: training statistics, an apply pass to ride in
        fin = self._fin_flag() if (self.training and not lazy and stat_repeat == 1) else None
        vec = None
        if self.training:
            mom = bn.momentum if bn.momentum is not None else 0.1
            if fin is None:
                vec = ops.bn_finalize(stats if stat_repeat == 1 else stats * float(stat_repeat), y.P * stat_repeat,
                                      bn.weight.detach(), bn.bias.detach(), bn.eps, mom, bn.running_mean, bn.running_var)
            if bn.num_batches_tracked is not None:
                self._bn_counters.append(bn.num_batches_tracked)   # bumped together in finish_forward()
        else:
            vec = self._bn_vectors(bn, None, y.P)      # running statistics: (scale, shift, mean, invstd)
        if lazy:
            act = ops.Act(y.buf, y.off, y.C, N, H, W, True)
            act.lazy = (vec[0], vec[1])
            pooled = None
        else:
            act = out if out is not None else self.new_act(N, H, W, Cout)
            if pool and pool_ceil:
                pooled = self.new_act(N, (H + 1) // 2, (W + 1) // 2, Cout)
            else:
                pooled = self.new_act(N, H // 2, W // 2, Cout) if pool else None
            assert relu or residual is None
            if fin is not None:
                vec = ops.bn_relu_apply_fin(y, stats, y.P, bn.weight.detach(), bn.bias.detach(), bn.eps, mom, bn.running_mean,
                                            bn.running_var, fin, act, pooled, residual, pool_ceil, relu=relu)
            else:
                ops.bn_relu_apply(y, vec[0], vec[1], act, pooled, residual, pool_ceil, relu=relu,
                                  reverse=self.reverse_element_passes)
        if self.record and self.training and relu and pooled is None and residual is None and stat_repeat == 1:
            act.bn_src = (y, vec)      # what a sole reader's input-gradient kernel needs (see sole_reader)

        frozen = not self.training     # model.eval() + backward(): BatchNorm is an affine map of constants
        if self.record:
            self._bn_channels += Cout

            def bwd():
                gs = self._sum_grads(act, 2)
                gp = self._sum_grads(pooled, 1)[0] if (pooled is not None and pooled.grads) else None
                g0 = gs[0] if len(gs) > 0 else None
                g1 = gs[1] if len(gs) > 1 else None
                if g0 is None and gp is None:
                    return  # nothing downstream used this activation
                if residual is not None:
                    # the residual branch needs the TOTAL gradient of act as one tensor; the pool's
                    # argmax is over act (= relu + residual), not over relu(bn(y))
                    if gp is not None or g1 is not None:
                        tot = self.new_act(N, H, W, Cout)
                        ops.pool_grad_combine(act, g0, g1, gp, tot, pool_ceil)
                        g0, g1, gp = tot, None, None
                    if residual.needs_grad:
                        residual.add_grad(g0)
                dy = self.new_act(N, H, W, Cout)
                dgamma, dbeta = self._dst(bn.weight), self._dst(bn.bias)
                if dgamma is None:
                    dgamma = torch.empty(Cout, dtype=torch.float32, device=self.device)
                if dbeta is None:
                    dbeta = torch.empty(Cout, dtype=torch.float32, device=self.device)
                # g0 came from a sole reader's input-gradient kernel with the reduction already done in its epilogue
                parts = getattr(g0, "bn_partials", None) if (g1 is None and gp is None and residual is None) else None
                ops.bn_relu_bwd(y, vec, g0, g1, gp, self._bn_sums(Cout), dy, dgamma, dbeta, pool_ceil, relu=relu,
                                partials=parts, frozen=frozen, fin_flag=None if frozen else self._fin_flag(),
                                reverse=self.reverse_element_passes)
                self._give_grad(bn.weight, dgamma)
                self._give_grad(bn.bias, dbeta)
                if conv.bias is not None:
                    if frozen:
                        self._bias_grad(conv.bias, dy)     # running statistics do not cancel a per-channel constant
                    else:
                        # d(bias) = sum_p dy == 0 analytically under train-mode BN (the batch mean
                        # removes any per-channel constant); the reference's value is rounding noise.
                        self._give_grad(conv.bias, None)
                if image is not None:
                    if image._version != image_version:
                        raise RuntimeError("the network input was modified in place between forward and backward: the first "
                                           "convolution's weight gradient reads it (pass a copy, or finish backward first)")
                    self._give_grad(conv.weight, ops.conv_first_wgrad(image, dy, out=self._dst(conv.weight)))
                elif im2col:
                    dwp = ops.wgrad(dy, x, (Cout, x.C), ntaps=1)
                    cin = conv.in_channels
                    dw = dwp[:, :9 * cin].reshape(Cout, 9, cin).permute(0, 2, 1).reshape(conv.weight.shape)
                    self._give_grad(conv.weight, dw.contiguous())
                else:
                    if ntaps == 1 and tmode == L.TAPS_CONV:
                        self._linear_wgrad(conv.weight, dy, x)      # a 1x1 convolution is a Linear layer on the pixel list
                    else:
                        self._give_grad(conv.weight, ops.wgrad(dy, x, tuple(conv.weight.shape), ntaps=ntaps,
                                                               dil=dil, taps_mode=tmode,
                                                               out=self._dst(conv.weight), xform=xf))
                    if x.needs_grad and upsample:
                        # gradient of the (virtual) upsampled tensor, then 2x2 sums
                        du = self.new_act(N, H, W, x.C)
                        ops.conv_igemm(dy, self._pack(conv.weight, L.PACK_CONV_DGRAD), None, du,
                                       ntaps=9, dil=dil)
                        dx = self.new_act(x.N, x.H, x.W, x.C)
                        ops.sum2x2(du, dx)
                        x.add_grad(dx)
                    elif x.needs_grad:
                        dx = self.new_act(N, H, W, x.C)
                        # per-channel sums of dx come for free from the kernel's statistics
                        # epilogue; a ConvTranspose2d feeding x takes its bias gradient from them
                        want = x.parts is not None
                        src = getattr(x, "bn_src", None) if (sole_reader and not want and self.fuse_bn_reduce) else None
                        part = ops.conv_igemm(dy, self._pack(conv.weight, L.PACK_CONV_DGRAD), None, dx,
                                              ntaps=ntaps, dil=dil, want_stats=want, bnred=src)
                        if want:
                            dx.colsums = (part, 0)
                        elif src is not None and part is not None:
                            dx.bn_partials = part
                        x.add_grad(dx)

            self.tape.append(bwd)
        return act, pooled

    def attention_gate(self, g: Act, x: Act, blk: nn.Module, out: Act) -> Act:
        """out = x * sigmoid(BN(W_psi relu(BN(W_g g) + BN(W_x x)))), written into its concat slot.
        Reference: AttentionBlock.forward (attention_unet.py:34-40); `blk` owns w_g, w_x, psi."""
        conv_g, bn_g = blk.w_g[0], blk.w_g[1]
        conv_x, bn_x = blk.w_x[0], blk.w_x[1]
        conv_q, bn_q = blk.psi[0], blk.psi[1]
        N, H, W, P = x.N, x.H, x.W, x.P
        Fi = conv_g.out_channels
        assert (g.N, g.H, g.W) == (N, H, W) and conv_q.out_channels == 1

        def conv1x1_bn(src: Act, conv: nn.Conv2d, bn: nn.BatchNorm2d):
            raw = self.new_act(N, H, W, conv.out_channels)
            st = ops.conv_igemm(src, self._pack(conv.weight, L.PACK_CONV_FWD),
                                conv.bias.detach() if conv.bias is not None else None, raw, ntaps=1,
                                want_stats=self.training)
            return raw, self._bn_vectors(bn, st, P)

        g1, vec_g = conv1x1_bn(g, conv_g, bn_g)
        x1, vec_x = conv1x1_bn(x, conv_x, bn_x)
        wpsi = conv_q.weight.detach().reshape(Fi)
        q, part = ops.attn_psi_fwd(g1, x1, vec_g, vec_x, wpsi,
                                   conv_q.bias.detach() if conv_q.bias is not None else None)
        vec_q = self._bn_vectors(bn_q, part, P)
        ops.attn_gate_fwd(x, q, vec_q, out)

        if self.record:
            frozen = not self.training     # model.eval() + backward(): BatchNorms on their running statistics

            def bwd():
                dout = self._sum_grads(out, 1)
                if not dout:
                    return
                dxd = self.new_act(N, H, W, x.C)
                dz, a01 = ops.attn_bwd_psi(dout[0], x, q, vec_q, dxd)
                dg1 = self.new_act(N, H, W, Fi)
                dx1 = self.new_act(N, H, W, Fi)
                tot = ops.attn_bwd_branches(g1, x1, q, dz, wpsi, vec_g, vec_x, vec_q, a01, dg1, dx1, frozen=frozen)
                totf, a01f = tot.float(), a01.float()
                self._give_grad(bn_q.weight, a01f[1:2])
                self._give_grad(bn_q.bias, a01f[0:1])
                self._give_grad(bn_g.weight, totf[Fi:2 * Fi])
                self._give_grad(bn_g.bias, totf[0:Fi])
                self._give_grad(bn_x.weight, totf[2 * Fi:3 * Fi])
                self._give_grad(bn_x.bias, totf[0:Fi].clone())
                self._give_grad(conv_q.weight, totf[3 * Fi:4 * Fi].reshape(conv_q.weight.shape))
                if frozen:     # no batch statistics behind them: the biases receive the column sums of their outputs' gradients
                    if conv_q.bias is not None:
                        self._give_grad(conv_q.bias, totf[4 * Fi:4 * Fi + 1].reshape(conv_q.bias.shape))
                    if conv_g.bias is not None:
                        self._bias_grad(conv_g.bias, dg1)
                    if conv_x.bias is not None:
                        self._bias_grad(conv_x.bias, dx1)
                else:
                    for c in (conv_q, conv_g, conv_x):
                        if c.bias is not None:
                            self._give_grad(c.bias, None)   # a train-mode BatchNorm follows: analytically zero
                self._linear_wgrad(conv_g.weight, dg1, g)      # 1x1 convolutions: with the deferred set (uz_wgrad_multi)
                self._linear_wgrad(conv_x.weight, dx1, x)
                if g.needs_grad:
                    dg = self.new_act(N, H, W, g.C)
                    ops.conv_igemm(dg1, self._pack(conv_g.weight, L.PACK_CONV_DGRAD), None, dg, ntaps=1)
                    g.add_grad(dg)
                if x.needs_grad:
                    dxx = self.new_act(N, H, W, x.C)
                    ops.conv_igemm(dx1, self._pack(conv_x.weight, L.PACK_CONV_DGRAD), None, dxx, ntaps=1)
                    x.add_grad(dxx)
                    x.add_grad(dxd)

            self.tape.append(bwd)
        return out

    def _bn_vectors(self, bn: nn.BatchNorm2d, stats: Optional[torch.Tensor], count: int) -> torch.Tensor:
        """(scale, shift, mean, invstd) rows for a BatchNorm: batch statistics + running-stat update
        in training, running statistics in eval."""
        if self.training:
            mom = bn.momentum if bn.momentum is not None else 0.1
            vec = ops.bn_finalize(stats, count, bn.weight.detach(), bn.bias.detach(), bn.eps, mom,
                                  bn.running_mean, bn.running_var)
            if bn.num_batches_tracked is not None:
                self._bn_counters.append(bn.num_batches_tracked)   # bumped together in finish_forward()
            return vec
        v2 = ops.bn_eval_scale(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps)
        return torch.cat([v2, bn.running_mean.reshape(1, -1), torch.rsqrt(bn.running_var + bn.eps).reshape(1, -1)])

    # ------------------------------------------------------------------ pre-activation residual pieces (resunet)
    def bn_act(self, x: Act, bn: nn.BatchNorm2d, relu: bool = True, pool: bool = False):
        """BatchNorm2d [+ ReLU] of a tensor that is NOT a convolution output, so nobody has its statistics yet:
        `ResidualConv.conv_block[0:2]` and `conv_skip[1]` (common_layers.py:186-187, :195).  One statistics pass
        (uz_colstats), then the same finalize / apply / two-pass backward kernels as the fused conv-BN-ReLU.
        pool=True: also MaxPool2d(2, 2) of the result, returned as (act, pooled) (multiresunet.py:83, :201-202)."""
        C = bn.num_features
        assert x.C == C
        vec = self._bn_vectors(bn, ops.colstats(x) if self.training else None, x.P)
        act = self.new_act(x.N, x.H, x.W, C)
        pooled = self.new_act(x.N, x.H // 2, x.W // 2, C) if pool else None
        ops.bn_relu_apply(x, vec[0], vec[1], act, pooled, relu=relu)
        frozen = not self.training
        if self.record:
            self._bn_channels += C

            def bwd():
                gs = self._sum_grads(act, 2)
                gp = self._sum_grads(pooled, 1)[0] if (pooled is not None and pooled.grads) else None
                if not gs and gp is None:
                    return
                dx = self.new_act(x.N, x.H, x.W, C)
                dgamma, dbeta = self._dst(bn.weight), self._dst(bn.bias)
                if dgamma is None:
                    dgamma = torch.empty(C, dtype=torch.float32, device=self.device)
                if dbeta is None:
                    dbeta = torch.empty(C, dtype=torch.float32, device=self.device)
                ops.bn_relu_bwd(x, vec, gs[0] if gs else None, gs[1] if len(gs) > 1 else None, gp, self._bn_sums(C), dx,
                                dgamma, dbeta, relu=relu, frozen=frozen)
                self._give_grad(bn.weight, dgamma)
                self._give_grad(bn.bias, dbeta)
                if x.needs_grad:
                    x.add_grad(dx)

            self.tape.append(bwd)
        return (act, pooled) if pool else act

    def conv_plain(self, x: Act, conv: nn.Conv2d, *, im2col: bool = False) -> Act:
        """Conv2d k3 p1 / k1 p0, stride 1, WITHOUT a following BatchNorm: the block tails and skip convolutions of
        resunet (resunet.py:28-32, common_layers.py:191, :194 after uz-side subsampling).  `im2col`: x holds the
        3x3 patches of the network input (input_im2col)."""
        Cout = conv.out_channels
        k = conv.kernel_size[0]
        assert conv.kernel_size in ((3, 3), (1, 1)) and conv.dilation == (1, 1)
        if im2col and isinstance(x, ImageInput):
            x = x.patches()
        if im2col:
            wp, ntaps = self._pack(conv.weight, L.PACK_IM2COL, x.C), 1
        else:
            assert conv.in_channels == x.C
            wp, ntaps = self._pack(conv.weight, L.PACK_CONV_FWD), k * k
        y = self.new_act(x.N, x.H, x.W, Cout)
        ops.conv_igemm(x, wp, conv.bias.detach() if conv.bias is not None else None, y, ntaps=ntaps)
        if self.record:
            def bwd():
                g = self._total_grad(y)
                if g is None:
                    return
                if conv.bias is not None:
                    self._bias_grad(conv.bias, g)
                if im2col:
                    dwp = ops.wgrad(g, x, (Cout, x.C), ntaps=1)
                    cin = conv.in_channels
                    dw = dwp[:, :9 * cin].reshape(Cout, 9, cin).permute(0, 2, 1).reshape(conv.weight.shape)
                    self._give_grad(conv.weight, dw.contiguous())
                    return
                if ntaps == 1:
                    self._linear_wgrad(conv.weight, g, x)
                else:
                    self._give_grad(conv.weight, ops.wgrad(g, x, tuple(conv.weight.shape), ntaps=ntaps,
                                                           out=self._dst(conv.weight)))
                if x.needs_grad:
                    dx = self.new_act(x.N, x.H, x.W, x.C)
                    ops.conv_igemm(g, self._pack(conv.weight, L.PACK_CONV_DGRAD), None, dx, ntaps=ntaps)
                    x.add_grad(dx)

            self.tape.append(bwd)
        return y

    def conv3x3_s2(self, x: Act, conv: nn.Conv2d) -> Act:
        """Conv2d(k3, stride 2, padding 1) without a BatchNorm behind it (ResidualConv.conv_block[2],
        common_layers.py:188): forward on the LDS-DMA GEMM with UZ_TAPS_CONV_S2 (nine stride-2 taps, zero outside),
        weight gradient by the LDS-DMA kernel's stride-2 gather (bf16, map widths 16 / 32 / 64k) -- no work on
        pixels the stride skips.  The input gradient still takes the dense route: dy spread between zeros, then the
        stride-1 input-gradient convolution (as does the fp32 weight gradient)."""
        assert conv.kernel_size == (3, 3) and conv.stride == (2, 2) and conv.padding == (1, 1) and conv.in_channels == x.C
        N, H, W, Cout = x.N, x.H, x.W, conv.out_channels
        Ho, Wo = (H + 1) // 2, (W + 1) // 2
        y = self.new_act(N, Ho, Wo, Cout)
        ops.conv_igemm(x, self._pack(conv.weight, L.PACK_CONV_FWD), conv.bias.detach() if conv.bias is not None else None,
                       y, ntaps=9, taps_mode=L.TAPS_CONV_S2)
        if self.record:
            def bwd():
                g = self._total_grad(y)
                if g is None:
                    return
                if conv.bias is not None:
                    self._bias_grad(conv.bias, g)
                fast_w = self.dtype == torch.bfloat16 and (Wo in (16, 32) or (Wo >= 64 and Wo % 64 == 0)) \
                    and Ho % (64 // min(Wo, 64)) == 0
                gfull = None
                if x.needs_grad or not fast_w:
                    gfull = self.new_act(N, H, W, Cout)        # dy between zeros: the dense stride-1 gradient routes
                    ops.resample2(g, gfull, ops.RESAMPLE_ZERO_INSERT)
                if fast_w:
                    dw = ops.wgrad(g, x, tuple(conv.weight.shape), ntaps=9, taps_mode=L.TAPS_CONV_S2, out=self._dst(conv.weight))
                else:
                    dw = ops.wgrad(gfull, x, tuple(conv.weight.shape), ntaps=9, out=self._dst(conv.weight))
                self._give_grad(conv.weight, dw)
                if x.needs_grad:
                    dx = self.new_act(N, H, W, x.C)
                    ops.conv_igemm(gfull, self._pack(conv.weight, L.PACK_CONV_DGRAD), None, dx, ntaps=9)
                    x.add_grad(dx)

            self.tape.append(bwd)
        return y

    def subsample2(self, x: Act) -> Act:
        """out[n, h, w] = x[n, 2h, 2w]: what a stride-2 convolution keeps of its stride-1 result (3x3, padding 1) or
        reads of its input (1x1); strided copy, the gradient is scattered back between zeros."""
        N, H, W, C = x.N, x.H, x.W, x.C
        Ho, Wo = (H + 1) // 2, (W + 1) // 2
        out = self.new_act(N, Ho, Wo, C)
        ops.resample2(x, out, ops.RESAMPLE_SUBSAMPLE)
        if self.record and x.needs_grad:
            def bwd():
                g = self._total_grad(out)
                if g is None:
                    return
                dx = self.new_act(N, H, W, C)
                ops.resample2(g, dx, ops.RESAMPLE_ZERO_INSERT)
                x.add_grad(dx)

            self.tape.append(bwd)
        return out

    def add(self, a: Act, b: Act, out: Optional[Act] = None) -> Act:
        """a + b (the residual sums, resunet.py:56, common_layers.py:199); `out`: a concat slot to write into"""
        assert (a.N, a.H, a.W, a.C) == (b.N, b.H, b.W, b.C)
        out = out if out is not None else self.new_act(a.N, a.H, a.W, a.C)
        assert (out.N, out.H, out.W, out.C) == (a.N, a.H, a.W, a.C)
        ops.add_acts(a, b, out)
        if self.record:
            def bwd():
                g = self._total_grad(out)
                if g is None:
                    return
                if a.needs_grad:
                    a.add_grad(g)
                if b.needs_grad:
                    b.add_grad(g)

            self.tape.append(bwd)
        return out

    def add_relu(self, a: Act, b: Act) -> Act:
        """relu(a + b): `x = x + temp; x = F.relu(x)` (multiresunet.py:79-80, 127-129, 133-135)"""
        assert (a.N, a.H, a.W, a.C) == (b.N, b.H, b.W, b.C)
        out = self.new_act(a.N, a.H, a.W, a.C)
        ops.add_relu(a, b, out)
        if self.record:
            def bwd():
                g = self._total_grad(out)
                if g is None:
                    return
                d = self.new_act(a.N, a.H, a.W, a.C)
                ops.relu_bwd(out, g, d)
                if a.needs_grad:
                    a.add_grad(d)
                if b.needs_grad:
                    b.add_grad(d)

            self.tape.append(bwd)
        return out

    def input_nhwc(self, x: torch.Tensor, cpad: int = 8) -> Act:
        """Network input (N, C, H, W) fp32 as an NHWC activation whose channels are zero-padded to `cpad` (a model whose
        first layer is not one 3x3 convolution: two convolutions read the image in multiresunet.py:74-76)."""
        L.require_cuda(x)
        N, C, H, W = x.shape
        a = self.new_act(N, H, W, _round_up(C, cpad), needs_grad=False)
        a.buf.zero_()
        a.buf.view(N, H, W, a.C)[..., :C].copy_(x.permute(0, 2, 3, 1))
        return a

    def act_to_logits(self, x: Act, K: int) -> torch.Tensor:
        """the first K channels of an activation as the model output (N, K, H, W) fp32: a head that ends in a
        BatchNorm rather than in a bare 1x1 convolution (`conv_final`, multiresunet.py:196-197, :238)"""
        assert K <= x.C
        logits = x.buf.view(x.N, x.H, x.W, x.ld)[..., x.off:x.off + K].permute(0, 3, 1, 2).float().contiguous()
        if self.record:
            def bwd(g_logits: torch.Tensor):
                g = self.new_act(x.N, x.H, x.W, x.C)
                g.buf.zero_()
                g.buf.view(x.N, x.H, x.W, x.C)[..., :K].copy_(g_logits.permute(0, 2, 3, 1))
                x.add_grad(g)

            self._heads.append((bwd, 1))
        return logits

    def conv_transpose2x2(self, x: Act, m: nn.ConvTranspose2d, out: Act, sole_reader: bool = False) -> Act:
        """ConvTranspose2d(k=2, s=2) written straight into its slot of the concat buffer.
        Reference: UpSample_UNet.up (common_layers.py:104,108).  sole_reader: nothing else reads x (the output of the
        decoder block / bottleneck below): the BatchNorm-backward reduction of the layer that produced x then rides in
        the epilogue of this layer's input-gradient GEMM (see conv_bn_relu)."""
        assert m.kernel_size == (2, 2) and m.stride == (2, 2) and m.in_channels == x.C
        Co = m.out_channels
        # an odd skip size leaves one row / column that the reference fills with F.pad zeros
        # (common_layers.py:110-113: pad = [0, dx, 0, dy] for dx, dy in {0, 1}); the result sits top-left
        assert out.C == Co and out.H - 2 * x.H in (0, 1) and out.W - 2 * x.W in (0, 1)
        if (out.H, out.W) != (2 * x.H, 2 * x.W):
            v = out.buf.view(out.N, out.H, out.W, out.ld)[..., out.off:out.off + Co]
            v[:, 2 * x.H:, :, :] = 0
            v[:, :, 2 * x.W:, :] = 0
        wp = self._pack(m.weight, L.PACK_CONVT_FWD)
        # (the bias of every sub-pixel: refreshed with the weights by the batched pack launch, not by a launch of its own)
        bias4 = self._cache.get(m.bias, L.PACK_VEC_REPEAT, 4, self.dtype) if m.bias is not None else None
        ops.conv_igemm(x, wp, bias4, out, ntaps=1, store_mode=L.STORE_SHUFFLE2X2, nout=4 * Co, co=Co)

        if self.record:
            def bwd():
                g = self._sum_grads(out, 1)[0]
                if m.bias is not None:
                    cs = g.channel_sums(out=self._dst(m.bias))
                    self._give_grad(m.bias, cs if cs is not None else ops.colsum(g, out=self._dst(m.bias)))
                self._give_grad(m.weight, ops.wgrad(x, g, tuple(m.weight.shape), ntaps=4,
                                                    taps_mode=L.TAPS_GATHER2X2, out=self._dst(m.weight)))
                if x.needs_grad:
                    dx = self.new_act(x.N, x.H, x.W, x.C)
                    src = getattr(x, "bn_src", None) if (sole_reader and self.fuse_bn_reduce_convt) else None
                    part = ops.conv_igemm(g, self._pack(m.weight, L.PACK_CONVT_DGRAD), None, dx, ntaps=4,
                                          taps_mode=L.TAPS_GATHER2X2, bnred=src)
                    if src is not None and part is not None:
                        dx.bn_partials = part
                    x.add_grad(dx)

            self.tape.append(bwd)
        return out

    # ------------------------------------------------------------------ Swin-UNet V2 blocks
    def _total_grad(self, a: Act) -> Optional[Act]:
        """All gradient contributions of `a` as ONE tensor (uz_pool_grad_combine sums pairs)."""
        gs = list(a.grads)
        if not gs:
            return None
        while len(gs) > 1:
            g1, g0 = gs.pop(), gs.pop()
            tot = self.new_act(a.N, a.H, a.W, a.C)
            ops.pool_grad_combine(a, g0, g1, None, tot)
            gs.append(tot)
        return gs[0]

    def linear(self, x: Act, lin: nn.Linear, out: Optional[Act] = None, residual: Optional[Act] = None) -> Act:
        """nn.Linear on a token tensor: y[p] = W x[p] + b [+ residual[p]], on the LDS-DMA GEMM (ntaps = 1); the
        residual sum of a transformer block (x + proj(.), tx + fc2(.)) rides in the GEMM's epilogue.  In the
        backward the input-gradient GEMM adds one gradient that x has already collected (its residual branch), so
        the two are never summed by a separate pass."""
        assert lin.in_features == x.C
        y = out if out is not None else self.new_act(x.N, x.H, x.W, lin.out_features)
        ops.conv_igemm(x, self._pack(lin.weight, L.PACK_CONV_FWD), lin.bias.detach() if lin.bias is not None else None,
                       y, ntaps=1, res=residual)
        if self.record:
            def bwd():
                g = self._total_grad(y)
                if g is None:
                    return
                if residual is not None and residual.needs_grad:
                    residual.add_grad(g)
                if lin.bias is not None:
                    self._bias_grad(lin.bias, g)
                self._linear_wgrad(lin.weight, g, x)
                if x.needs_grad:
                    prev = x.grads.pop() if (x.grads and x.parts is None and x.rparts is None) else None
                    dx = self.new_act(x.N, x.H, x.W, x.C)
                    ops.conv_igemm(g, self._pack(lin.weight, L.PACK_CONV_DGRAD), None, dx, ntaps=1, res=prev)
                    x.add_grad(dx)

            self.tape.append(bwd)
        return y

    def linear_heads(self, x: Act, lins: Sequence[nn.Linear]) -> Act:
        """[lin(x) for lin in lins] side by side in one (tokens, len(lins) * out_features) activation: the per-head
        query / key / value Linear layers of UCTransNet's Attention_org (uctransnet.py:104-116, :140-158).  The heads'
        weights are stacked (one concat of their packed copies), so forward, input gradient and weight gradient are ONE
        product each instead of one per head -- on 1024 tokens every launch is latency, not work -- and the activation is
        one tensor (not a concat of parts), so the gradients several consumers leave on it accumulate in their GEMM
        epilogues."""
        Co, Ci, Hn = lins[0].out_features, x.C, len(lins)
        assert all(l.in_features == Ci and l.out_features == Co for l in lins)
        has_bias = lins[0].bias is not None
        assert all((l.bias is not None) == has_bias for l in lins)
        dt, P = self.dtype, x.P
        y = self.new_act(x.N, x.H, x.W, Hn * Co)
        wst = torch.cat([self._pack(l.weight, L.PACK_CONV_FWD) for l in lins], 0)               # (Hn Co, Ci)
        bias = torch.cat([l.bias.detach() for l in lins]) if has_bias else None
        ops.gemm_nt(dt, 1, P, Hn * Co, Ci, x.ptr(), x.ld, 0, wst.data_ptr(), Ci, 0, y.ptr(), y.ld, 0, bias=bias)
        if self.record:
            def bwd():
                g = self._total_grad(y)
                if g is None:
                    return
                if has_bias:
                    for h, lin in enumerate(lins):
                        self._bias_grad(lin.bias, g.window(h * Co, Co))
                # the heads' gradient destinations are consecutive in a flat gradient buffer (parameters of one ModuleList):
                # then the stacked product writes all of them in place
                dsts = [self._dst(l.weight) for l in lins]
                out = None
                if all(d is not None and d.is_contiguous() for d in dsts) and \
                        all(dsts[h].data_ptr() == dsts[0].data_ptr() + 4 * h * Co * Ci and
                            dsts[h].untyped_storage().data_ptr() == dsts[0].untyped_storage().data_ptr() for h in range(Hn)):
                    out = torch.as_strided(dsts[0], (Hn * Co, Ci), (Ci, 1))      # one view over the heads' slots of the flat buffer
                dW = ops.wgrad(g, x, (Hn * Co, Ci), ntaps=1, out=out)
                for h, lin in enumerate(lins):
                    self._give_grad(lin.weight, dW[h * Co:(h + 1) * Co])
                if x.needs_grad:       # dx = sum_h g_h W_h = g [W_0^T ... W_H^T]^T (+ what x has collected so far)
                    prev = x.grads.pop() if (x.grads and x.parts is None and x.rparts is None) else None
                    wdg = torch.cat([self._pack(l.weight, L.PACK_CONV_DGRAD) for l in lins], 1)   # (Ci, Hn Co)
                    dx = self.new_act(x.N, x.H, x.W, Ci)
                    ops.gemm_nt(dt, 1, P, Ci, Hn * Co, g.ptr(), g.ld, 0, wdg.data_ptr(), Hn * Co, 0, dx.ptr(), dx.ld, 0,
                                res_ptr=prev.ptr() if prev is not None else None, ldres=prev.ld if prev is not None else 0)
                    x.add_grad(dx)

            self.tape.append(bwd)
        return y

    def linear_expand2(self, x: Act, lin: nn.Linear) -> Act:
        """PatchExpand's Linear(C, 2C, bias=False) + 'b h w (p1 p2 c) -> b (h p1) (w p2) c' (p = 2):
        the GEMM's pixel-shuffle store writes the rearranged tensor directly (swin_unet_v2.py:352-360)."""
        assert lin.bias is None and lin.in_features == x.C and lin.out_features % 4 == 0
        Co = lin.out_features // 4
        y = self.new_act(x.N, 2 * x.H, 2 * x.W, Co)
        ops.conv_igemm(x, self._pack(lin.weight, L.PACK_CONV_FWD), None, y, ntaps=1,
                       store_mode=L.STORE_SHUFFLE2X2, nout=4 * Co, co=Co)
        if self.record:
            def bwd():
                g = self._total_grad(y)
                if g is None:
                    return
                d = L.WgradDesc(L.dtype_code(self.dtype), x.N, x.H, x.W, g.H, g.W, x.C, x.ld, Co, g.ld, 4, L.TAPS_GATHER2X2, 1)
                if self.dtype == torch.bfloat16 and self.defer_linear_wgrads and ops.wgrad_kernel_name(d).startswith("wgrad_bf16_"):
                    # token maps the four-tap gather kernel does not take (7 / 14 / 28 wide at 224 x 224) fell to the first
                    # generation kernel on 8 ... 72 workgroups: dy back in the Linear's own (p1 p2 c) column order, then it
                    # is one more Linear weight gradient of the deferred set, already in the parameter's layout
                    gs = self.new_act(x.N, x.H, x.W, 4 * Co, needs_grad=False)
                    ops.space_to_depth(g, gs, 2)
                    self._linear_wgrad(lin.weight, gs, x)
                else:
                    dwt = ops.wgrad(x, g, (x.C, Co, 2, 2), ntaps=4, taps_mode=L.TAPS_GATHER2X2)   # [cin][co][tap]
                    self._give_grad(lin.weight, dwt.view(x.C, Co, 4).permute(2, 1, 0).reshape(4 * Co, x.C).contiguous())
                if x.needs_grad:
                    dx = self.new_act(x.N, x.H, x.W, x.C)
                    ops.conv_igemm(g, self._pack(lin.weight, L.PACK_CONV_DGRAD), None, dx, ntaps=4,
                                   taps_mode=L.TAPS_GATHER2X2)
                    x.add_grad(dx)

            self.tape.append(bwd)
        return y

    def layer_norm(self, x: Act, ln: nn.LayerNorm, *, out: Optional[Act] = None, mode: int = L.LN_PLAIN, r: int = 1,
                   residual: Optional[Act] = None, image_scale: Optional[torch.Tensor] = None, gelu: bool = False) -> Act:
        """out = [residual +] [image_scale[b] *] LayerNorm(x); mode folds PatchMerging's gather+concat
        (LN_MERGE) or PatchExpand's rearrange (LN_EXPAND, factor r) into the addressing."""
        C = ln.normalized_shape[0]
        if mode == L.LN_PLAIN:
            shape = (x.N, x.H, x.W)
            assert C == x.C
        elif mode == L.LN_MERGE:
            shape = (x.N, x.H // 2, x.W // 2)
            assert C == 4 * x.C and x.H % 2 == 0 and x.W % 2 == 0
        else:
            shape = (x.N, x.H * r, x.W * r)
            assert C * r * r == x.C
        y = out if out is not None else self.new_act(*shape, C)
        assert (y.N, y.H, y.W, y.C) == (*shape, C)
        gamma, beta = ln.weight.detach(), ln.bias.detach()
        stats = ops.layernorm_fwd(x, gamma, beta, y, mode=mode, r=r, eps=ln.eps, res=residual, image_scale=image_scale, gelu=gelu)
        if self.record:
            def bwd():
                g = self._total_grad(y)
                if g is None:
                    return
                if residual is not None and residual.needs_grad:
                    residual.add_grad(g)
                dx = self.new_act(x.N, x.H, x.W, x.C)
                dgam, dbet = ops.layernorm_bwd(x, gamma, stats, g, dx, mode=mode, r=r, eps=ln.eps,
                                               image_scale=image_scale, dgamma=self._dst(ln.weight),
                                               dbeta=self._dst(ln.bias), gelu_beta=beta if gelu else None,
                                               defer=self._rowsum_items)
                self._after_rowsums(lambda: (self._give_grad(ln.weight, dgam), self._give_grad(ln.bias, dbet)))
                if x.needs_grad:
                    x.add_grad(dx)

            self.tape.append(bwd)
        return y

    def patch_embed(self, x: torch.Tensor, conv: nn.Conv2d) -> Act:
        """PatchEmbed.proj: Conv2d(kernel = stride = patch) as patch extraction + GEMM
        (swin_unet_v2.py:548-553); returns the (N, H/p, W/p, embed_dim) token tensor."""
        L.require_cuda(x)
        ps = conv.kernel_size[0]
        assert conv.kernel_size == conv.stride == (ps, ps) and x.dim() == 4 and x.shape[1] == conv.in_channels
        K = ps * ps * conv.in_channels
        kpad = _round_up(K, self.bk)
        p = ops.patchify(x.contiguous().float(), ps, kpad, self.dtype)
        y = self.new_act(p.N, p.H, p.W, conv.out_channels)
        ops.conv_igemm(p, self._pack(conv.weight, L.PACK_IM2COL, kpad), conv.bias.detach() if conv.bias is not None else None,
                       y, ntaps=1)
        if self.record:
            def bwd():
                g = self._total_grad(y)
                if g is None:
                    return
                if conv.bias is not None:
                    self._bias_grad(conv.bias, g)
                dwp = ops.wgrad(g, p, (conv.out_channels, kpad), ntaps=1)
                dw = dwp[:, :K].reshape(conv.out_channels, ps * ps, conv.in_channels).permute(0, 2, 1)
                self._give_grad(conv.weight, dw.reshape(conv.weight.shape).contiguous())

            self.tape.append(bwd)
        return y

    # ---- MISSFormer / MiT blocks (missformer.py) ------------------------------------
    def conv_input(self, x: torch.Tensor, conv: nn.Conv2d) -> Act:
        """A strided k x k convolution of the NCHW fp32 network input as im2col + GEMM
        (OverlapPatchEmbeddings.proj = Conv2d(3, 64, 7, 4, 3), missformer.py:242,312)."""
        L.require_cuda(x)
        k, st, pd = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        assert conv.kernel_size == (k, k) and conv.stride == (st, st) and conv.padding == (pd, pd) and x.shape[1] == conv.in_channels
        K = k * k * conv.in_channels
        kpad = _round_up(K, self.bk)
        p = ops.im2col_nchw(x.contiguous().float(), k, st, pd, kpad, self.dtype)
        y = self.new_act(p.N, p.H, p.W, conv.out_channels)
        ops.conv_igemm(p, self._pack(conv.weight, L.PACK_IM2COL, kpad), conv.bias.detach() if conv.bias is not None else None,
                       y, ntaps=1)
        if self.record:
            def bwd():
                g = self._total_grad(y)
                if g is None:
                    return
                if conv.bias is not None:
                    self._bias_grad(conv.bias, g)
                dwp = ops.wgrad(g, p, (conv.out_channels, kpad), ntaps=1)
                dw = dwp[:, :K].reshape(conv.out_channels, k * k, conv.in_channels).permute(0, 2, 1)
                self._give_grad(conv.weight, dw.reshape(conv.weight.shape).contiguous())

            self.tape.append(bwd)
        return y

    def patch_conv(self, x: Act, conv: nn.Conv2d, out: Optional[Act] = None) -> Act:
        """Conv2d(C, C', r, r) with stride r (EfficientSelfAtten.sr / Scale_reduce.sr_convs, missformer.py:17,76):
        space-to-depth, then the token GEMM over r*r*C columns; its input gradient is the transposed GEMM
        scattered back (= ConvTranspose2d with the same weight)."""
        r = conv.kernel_size[0]
        assert conv.kernel_size == conv.stride == (r, r) and conv.padding == (0, 0) and conv.in_channels == x.C
        assert x.H % r == 0 and x.W % r == 0, f"map {x.H}x{x.W} is not a multiple of the reduction ratio {r}"
        Cout, T = conv.out_channels, r * r
        xs = self.new_act(x.N, x.H // r, x.W // r, T * x.C)
        ops.space_to_depth(x, xs, r)
        y = out if out is not None else self.new_act(xs.N, xs.H, xs.W, Cout)
        ops.conv_igemm(xs, self._pack(conv.weight, L.PACK_CONV_FWD), conv.bias.detach() if conv.bias is not None else None,
                       y, ntaps=1)
        if self.record:
            def bwd():
                g = self._total_grad(y)
                if g is None:
                    return
                if conv.bias is not None:
                    self._bias_grad(conv.bias, g)
                dwp = ops.wgrad(g, xs, (Cout, T * x.C), ntaps=1)                      # [co][tap*Ci + ci]
                self._give_grad(conv.weight, dwp.view(Cout, T, x.C).permute(0, 2, 1).reshape(conv.weight.shape).contiguous())
                if x.needs_grad:
                    dxs = self.new_act(xs.N, xs.H, xs.W, xs.C)
                    ops.conv_igemm(g, self._pack(conv.weight, L.PACK_CONVT_FWD), None, dxs, ntaps=1)
                    dx = self.new_act(x.N, x.H, x.W, x.C)
                    ops.space_to_depth(dxs, dx, r, inverse=True)
                    x.add_grad(dx)

            self.tape.append(bwd)
        return y

    def gelu(self, x: Act) -> Act:
        """nn.GELU() (missformer.py:196)"""
        y = self.new_act(x.N, x.H, x.W, x.C)
        ops.gelu_fwd(x, y)
        if self.record and x.needs_grad:
            def bwd():
                g = self._total_grad(y)
                if g is None:
                    return
                dx = self.new_act(x.N, x.H, x.W, x.C)
                ops.gelu_bwd(x, g, dx)
                x.add_grad(dx)

            self.tape.append(bwd)
        return y

    def dwconv_skip(self, x: Act, conv: nn.Conv2d, skip: bool = True) -> Act:
        """DWConv(x) + x: the depthwise 3x3 of MixFFN_skip with its skip (missformer.py:168-177, :204-205);
        skip=False: the bare DWConv of MixFFN (:186-189)"""
        C = x.C
        assert conv.groups == C == conv.in_channels == conv.out_channels and conv.kernel_size == (3, 3) and conv.padding == (1, 1)
        wt = conv.weight.detach().reshape(C, 9).t().contiguous()      # [9][C]
        y = self.new_act(x.N, x.H, x.W, C)
        ops.dwconv3x3(x, wt, conv.bias.detach() if conv.bias is not None else None, y, skip=skip)
        if self.record:
            def bwd():
                g = self._total_grad(y)
                if g is None:
                    return
                dwb = ops.dwconv3x3_wgrad(x, g, defer=self._rowsum_items)

                def give():
                    self._give_grad(conv.weight, dwb[:9].t().reshape(conv.weight.shape).contiguous())
                    if conv.bias is not None:
                        self._give_grad(conv.bias, dwb[9].contiguous())

                self._after_rowsums(give)
                if x.needs_grad:
                    dx = self.new_act(x.N, x.H, x.W, C)
                    ops.dwconv3x3(g, wt, None, dx, skip=skip, flip=True)
                    x.add_grad(dx)

            self.tape.append(bwd)
        return y

    def sr_attention(self, q: Act, kv: Act, B: int, heads: int, kps: int, scale: float,
                     segments: Optional[Sequence[Tuple[int, int]]] = None) -> Act:
        """softmax(q k^T * scale) v per (image, head), head_dim 64 (missformer.py:30-36, :122-125).  `segments`:
        (first row, queries per image) of the row blocks of q that attend to the same keys (the bridge's four
        scales, each stored [B][n_s]); default: q is one [B][N] block."""
        if segments is None:
            segments = [(0, q.P // B)]
        out = self.new_act(q.N, q.H, q.W, q.C)
        lses = []
        for r0, n in segments:
            lses.append(ops.sra_fwd(q.rows(r0, B, 1, n), kv, out.rows(r0, B, 1, n), B, heads, kps, scale))
        if self.record:
            def bwd():
                g = self._total_grad(out)
                if g is None:
                    return
                dq = self.new_act(q.N, q.H, q.W, q.C)
                for (r0, n), lse in zip(segments, lses):
                    dkv = self.new_act(kv.N, kv.H, kv.W, kv.C)
                    ops.sra_bwd(q.rows(r0, B, 1, n), kv, out.rows(r0, B, 1, n), lse, g.rows(r0, B, 1, n),
                                dq.rows(r0, B, 1, n), dkv, B, heads, kps, scale)
                    kv.add_grad(dkv)
                q.add_grad(dq)

            self.tape.append(bwd)
        return out

    def new_rows(self, shapes: Sequence[Tuple[int, int, int]], C: int) -> Tuple[Act, List[Act]]:
        """One token buffer holding a concat along the token axis (torch.cat(..., -2), missformer.py:98,681,699),
        block s = an (N, H, W, C) tensor written in place by its producer."""
        total = sum(n * h * w for n, h, w in shapes)
        full = self.new_act(1, 1, total, C)
        parts, r0 = [], 0
        for n, h, w in shapes:
            parts.append((full.rows(r0, n, h, w), r0))
            r0 += n * h * w
        full.rparts = parts
        return full, [p for p, _ in parts]

    def row_views(self, flat: Act, shapes: Sequence[Tuple[int, int, int]]) -> List[Act]:
        """The row blocks of `flat` as tensors of their own (the slices tx[:, a:b, :] of missformer.py:86,689-692);
        their gradients are gathered into one buffer for `flat`."""
        views, r0s, r0 = [], [], 0
        for n, h, w in shapes:
            views.append(flat.rows(r0, n, h, w))
            r0s.append(r0)
            r0 += n * h * w
        assert r0 == flat.P
        if self.record and flat.needs_grad:
            def bwd():
                gs = [self._total_grad(v) for v in views]
                if all(g is None for g in gs):
                    return
                gflat = self.new_act(flat.N, flat.H, flat.W, flat.C)
                for v, g, a in zip(views, gs, r0s):
                    dst = gflat.rows(a, v.N, v.H, v.W)
                    if g is None:
                        dst.buf.zero_()
                    else:
                        ops.resample2(g, dst, ops.RESAMPLE_COPY)
                flat.add_grad(gflat)

            self.tape.append(bwd)
        return views

    def add_const(self, x: Act, const_map: torch.Tensor) -> Act:
        """x + c for a parameter-free (H*W, C) map c broadcast over the batch: the sinusoidal position encodings
        `x + self.pe(x)` of unet_transformer.py:133-134, :181-185.  The gradient passes through unchanged."""
        assert const_map.shape == (x.H * x.W, x.C), (tuple(const_map.shape), (x.H * x.W, x.C))
        out = self.new_act(x.N, x.H, x.W, x.C, x.needs_grad)
        ops.add_map(x, const_map.to(self.device, torch.float32).contiguous(), out)
        if self.record and x.needs_grad:
            def bwd():
                g = self._total_grad(out)
                if g is not None:
                    x.add_grad(g)

            self.tape.append(bwd)
        return out

    def add_param_map(self, x: Act, p: nn.Parameter, out: Optional[Act] = None) -> Act:
        """x + p for a (1, H*W, C) parameter broadcast over the batch: `x = x + self.absolute_pos_embed`
        (swin_unet_v2.py:714-715).  d(p) = sum over the batch of the incoming gradient (N terms per element)."""
        assert tuple(p.shape) == (1, x.H * x.W, x.C), (tuple(p.shape), (1, x.H * x.W, x.C))
        out = out if out is not None else self.new_act(x.N, x.H, x.W, x.C)
        ops.add_map(x, p.detach()[0], out)
        if self.record:
            def bwd():
                g = self._total_grad(out)
                if g is None:
                    return
                gv = g.buf.view(x.N, x.H * x.W, g.ld)[..., g.off:g.off + x.C].float()
                self._give_grad(p, gv.sum(0, keepdim=True))
                if x.needs_grad:
                    x.add_grad(g)

            self.tape.append(bwd)
        return out

    def dropout(self, x: Act, p: float, out: Optional[Act] = None) -> Act:
        """nn.Dropout(p) in training mode (identity otherwise): Bernoulli(1 - p) mask from torch's generator, scaled by
        1 / (1 - p); the same mask multiplies the gradient (swin_unet_v2.py:158, :716)."""
        if p <= 0.0 or not self.training:
            if out is not None and out is not x:
                return self.copy_into(x, out)
            return x
        out = out if out is not None else self.new_act(x.N, x.H, x.W, x.C)
        # the draw is torch's (one launch); mask, fp32 scale 1 / (1 - p) (in bf16 1.1111 rounds to 1.1094: every kept
        # activation and its gradient would come out 0.16 % low against nn.Dropout) and store are one kernel
        u = torch.rand((x.P, x.C), device=self.device)
        ops.dropout(x, u, p, out)
        if self.record and x.needs_grad:
            def bwd():
                g = self._total_grad(out)
                if g is None:
                    return
                dx = self.new_act(x.N, x.H, x.W, x.C)
                ops.dropout(g, u, p, dx)
                x.add_grad(dx)

            self.tape.append(bwd)
        return out

    def max_pool2x2(self, x: Act) -> Act:
        """nn.MaxPool2d(2) of a tensor that is not a BatchNorm/ReLU output (`Sconv_process[0]`,
        unet_transformer.py:151): the fused kernel with unit scale, zero shift and no ReLU, writing `x` back in place;
        the backward routes each window's gradient to its first maximum (uz_pool_grad_combine)."""
        assert x.H >= 2 and x.W >= 2
        one = torch.ones(x.C, dtype=torch.float32, device=self.device)
        zero = torch.zeros(x.C, dtype=torch.float32, device=self.device)
        pooled = self.new_act(x.N, x.H // 2, x.W // 2, x.C, x.needs_grad)
        ops.bn_relu_apply(x, one, zero, x, pooled, relu=False)
        if self.record and x.needs_grad:
            def bwd():
                g = self._total_grad(pooled)
                if g is None:
                    return
                dx = self.new_act(x.N, x.H, x.W, x.C)
                ops.pool_grad_combine(x, None, None, g, dx)
                x.add_grad(dx)

            self.tape.append(bwd)
        return pooled

    # ------------------------------------------------------------------ library-GEMM glue (bottleneck attention)
    # ------------------------------------------------------------------ token grids the attention kernels take
    # The dense attention kernels address score matrices with 16-byte rows: H * W tokens per image must be a multiple of
    # 8.  Other maps (a 9 x 9 bottleneck of a 72 x 72 input) run on a grid WIDENED to the next multiple of 8 columns:
    # zero tokens there, masked out of every softmax (they must not count as keys / queries), cropped from the result.
    # (Rounds 2-4 sent such shapes through torch.bmm / autograd -- a second backend; this replaces it.)
    MASKED_SCORE = -30000.0   # a score no real one comes near; finite, so that a slice of nothing but masked entries
                              # still has a maximum to subtract (exp(-inf - (-inf)) is a NaN)

    @staticmethod
    def padded_width(H: int, W: int) -> int:
        return W if (H * W) % 8 == 0 else (W + 7) // 8 * 8

    def pad_w(self, x: Act, Wp: int) -> Act:
        """x on a token grid of Wp >= x.W columns, the added columns zero"""
        if Wp == x.W:
            return x
        xp = self.new_act(x.N, x.H, Wp, x.C, needs_grad=x.needs_grad)
        xp.buf.zero_()
        xp.buf.view(x.N, x.H, Wp, x.C)[:, :, :x.W].copy_(x.buf.view(x.N, x.H, x.W, x.ld)[..., x.off:x.off + x.C])
        if self.record and x.needs_grad:
            def bwd():
                g = self._total_grad(xp)
                if g is None:
                    return
                dx = self.new_act(x.N, x.H, x.W, x.C)
                dx.buf.view(x.N, x.H, x.W, x.C).copy_(g.buf.view(x.N, x.H, Wp, g.ld)[:, :, :x.W, g.off:g.off + x.C])
                x.add_grad(dx)
            self.tape.append(bwd)
        return xp

    def crop_w(self, xp: Act, W: int, out: Optional[Act] = None) -> Act:
        """the first W columns of xp's token grid (into `out` if given)"""
        if W == xp.W and out is None:
            return xp
        y = out if out is not None else self.new_act(xp.N, xp.H, W, xp.C, needs_grad=xp.needs_grad)
        y.buf.view(y.N, y.H, W, y.ld)[..., y.off:y.off + y.C].copy_(
            xp.buf.view(xp.N, xp.H, xp.W, xp.ld)[:, :, :W, xp.off:xp.off + xp.C])
        if self.record and xp.needs_grad:
            def bwd():
                g = self._total_grad(y)
                if g is None:
                    return
                dp = self.new_act(xp.N, xp.H, xp.W, xp.C)
                dp.buf.zero_()
                dp.buf.view(xp.N, xp.H, xp.W, xp.C)[:, :, :W].copy_(g.buf.view(y.N, y.H, W, g.ld)[..., g.off:g.off + y.C])
                xp.add_grad(dp)
            self.tape.append(bwd)
        return y

    def adaptive_avg_pool(self, x: Act, Ho: int, Wo: int) -> Act:
        """F.adaptive_avg_pool2d(x, (Ho, Wo)) (unet_transformer.py:196-198); the identity when the map already has that
        size"""
        if (x.H, x.W) == (Ho, Wo):
            return x
        y = self.new_act(x.N, Ho, Wo, x.C, needs_grad=x.needs_grad)
        ops.adaptive_avgpool_fwd(x, y)
        if self.record and x.needs_grad:
            def bwd():
                g = self._total_grad(y)
                if g is None:
                    return
                dx = self.new_act(x.N, x.H, x.W, x.C)
                ops.adaptive_avgpool_bwd(g, dx)
                x.add_grad(dx)

            self.tape.append(bwd)
        return y

    def _tn_images(self, Lt: Act, Rt: Act) -> torch.Tensor:
        """out_b = L_b^T R_b per image, fp32 (N * Lt.C, Rt.C): the products that contract over the ROWS of both operands
        (dV = A^T dO, dK = dS^T Q, the channel Gram matrix x^T x) on uz_wgrad's one-tap kernel"""
        return ops.wgrad_batched(Lt, Rt).view(Lt.N * Lt.C, Rt.C)

    @staticmethod
    def _matrices_as_act(m: torch.Tensor) -> Act:
        """a contiguous (B, rows, cols) batch of matrices as the activation (B, 1, rows, cols)"""
        B, R, C = m.shape
        return Act(m.view(B * R, C), 0, C, B, 1, R)

    @staticmethod
    def _transposed(a: Act) -> torch.Tensor:
        """(B, C, tokens) copy of a token-major activation -- glue for operands of a few MB (q / k / v of a bottleneck)"""
        return a.buf.view(a.N, a.H * a.W, a.ld)[..., a.off:a.off + a.C].transpose(1, 2).contiguous()

    def row_attention(self, q: Act, k: Act, v: Act, out: Act, valid_w: Optional[int] = None) -> Act:
        """out_i = sum_j softmax_j(q_i . k_j) v_j per image: PAM_Module.forward between its 1x1 convolutions and the
        `gamma * out + x` (transatt_unet.py:41-49).  energy and its gradient are batched NT products, softmax over the
        key axis in place (kept for the backward), dv = att^T g and dk = dE^T q on the one-tap weight-gradient kernel."""
        B, Nt, C, dq = q.N, q.H * q.W, v.C, q.C
        assert (k.N, k.H * k.W, k.C) == (B, Nt, dq) and (v.N, v.H * v.W) == (B, Nt) and (out.N, out.H * out.W, out.C) == (B, Nt, C)
        dt, dev = self.dtype, self.device
        vt = self._transposed(v)
        E = torch.empty((B, Nt, Nt), dtype=dt, device=dev)
        ops.gemm_nt(dt, B, Nt, Nt, dq, q.ptr(), q.ld, Nt * q.ld, k.ptr(), k.ld, Nt * k.ld, E.data_ptr(), Nt, Nt * Nt)
        if valid_w is not None and valid_w < k.W:    # widened grid (pad_w): the added KEYS leave every query's softmax
            E.view(B, Nt, k.H, k.W)[..., valid_w:] = self.MASKED_SCORE
        ops.softmax_fwd(E, 1, 1.0)
        ops.gemm_nt(dt, B, Nt, C, Nt, E.data_ptr(), Nt, Nt * Nt, vt.data_ptr(), Nt, C * Nt, out.ptr(), out.ld, Nt * out.ld)
        del vt
        if not self.record:
            return out

        def bwd():
            g = self._total_grad(out)
            if g is None:
                return
            dE = torch.empty((B, Nt, Nt), dtype=dt, device=dev)
            ops.gemm_nt(dt, B, Nt, Nt, C, g.ptr(), g.ld, Nt * g.ld, v.ptr(), v.ld, Nt * v.ld, dE.data_ptr(), Nt, Nt * Nt)
            if v.needs_grad:
                dv = self.new_act(B, v.H, v.W, C)
                ops.cast_rows(self._tn_images(self._matrices_as_act(E), g), dv)
                v.add_grad(dv)
            ops.softmax_bwd(E, dE, 1, 1.0)
            if q.needs_grad:
                kt = self._transposed(k)
                dqa = self.new_act(B, q.H, q.W, dq)
                ops.gemm_nt(dt, B, Nt, dq, Nt, dE.data_ptr(), Nt, Nt * Nt, kt.data_ptr(), Nt, dq * Nt, dqa.ptr(), dq, Nt * dq)
                q.add_grad(dqa)
            if k.needs_grad:
                dka = self.new_act(B, k.H, k.W, dq)
                ops.cast_rows(self._tn_images(self._matrices_as_act(dE), q), dka)
                k.add_grad(dka)

        self.tape.append(bwd)
        return out

    def channel_attention(self, x: Act, temperature: float, p_drop: float, out: Act) -> Act:
        """out = dropout(softmax((x / T) x^T, dim=-1)) x on the (d, tokens) view of every image:
        ScaledDotProductAttention.forward as TransAttUNet calls it (transatt_unet.py:91-107, q = k = v = the bottleneck
        map).  The (d, d) Gram matrix is a TN product (fp32 out of the one-tap weight-gradient kernel), softmax runs on it
        in fp32, the two products with x are batched NT products; the only transposes are of (d, d) matrices."""
        B, Nt, d = x.N, x.H * x.W, x.C
        assert (out.N, out.H * out.W, out.C) == (B, Nt, d)
        dt, dev = self.dtype, self.device
        P = self._tn_images(x, x).view(B, d, d)
        ops.softmax_fwd(P, 1, 1.0 / temperature)
        drop = p_drop > 0.0 and self.training
        if drop:       # nn.Dropout in training mode: Bernoulli(1 - p) mask from torch's generator, scaled by 1 / (1 - p)
            keep = (torch.rand((B, d, d), device=dev) >= p_drop).to(torch.float32).mul_(1.0 / (1.0 - p_drop))
            Pd = (P * keep).to(dt)
        else:
            keep, Pd = None, P.to(dt)
        ops.gemm_nt(dt, B, Nt, d, d, x.ptr(), x.ld, Nt * x.ld, Pd.data_ptr(), d, d * d, out.ptr(), out.ld, Nt * out.ld)
        if not (self.record and x.needs_grad):
            return out

        def bwd():
            g = self._total_grad(out)
            if g is None:
                return
            dP = self._tn_images(g, x).view(B, d, d)                  # d(loss)/d(Pd)[c1][c2] = sum_n g[n][c1] x[n][c2]
            if keep is not None:
                dP.mul_(keep)
            ops.softmax_bwd(P, dP, 1, 1.0 / temperature)              # now d(loss)/d(x^T x)
            sym = (dP + dP.transpose(1, 2)).to(dt)                     # x enters the Gram matrix on both sides
            PdT = Pd.transpose(1, 2).contiguous()
            dx = self.new_act(B, x.H, x.W, d)
            ops.gemm_nt(dt, B, Nt, d, d, g.ptr(), g.ld, Nt * g.ld, PdT.data_ptr(), d, d * d, dx.ptr(), d, Nt * d)
            ops.gemm_nt(dt, B, Nt, d, d, x.ptr(), x.ld, Nt * x.ld, sym.data_ptr(), d, d * d, dx.ptr(), d, Nt * d,
                        res_ptr=dx.ptr(), ldres=d, resb=Nt * d)
            x.add_grad(dx)

        self.tape.append(bwd)
        return out

    def add_row_col_embed(self, x: Act, row_w: nn.Parameter, col_w: nn.Parameter) -> Act:
        """x + cat([col_embed(j) for every row, row_embed(i) for every column], channel) -- PositionEmbeddingLearned as
        TransAttUNet adds it to the bottleneck (transatt_unet.py:66-82, :144-145).  The gradient of an embedding row is
        the sum of the incoming gradient over the batch and over the other map axis (256 terms at 16 x 16, B = 16)."""
        F_ = col_w.shape[1]
        assert x.C == 2 * F_ and row_w.shape[1] == F_ and x.H <= row_w.shape[0] and x.W <= col_w.shape[0]
        H, W = x.H, x.W
        pos = torch.cat([col_w.detach()[:W].unsqueeze(0).expand(H, W, F_), row_w.detach()[:H].unsqueeze(1).expand(H, W, F_)],
                        dim=-1).reshape(H * W, 2 * F_)
        out = self.new_act(x.N, H, W, x.C)
        ops.add_map(x, pos.contiguous(), out)
        if self.record:
            def bwd():
                g = self._total_grad(out)
                if g is None:
                    return
                gv = g.buf.view(x.N, H, W, g.ld)[..., g.off:g.off + x.C].float()
                d_col = torch.zeros_like(col_w)
                d_row = torch.zeros_like(row_w)
                d_col[:W] = gv[..., :F_].sum((0, 1))
                d_row[:H] = gv[..., F_:].sum((0, 2))
                self._give_grad(col_w, d_col)
                self._give_grad(row_w, d_row)
                if x.needs_grad:
                    x.add_grad(g)

            self.tape.append(bwd)
        return out

    def channel_cross_attention(self, Q: Act, K: Act, V: Act, heads: int, eps: float = 1e-5,
                                probs_out: Optional[list] = None) -> Act:
        """The channel-wise cross attention of one scale of UCTransNet between its Linear layers
        (Attention_org.forward, uctransnet.py:160-199): Q (tokens, heads * C), K and V (tokens, heads * KV) hold the heads
        side by side; per (image, head) scores = Q_h^T K_h / sqrt(KV) (a product over the TOKENS: the one-tap
        weight-gradient kernel), InstanceNorm2d over the (C, KV) plane, softmax over KV, context = P V_h^T, mean over the
        heads -- the last two as ONE product over K = heads * KV with P / heads laid out (C, heads * KV).  Returns the
        (tokens, C) context; `probs_out` (a list) receives `attention_probs.mean(1)` (B, C, KV), detached fp32: the
        visualisation output of `vis=True` (uctransnet.py:180-185)."""
        B, n, H = Q.N, Q.H * Q.W, heads
        C, KV = Q.C // H, K.C // H
        assert Q.C == H * C and K.C == H * KV and V.C == H * KV and (K.N, K.H * K.W) == (B, n) and (V.N, V.H * V.W) == (B, n)
        dt, dev = self.dtype, self.device
        es = 2 if dt == torch.bfloat16 else 4
        scale = 1.0 / math.sqrt(KV)
        scores = ops.wgrad_heads(Q, K, H)                               # (B, H, C, KV): one launch for all (image, head) pairs
        pcat, pcat_t = ops.chanattn_probs_fwd(scores, scale, eps, dt)
        if probs_out is not None:
            probs_out.append(pcat.view(B, C, H, KV).float().sum(2))      # pcat holds P / heads
        ctx = self.new_act(B, Q.H, Q.W, C)
        ops.gemm_nt(dt, B, n, C, H * KV, V.ptr(), V.ld, n * V.ld, pcat.data_ptr(), H * KV, C * H * KV, ctx.ptr(), ctx.ld, n * ctx.ld)
        if not self.record:
            return ctx

        def bwd():
            g = self._total_grad(ctx)
            if g is None:
                return
            dpc = ops.wgrad_batched(g, V)                                   # (B, C, H KV): d(loss)/d(P / heads)
            if V.needs_grad:
                prevV = V.grads.pop() if V.grads else None
                dV = self.new_act(B, V.H, V.W, H * KV)
                ops.gemm_nt(dt, B, n, H * KV, C, g.ptr(), g.ld, n * g.ld, pcat_t.data_ptr(), C, H * KV * C, dV.ptr(), dV.ld, n * dV.ld,
                            res_ptr=prevV.ptr() if prevV is not None else None, ldres=prevV.ld if prevV is not None else 0,
                            resb=n * prevV.ld if prevV is not None else 0)
                V.add_grad(dV)
            ds, ds_t = ops.chanattn_probs_bwd(scores, dpc, scale, eps, dt)
            dQ = self.new_act(B, Q.H, Q.W, H * C)
            dK = self.new_act(B, K.H, K.W, H * KV)
            # K (and V) serve every scale: a gradient another scale has already left is added in the product's epilogue
            # instead of by a separate pass
            prevK = K.grads.pop() if (K.needs_grad and K.grads) else None
            # dQ_h = K_h dS_h^T and dK_h = Q_h dS_h for every (image, head): one launch each (second batch level = head)
            ops.gemm_nt(dt, B, n, C, KV, K.ptr(), K.ld, n * K.ld, ds.data_ptr(), KV, H * C * KV, dQ.ptr(), dQ.ld, n * dQ.ld,
                        batch2=H, xb2=KV, wb2=C * KV, yb2=C)
            ops.gemm_nt(dt, B, n, KV, C, Q.ptr(), Q.ld, n * Q.ld, ds_t.data_ptr(), C, H * KV * C, dK.ptr(), dK.ld, n * dK.ld,
                        res_ptr=prevK.ptr() if prevK is not None else None, ldres=prevK.ld if prevK is not None else 0,
                        resb=n * prevK.ld if prevK is not None else 0,
                        batch2=H, xb2=C, wb2=KV * C, yb2=KV, resb2=KV)
            if Q.needs_grad:
                Q.add_grad(dQ)
            if K.needs_grad:
                K.add_grad(dK)

        self.tape.append(bwd)
        return ctx

    def token_attention(self, xq: Act, xv: Act, wq: nn.Parameter, wk: nn.Parameter, wv: nn.Parameter, out: Act,
                        valid_w: Optional[int] = None) -> Act:
        """out_b = softmax_over_queries((X_b wq)(X_b wk)^T / sqrt(c)) (XV_b wv) on the tokens of xq / xv (NHWC rows = the
        reference's `flatten(2).permute(0, 2, 1)`): MultiHeadSelfAttention.forward (xq is xv) and the attention core of
        MultiHeadCrossAttention.forward (unet_transformer.py:126-137, :200-213).  `nn.Softmax(dim=1)` on the (b, queries,
        keys) scores normalises over the QUERY axis, as the reference has it.

        Batched NT products (uz_gemm_nt; the transposed projections V^T = wv^T XV^T, K^T = wk^T X^T come out of the same
        kernel with the operand roles swapped, so no transposing pass exists), column softmax in place on the score
        matrix (kept for the backward), TN products (dV = A^T dO, dK = dS^T Q, the weight gradients) on uz_wgrad's
        one-tap kernel."""
        B, Nt, c = xq.N, xq.H * xq.W, xq.C
        assert (xv.N, xv.H * xv.W, xv.C) == (B, Nt, c) and (out.N, out.H * out.W, out.C) == (B, Nt, c)
        assert tuple(wq.shape) == (c, c) and tuple(wk.shape) == (c, c) and tuple(wv.shape) == (c, c)
        dt, dev = self.dtype, self.device
        scale = 1.0 / math.sqrt(c)
        es = 2 if dt == torch.bfloat16 else 4

        def proj(x: Act, w: nn.Parameter) -> torch.Tensor:            # (B Nt, c) = X w
            y = torch.empty((B * Nt, c), dtype=dt, device=dev)
            ops.gemm_nt(dt, 1, B * Nt, c, c, x.ptr(), x.ld, 0, self._pack(w, L.PACK_CONV_DGRAD).data_ptr(), c, 0,
                        y.data_ptr(), c, 0)
            return y

        def proj_t(x: Act, w: nn.Parameter) -> torch.Tensor:          # (B, c, Nt) = (X_b w)^T = w^T X_b^T
            y = torch.empty((B, c, Nt), dtype=dt, device=dev)
            ops.gemm_nt(dt, B, c, Nt, c, self._pack(w, L.PACK_CONV_DGRAD).data_ptr(), c, 0, x.ptr(), x.ld, Nt * x.ld,
                        y.data_ptr(), Nt, c * Nt)
            return y

        Q, K = proj(xq, wq), proj(xq, wk)
        Vt = proj_t(xv, wv)
        A = torch.empty((B, Nt, Nt), dtype=dt, device=dev)
        ops.gemm_nt(dt, B, Nt, Nt, c, Q.data_ptr(), c, Nt * c, K.data_ptr(), c, Nt * c, A.data_ptr(), Nt, Nt * Nt)
        if valid_w is not None and valid_w < xq.W:   # widened grid: the added QUERIES leave every key's softmax (over queries)
            A.view(B, xq.H, xq.W, Nt)[:, :, valid_w:, :] = self.MASKED_SCORE
        ops.softmax_fwd(A, 0, scale)
        ops.gemm_nt(dt, B, Nt, c, Nt, A.data_ptr(), Nt, Nt * Nt, Vt.data_ptr(), Nt, c * Nt, out.ptr(), out.ld, Nt * out.ld)
        del Vt
        if not self.record:
            return out

        def tn_per_image(Lm: torch.Tensor, R: Act) -> torch.Tensor:
            return self._tn_images(self._matrices_as_act(Lm), R)

        def as_act(t: torch.Tensor) -> Act:
            return Act(t, 0, c, B, xq.H, xq.W)

        def bwd():
            g = self._total_grad(out)
            if g is None:
                return
            V = proj(xv, wv)
            dA = torch.empty((B, Nt, Nt), dtype=dt, device=dev)
            ops.gemm_nt(dt, B, Nt, Nt, c, g.ptr(), g.ld, Nt * g.ld, V.data_ptr(), c, Nt * c, dA.data_ptr(), Nt, Nt * Nt)
            dV32 = tn_per_image(A, g)
            # sum_q A[q][k] dA[q][k] = dV[k] . V[k]: the softmax gradient's column sums without a pass over A and dA
            dot = ops.rowdot_f32(dV32, V).view(B, Nt)
            ops.softmax_bwd(A, dA, 0, scale, dot)          # dA now holds dS
            dV = self.new_act(B, xq.H, xq.W, c)
            ops.cast_rows(dV32, dV)
            del dV32, V
            Kt = proj_t(xq, wk)
            dQ = self.new_act(B, xq.H, xq.W, c)
            ops.gemm_nt(dt, B, Nt, c, Nt, dA.data_ptr(), Nt, Nt * Nt, Kt.data_ptr(), Nt, c * Nt, dQ.ptr(), c, Nt * c)
            del Kt
            dK32 = tn_per_image(dA, as_act(Q))
            dK = self.new_act(B, xq.H, xq.W, c)
            ops.cast_rows(dK32, dK)
            del dK32, dA
            self._give_grad(wq, ops.wgrad(xq, dQ, (c, c), ntaps=1, out=self._dst(wq)))
            self._give_grad(wk, ops.wgrad(xq, dK, (c, c), ntaps=1, out=self._dst(wk)))
            self._give_grad(wv, ops.wgrad(xv, dV, (c, c), ntaps=1, out=self._dst(wv)))
            if xq.needs_grad:
                dx = self.new_act(B, xq.H, xq.W, c)
                ops.gemm_nt(dt, 1, B * Nt, c, c, dQ.ptr(), c, 0, self._pack(wq, L.PACK_CONV_FWD).data_ptr(), c, 0, dx.ptr(), c, 0)
                ops.gemm_nt(dt, 1, B * Nt, c, c, dK.ptr(), c, 0, self._pack(wk, L.PACK_CONV_FWD).data_ptr(), c, 0, dx.ptr(), c, 0,
                            res_ptr=dx.ptr(), ldres=c)
                xq.add_grad(dx)
            if xv.needs_grad:
                dx = self.new_act(B, xq.H, xq.W, c)
                ops.gemm_nt(dt, 1, B * Nt, c, c, dV.ptr(), c, 0, self._pack(wv, L.PACK_CONV_FWD).data_ptr(), c, 0, dx.ptr(), c, 0)
                xv.add_grad(dx)

        self.tape.append(bwd)
        return out

    def upsample_nearest(self, x: Act, factor: int, out: Act, add: Optional[Act] = None) -> Act:
        """out = nearest-neighbour upsampling of x by an integer factor (+ add): nn.Upsample(scale_factor=f) of
        uctransnet.py:75, :435 and the residual `x1 + en1` of :358-361.  Streaming glue on the native layout (a
        broadcast copy); the backward sums each f x f block in two short reductions (f elements each)."""
        f = factor
        assert (out.N, out.H, out.W, out.C) == (x.N, x.H * f, x.W * f, x.C)
        src = x.buf.view(x.N, x.H, 1, x.W, 1, x.ld)[..., x.off:x.off + x.C]
        dst = out.buf.view(x.N, x.H, f, x.W, f, out.ld)[..., out.off:out.off + x.C]
        if add is None:
            dst.copy_(src.expand(x.N, x.H, f, x.W, f, x.C))
        else:
            assert (add.N, add.H, add.W, add.C) == (out.N, out.H, out.W, out.C)
            av = add.buf.view(x.N, x.H, f, x.W, f, add.ld)[..., add.off:add.off + x.C]
            dst.copy_(av.float() + src.float())
        if self.record:
            def bwd():
                g = self._total_grad(out)
                if g is None:
                    return
                if add is not None and add.needs_grad:
                    add.add_grad(g)
                if x.needs_grad:
                    gv = g.buf.view(x.N, x.H, f, x.W, f, g.ld)[..., g.off:g.off + x.C].float()
                    dx = self.new_act(x.N, x.H, x.W, x.C)
                    dx.buf.view(x.N, x.H, x.W, x.C).copy_(gv.sum(4).sum(2))
                    x.add_grad(dx)

            self.tape.append(bwd)
        return out

    def cca_gate(self, g_low: Act, x: Act, lin_x: nn.Linear, lin_g: nn.Linear, out: Act) -> Act:
        """CCA.forward (uctransnet.py:417-427): out = relu(x * sigmoid((mlp_x(avgpool(x)) + mlp_g(avgpool(g))) / 2)).
        `g_low` is the decoder tensor BEFORE its nearest x2 upsampling (the global average is the same).  The global
        averages and the gradient of the per-(image, channel) scale are reduced by uz_colsum_batched; the two Linear
        layers on (N, C) vectors are torch ops; the gating and its two gradient passes are uz_chanscale_relu."""
        N, C = x.N, x.C
        assert lin_x.in_features == C and lin_x.out_features == C and lin_g.out_features == C and lin_g.in_features == g_low.C

        def image_means(a: Act) -> torch.Tensor:
            sums = torch.empty((a.N, a.C), dtype=torch.float32, device=self.device)
            hw = a.H * a.W
            ops.colsum_batched([(a.rows(i * hw, 1, a.H, a.W), sums[i]) for i in range(a.N)])
            return sums / float(hw)

        ax, ag = image_means(x), image_means(g_low)
        if self.record:
            ax.requires_grad_(True)
            ag.requires_grad_(True)
        with torch.set_grad_enabled(self.record):
            att = (torch.nn.functional.linear(ax, lin_x.weight, lin_x.bias) + torch.nn.functional.linear(ag, lin_g.weight, lin_g.bias)) / 2.0
            scale = torch.sigmoid(att)                                         # (N, C)
        sc = scale.detach().contiguous()
        ops.chanscale_relu(2, None, x, sc, None, out)
        if self.record:
            def bwd():
                g = self._total_grad(out)
                if g is None:
                    return
                hw = x.H * x.W
                prod = self.new_act(N, x.H, x.W, C)
                ops.chanscale_relu(0, g, x, None, None, prod)                 # g * [x > 0] * x (sigmoid > 0: the ReLU mask is x > 0)
                dscale = torch.empty((N, C), dtype=torch.float32, device=self.device)
                ops.colsum_batched([(prod.rows(i * hw, 1, x.H, x.W), dscale[i]) for i in range(N)])
                params = [lin_x.weight, lin_x.bias, lin_g.weight, lin_g.bias]
                gr = torch.autograd.grad(scale, [ax, ag] + params, dscale)
                for p_, gp in zip(params, gr[2:]):
                    self._give_grad(p_, gp)
                if x.needs_grad:
                    dx = self.new_act(N, x.H, x.W, C)
                    ops.chanscale_relu(1, g, x, sc, (gr[0] / float(hw)).contiguous(), dx)
                    x.add_grad(dx)
                if g_low.needs_grad:
                    dg = self.new_act(g_low.N, g_low.H, g_low.W, g_low.C)
                    ghw = g_low.H * g_low.W
                    dg.buf.view(N, ghw, g_low.C).copy_((gr[1] / float(ghw))[:, None, :].expand(N, ghw, g_low.C))
                    g_low.add_grad(dg)

            self.tape.append(bwd)
        return out

    def scale_residual(self, a: Act, gamma: nn.Parameter, x: Act) -> Act:
        """gamma * a + x for a one-element parameter (PAM_Module's `self.gamma * out + x`, transatt_unet.py:51).
        d(gamma) = <g, a> is reduced by this library's own row-sum kernel, not by a torch reduction."""
        assert gamma.numel() == 1 and (a.N, a.H, a.W, a.C) == (x.N, x.H, x.W, x.C)
        out = self.new_act(x.N, x.H, x.W, x.C)
        av = a.buf[:, a.off:a.off + a.C]
        xv = x.buf[:, x.off:x.off + x.C]
        out.buf.copy_((xv.float() + av.float() * gamma.detach().float()).to(out.dtype))
        if self.record:
            def bwd():
                g = self._total_grad(out)
                if g is None:
                    return
                gv = g.buf[:, g.off:g.off + g.C].float()
                per_pixel = (gv * av.float()).sum(1).contiguous()          # [P]: one short reduction per row
                dg = self._dst(gamma)
                if dg is None:
                    dg = torch.empty(1, dtype=torch.float32, device=self.device)
                ops.sum_rows_f32(per_pixel, per_pixel.numel(), dg.view(1))
                self._give_grad(gamma, dg.view(gamma.shape))
                if x.needs_grad:
                    x.add_grad(g)
                if a.needs_grad:
                    da = self.new_act(a.N, a.H, a.W, a.C)
                    da.buf.copy_((gv * gamma.detach().float()).to(da.dtype))
                    a.add_grad(da)

            self.tape.append(bwd)
        return out

    def finish_forward(self) -> None:
        """End of the forward: `num_batches_tracked += 1` of every train-mode BatchNorm (batchnorm.py of torch,
        as `nn.BatchNorm2d.forward` does) in ONE multi-tensor launch instead of one 5 us kernel per layer."""
        if self._bn_counters:
            # a module applied twice in one forward (Multiresblock.batch_norm1, multiresunet.py:77, :81) counts twice
            uniq: Dict[int, list] = {}
            for t in self._bn_counters:
                uniq.setdefault(id(t), [t, 0])[1] += 1
            by_count: Dict[int, list] = {}
            for t, k in uniq.values():
                by_count.setdefault(k, []).append(t)
            for k, ts in by_count.items():
                torch._foreach_add_(ts, k)
            self._bn_counters = []

    def position_biases(self, attns: Sequence[Tuple[nn.Module, int]]) -> None:
        """Evaluate the continuous position bias of every (WindowAttention module, window_size) pair in one
        launch (they depend on parameters only) and, through the tape entry appended here -- the last one the
        backward reaches --, differentiate them in one launch once every attention has left its d(bias).
        window_attention() picks the results up; modules not announced here keep their own launches."""
        mods = []
        for attn, ws in attns:
            N = ws * ws
            cpb = attn.cpb
            heads = cpb.fc2.out_features
            idx = attn.log_relative_position_index[:N, :N].reshape(N * N, 2).contiguous()
            m = {"N": N, "attn": attn, "idx": idx, "w1": cpb.fc1.weight.detach(), "b1": cpb.fc1.bias.detach(),
                 "w2": cpb.fc2.weight.detach(), "b2": cpb.fc2.bias.detach(),
                 "bias": torch.empty((heads, N * N), dtype=torch.float32, device=self.device)}
            mods.append(m)
            self._cpb[attn] = m
        if not mods:
            return
        ops.cpb_fwd_batched(mods)
        if self.record:
            def bwd():
                self._flush_rowsums()      # the d(bias) of this range's attentions may still be partial rows
                live = [m for m in mods if m.get("G") is not None]
                for m in live:
                    cpb = m["attn"].cpb
                    m["params"] = (cpb.fc1.weight, cpb.fc1.bias, cpb.fc2.weight, cpb.fc2.bias)
                    for name, p_ in zip(("dw1", "db1", "dw2", "db2"), m["params"]):
                        d_ = self._dst(p_)
                        m[name] = d_ if d_ is not None else torch.empty(p_.shape, dtype=torch.float32, device=self.device)
                if live:
                    ops.cpb_bwd_batched(live)
                for m in live:
                    for name, p_ in zip(("dw1", "db1", "dw2", "db2"), m["params"]):
                        self._give_grad(p_, m[name])
                    m["G"] = None

            self.tape.append(bwd)

    def window_attention(self, x: Act, attn: nn.Module, heads: int, ws: int, shift: int) -> Act:
        """WindowAttention on the un-partitioned token tensor (swin_unet_v2.py:127-159 with the roll /
        window_partition / window_reverse of :246-262 folded into the core kernel's addressing):
        qkv Linear -> cosine-attention core -> proj Linear.  The continuous position bias
        cpb(log_relative_position_index) is a function of parameters only ((ws^2)^2 x 2 inputs): it is
        evaluated with torch ops like the weight re-packing, its parameter gradients by autograd from
        the kernel's d(bias)."""
        N = ws * ws
        qkv = self.linear(x, attn.qkv)
        p_attn = attn.attn_drop.p if self.training else 0.0
        if p_attn > 0.0 or x.C // heads != 32:
            # (rounds 2-4 ran these through torch GEMMs and autograd; the product has ONE backend, and it has no window
            # core for them: dropout on the attention probabilities inside the softmax-times-V product, head widths
            # other than 32)
            raise NotImplementedError(
                f"swin_unet_v2 on the HIP engine: the window-attention kernels take head_dim 32 (got {x.C // heads}) and no "
                f"attention dropout in training (attn_drop_rate={attn.attn_drop.p}); drop_rate / drop_path_rate are supported")
        cpb = attn.cpb
        pre = self._cpb.get(attn)                      # evaluated by position_biases() at the start of the forward
        if pre is not None and pre["N"] == N:
            idx, bias = pre["idx"], pre["bias"].view(heads, N, N)
        else:
            pre = None
            w1, b1, w2, b2 = (t.detach() for t in (cpb.fc1.weight, cpb.fc1.bias, cpb.fc2.weight, cpb.fc2.bias))
            idx = attn.log_relative_position_index[:N, :N].reshape(N * N, 2).contiguous()
            bias = ops.cpb_fwd(idx, w1, b1, w2, b2).view(heads, N, N)
        tau = attn.tau.detach()
        o = self.new_act(x.N, x.H, x.W, x.C)
        lse = ops.winattn_fwd(qkv, tau, bias, o, heads, ws, shift, scale=attn.scale)
        if self.record:
            def bwd():
                g = self._total_grad(o)
                if g is None:
                    return
                dqkv = self.new_act(qkv.N, qkv.H, qkv.W, qkv.C)
                whole = tuple(attn.tau.shape) == (heads, N, N)
                # d(bias) and d(tau) leave the kernel as per-workgroup rows; nothing reads their sums before the position
                # biases' own backward (the last tape entry) and the optimizer: summed with the range's other rows in one
                # launch (12 launches of 5 us per swin step)
                late = pre is not None and whole
                dbias, dtau = ops.winattn_bwd(qkv, tau, bias, o, lse, g, dqkv, heads, ws, shift,
                                              dtau=self._dst(attn.tau) if whole else None, scale=attn.scale,
                                              defer=self._rowsum_items if late else None)
                qkv.add_grad(dqkv)
                if late:
                    def give():
                        self._give_grad(attn.tau, dtau)
                        pre["G"] = dbias.view(heads, N * N)     # differentiated with all the others (position_biases)
                    self._after_rowsums(give)
                    return
                if not whole:                                                   # window clipped to the map size
                    full = torch.zeros_like(attn.tau)
                    full[:, :N, :N] = dtau
                    dtau = full
                self._give_grad(attn.tau, dtau)
                if pre is not None:
                    pre["G"] = dbias.view(heads, N * N)  # differentiated with all the others (position_biases)
                    return
                gs = []
                for p_ in (cpb.fc1.weight, cpb.fc1.bias, cpb.fc2.weight, cpb.fc2.bias):
                    d_ = self._dst(p_)
                    gs.append(d_ if d_ is not None else torch.empty(p_.shape, dtype=torch.float32, device=self.device))
                ops.cpb_bwd(idx, w1, b1, w2, dbias.view(heads, N * N), *gs)
                for p_, g_ in zip((cpb.fc1.weight, cpb.fc1.bias, cpb.fc2.weight, cpb.fc2.bias), gs):
                    self._give_grad(p_, g_)

            self.tape.append(bwd)
        return self.dropout(self.linear(o, attn.proj), attn.proj_drop.p)

    def resize_bilinear(self, x: Act, out: Act, align_corners: bool = False) -> Act:
        """out = F.interpolate(x, size=out's, mode='bilinear', align_corners=...), written into its
        concat slot.  Reference: _upsample_like (u2net.py:19-22; align_corners=False),
        nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) (nested_unet.py:32)."""
        assert x.N == out.N and x.C == out.C
        ops.bilinear_fwd(x, out, align_corners)
        if self.record and x.needs_grad:
            def bwd():
                gs = self._sum_grads(out, 1)
                if not gs:
                    return
                dx = self.new_act(x.N, x.H, x.W, x.C)
                ops.bilinear_bwd(gs[0], dx, align_corners)
                x.add_grad(dx)

            self.tape.append(bwd)
        return out

    def copy_into(self, src: Act, dst: Act) -> Act:
        """dst (a slot of a concat buffer) = src: a tensor that appears in SEVERAL torch.cat calls of the
        reference (UNet++'s dense skips, nested_unet.py:80-93) lives in the slot of its first consumer and is
        copied into the others; the slot's gradient flows back to `src` without a copy."""
        assert (src.N, src.H, src.W, src.C) == (dst.N, dst.H, dst.W, dst.C) and src.dtype == dst.dtype
        ops.resample2(src, dst, ops.RESAMPLE_COPY)
        if self.record and src.needs_grad:
            def bwd():
                g = self._total_grad(dst)
                if g is not None:
                    src.add_grad(g)

            self.tape.append(bwd)
        return dst

    def u2net_heads(self, feats: Sequence[Act], sides: Sequence[nn.Conv2d], fuse: nn.Conv2d) -> List[torch.Tensor]:
        """The six 3x3 side heads, their bilinear resize to the first head's resolution, and the 1x1
        fuse convolution over their concat.  Returns [d0, d1, ..., d6] as (N, K, H, W) fp32 (d1..d6
        are channel slices of one concat buffer).  Reference: U2NET.forward, u2net.py:277-298."""
        N, H, W = feats[0].N, feats[0].H, feats[0].W
        S, K, HW = len(feats), sides[0].out_channels, H * W
        assert fuse.kernel_size == (1, 1) and fuse.in_channels == S * K and fuse.out_channels == K
        dev = self.device
        dcat = torch.empty((N, S * K, H, W), dtype=torch.float32, device=dev)
        taps = torch.empty(N * 9 * HW, dtype=torch.float32, device=dev)
        for s_, (f, conv) in enumerate(zip(feats, sides)):
            assert conv.kernel_size == (3, 3) and conv.padding == (1, 1) and conv.in_channels == f.C
            wt = conv.weight.detach()
            hw = f.H * f.W
            low = None if (f.H, f.W) == (H, W) else torch.empty((N, K, f.H, f.W), dtype=torch.float32, device=dev)
            for o in range(K):
                wp = wt.data_ptr() + o * f.C * 9 * 4
                bp = conv.bias.detach().data_ptr() + 4 * o if conv.bias is not None else None
                plane = dcat.data_ptr() + (s_ * K + o) * HW * 4
                if low is None:
                    ops.sideconv_fwd(f, wp, bp, taps, plane, S * K * HW)
                else:
                    lp = low.data_ptr() + o * hw * 4
                    ops.sideconv_fwd(f, wp, bp, taps, lp, K * hw)
                    ops.bilinear_planes(lp, K * hw, f.H, f.W, plane, S * K * HW, H, W, N)
        wf = fuse.weight.detach().reshape(K, S * K)
        d0 = ops.fuse1x1_fwd(dcat, wf, fuse.bias.detach() if fuse.bias is not None else None)
        outs = [d0] + [dcat[:, s_ * K:(s_ + 1) * K] for s_ in range(S)]

        if self.record:
            def bwd(*gs: Optional[torch.Tensor]):
                if all(g is None for g in gs):
                    return
                g0 = gs[0].contiguous().float() if gs[0] is not None else None
                extras = [g.contiguous().float() if g is not None else None for g in gs[1:]]
                dwf, dbf = self._dst(fuse.weight), (self._dst(fuse.bias) if fuse.bias is not None else None)
                if dwf is None:
                    dwf = torch.empty(fuse.weight.shape, dtype=torch.float32, device=dev)
                if dbf is None and fuse.bias is not None:
                    dbf = torch.empty(K, dtype=torch.float32, device=dev)
                dcat_g = ops.fuse1x1_bwd(dcat, wf, g0, extras, dwf, dbf)
                self._give_grad(fuse.weight, dwf)
                if fuse.bias is not None:
                    self._give_grad(fuse.bias, dbf)
                for s_, (f, conv) in enumerate(zip(feats, sides)):
                    hw = f.H * f.W
                    dw, db = self._dst(conv.weight), (self._dst(conv.bias) if conv.bias is not None else None)
                    if dw is None:
                        dw = torch.empty(conv.weight.shape, dtype=torch.float32, device=dev)
                    if db is None and conv.bias is not None:
                        db = torch.empty(K, dtype=torch.float32, device=dev)
                    low = None if (f.H, f.W) == (H, W) else torch.empty((N, K, f.H, f.W), dtype=torch.float32, device=dev)
                    wt = conv.weight.detach()
                    for o in range(K):
                        plane = dcat_g.data_ptr() + (s_ * K + o) * HW * 4
                        if low is None:
                            gp_, gi = plane, S * K * HW
                        else:
                            gp_, gi = low.data_ptr() + o * hw * 4, K * hw
                            ops.bilinear_planes(plane, S * K * HW, f.H, f.W, gp_, gi, H, W, N, backward=True)
                        dx = self.new_act(f.N, f.H, f.W, f.C) if f.needs_grad else None
                        ops.sideconv_bwd(f, wt.data_ptr() + o * f.C * 9 * 4, gp_, gi, dx,
                                         dw.data_ptr() + o * f.C * 9 * 4,
                                         db.data_ptr() + 4 * o if db is not None else None)
                        if dx is not None:
                            f.add_grad(dx)
                    self._give_grad(conv.weight, dw)
                    if conv.bias is not None:
                        self._give_grad(conv.bias, db)

            self._heads.append((bwd, len(outs)))
        return outs

    def out_conv(self, x: Act, conv: nn.Conv2d, sole_reader: bool = False) -> torch.Tensor:
        """1x1 convolution to the logits, (N, K, H, W) fp32.  Reference: OutConv (common_layers.py:125).  sole_reader: as
        in conv_bn_relu (the last decoder block's output feeds only the head)."""
        assert conv.kernel_size == (1, 1) and conv.in_channels == x.C
        K = conv.out_channels
        w = conv.weight.detach().reshape(K, x.C)
        b = conv.bias.detach() if conv.bias is not None else torch.zeros(K, device=self.device)
        xf = getattr(x, "lazy", None)      # x is the raw output of a convolution (conv_bn_relu(defer_apply=this head))
        assert xf is None or sole_reader, "a lazy activation goes to the head it was deferred for"
        logits = ops.outconv_fwd(x, w, b, xform=xf)
        if self.record:
            def bwd(g_logits: torch.Tensor):
                dx = self.new_act(x.N, x.H, x.W, x.C) if (x.needs_grad or xf is not None) else None
                dwt = self._dst(conv.weight)
                dbt = self._dst(conv.bias) if conv.bias is not None else None
                src = getattr(x, "bn_src", None) if (sole_reader and (self.fuse_bn_reduce_convt or xf is not None)) else None
                dw, db = ops.outconv_bwd(x, w, g_logits.contiguous().float(), dx,
                                         dwt.view(K, x.C) if dwt is not None else None, dbt, bnred=src, lazy=xf is not None)
                self._give_grad(conv.weight, dwt if dwt is not None else dw.reshape(conv.weight.shape))
                if conv.bias is not None:
                    self._give_grad(conv.bias, db)
                if dx is not None:
                    x.add_grad(dx)

            self._heads.append((bwd, 1))
        return logits

    def layer_norm_head(self, x: Act, ln: nn.LayerNorm, conv: nn.Conv2d, *, mode: int = L.LN_PLAIN, r: int = 1) -> torch.Tensor:
        """conv(LayerNorm(x)) for a 1x1 `conv` to the logits, (N, K, H, W) fp32, without materialising the
        normalised tensor (FinalPatchExpand_X4's norm + `output`, swin_unet_v2.py:385 / :753): at B=16
        256x256 that tensor is 201 MB each way.  Falls back to layer_norm + out_conv for shapes the fused
        kernels do not take."""
        C, K = ln.normalized_shape[0], conv.out_channels
        assert conv.kernel_size == (1, 1) and conv.in_channels == C
        if not ops.ln_head_supported(C, K, self.dtype):
            return self.out_conv(self.layer_norm(x, ln, mode=mode, r=r), conv)
        if mode == L.LN_EXPAND:
            assert x.C == r * r * C
            N, Ho, Wo = x.N, x.H * r, x.W * r
        else:
            assert mode == L.LN_PLAIN and x.C == C
            N, Ho, Wo = x.N, x.H, x.W
        gamma, beta = ln.weight.detach(), ln.bias.detach()
        w = conv.weight.detach().reshape(K, C)
        b = conv.bias.detach() if conv.bias is not None else None
        logits, stats = ops.ln_head_fwd(x, gamma, beta, w, b, N, Ho, Wo, C, mode=mode, r=r, eps=ln.eps)
        if self.record:
            def bwd(g_logits: torch.Tensor):
                dx = self.new_act(x.N, x.H, x.W, x.C)
                dwt = self._dst(conv.weight)
                outs = ops.ln_head_bwd(x, gamma, beta, w, stats, g_logits.contiguous().float(), dx, mode=mode, r=r,
                                       eps=ln.eps, dgamma=self._dst(ln.weight), dbeta=self._dst(ln.bias),
                                       dw=dwt.view(K, C) if dwt is not None else None,
                                       db=self._dst(conv.bias) if conv.bias is not None else None)
                self._give_grad(conv.weight, dwt if dwt is not None else outs[2].reshape(conv.weight.shape))
                if conv.bias is not None:
                    self._give_grad(conv.bias, outs[3])
                self._give_grad(ln.weight, outs[0])
                self._give_grad(ln.bias, outs[1])
                if x.needs_grad:
                    x.add_grad(dx)

            self._heads.append((bwd, 1))
        return logits

    # ------------------------------------------------------------------ backward
    def backward(self, grad_outputs: Sequence[Optional[torch.Tensor]]) -> Dict[nn.Parameter, torch.Tensor]:
        """Run the recorded tape in reverse.  `grad_outputs` pairs with the out_conv heads in
        emission order."""
        self.backward_range(grad_outputs, len(self.tape), 0)
        self.tape.clear()
        return self.param_grads

    def backward_range(self, grad_outputs: Optional[Sequence[Optional[torch.Tensor]]], hi: int, lo: int) -> None:
        """Part of the backward: the heads (when `grad_outputs` is given), then tape entries
        hi-1, hi-2, ..., lo.  Lets a caller run the backward in phases (graph.PhasedStep) so that the
        gradients of a finished phase can be all-reduced while the next phase computes."""
        if grad_outputs is not None:
            self._cur_entry = len(self.tape)
            gl, pos, calls = list(grad_outputs), 0, []
            for fn, n in self._heads:
                calls.append((fn, gl[pos:pos + n]))
                pos += n
            assert pos == len(gl)
            for fn, gs in reversed(calls):
                if any(g is not None for g in gs):
                    fn(*gs)
        for i in range(hi - 1, lo - 1, -1):
            self._cur_entry = i
            self.tape[i]()
        self._flush_colsums()
        self._cur_entry = -1
