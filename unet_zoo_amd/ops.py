"""Python face of the C ABI: each function launches exactly one HIP kernel of libunetzoo_hip.so on
torch's current stream.  Activations are :class:`Act` views — NHWC, possibly a channel window of
a wider buffer (that is how the reference's ``torch.cat`` skip-concats, common_layers.py:115,
are never materialised).
"""
from __future__ import annotations

from ctypes import byref
from typing import List, Optional, Sequence

import math

import torch

from . import _lib as L


class Act:
    """NHWC activation: element (pixel p, channel c) at ``buf[p, off + c]``; ``buf`` is (P, ld)."""

    __slots__ = ("buf", "off", "C", "N", "H", "W", "grads", "parts", "rparts", "needs_grad", "colsums", "bn_src",
                 "bn_partials", "lazy")

    def __init__(self, buf: torch.Tensor, off: int, C: int, N: int, H: int, W: int,
                 needs_grad: bool = True):
        assert buf.dim() == 2 and buf.is_contiguous() and buf.shape[0] == N * H * W
        assert 0 <= off and off + C <= buf.shape[1]
        self.buf, self.off, self.C, self.N, self.H, self.W = buf, off, C, N, H, W
        self.grads: List["Act"] = []        # gradient contributions (same resolution as self)
        self.parts: Optional[Sequence["Act"]] = None  # set on the full view of a concat buffer
        self.rparts: Optional[Sequence[tuple]] = None  # (part, first row): a token-concat whose producers wrote row blocks
        self.needs_grad = needs_grad
        # optional (partials [G, 2, Ctot], channel offset): per-channel sums of this tensor that its
        # producing kernel delivered for free (used for ConvTranspose2d bias gradients)
        self.colsums = None
        # (scale, shift): the buffer holds the RAW output of a convolution and stands for relu(buf * scale + shift), which
        # nobody has written down (Engine.conv_bn_relu(defer_apply=True)); only the kernels that read through that map
        # (conv_igemm / wgrad with xform=...) may take it
        self.lazy = None

    @property
    def ld(self) -> int:
        return self.buf.shape[1]

    @property
    def P(self) -> int:
        return self.buf.shape[0]

    @property
    def dtype(self) -> torch.dtype:
        return self.buf.dtype

    def ptr(self) -> int:
        return self.buf.data_ptr() + self.off * self.buf.element_size()

    def window(self, off: int, C: int) -> "Act":
        w = Act(self.buf, self.off + off, C, self.N, self.H, self.W, self.needs_grad)
        if self.colsums is not None:
            w.colsums = (self.colsums[0], self.colsums[1] + off)
        return w

    def rows(self, r0: int, N: int, H: int, W: int) -> "Act":
        """rows [r0, r0 + N*H*W) as an (N, H, W, C) tensor of their own (token-concats along dim -2)"""
        return Act(self.buf[r0:r0 + N * H * W], self.off, self.C, N, H, W, self.needs_grad)

    def channel_sums(self, out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """sum over pixels per channel (fp32) if the producer recorded partial sums, else None; out: written in place
        (a parameter's .grad: no copy launch afterwards)"""
        if self.colsums is None:
            return None
        part, o = self.colsums
        # own kernel rather than part[:, 0, o:o+C].sum(0): no library reduction on the (capturable) hot path --
        # torch's multi-block reductions reset their semaphores with a memset node, which a replayed hipGraph
        # executes correctly only once on this stack
        if out is None:
            out = torch.empty(self.C, dtype=torch.float32, device=part.device)
        assert out.numel() == self.C and out.dtype == torch.float32 and out.is_contiguous()
        L.check(L.load().uz_sum_rows_f32_ld(part.data_ptr() + 4 * o, part.shape[1] * part.shape[2], part.shape[0],
                                            self.C, out.data_ptr(), self.C, None, L.stream_ptr()), "uz_sum_rows_f32_ld")
        return out

    def add_grad(self, g: "Act") -> None:
        """Register a gradient contribution; a concat view forwards channel windows to its parts."""
        if self.parts is not None:
            o = 0
            for part in self.parts:
                part.add_grad(g.window(o, part.C))
                o += part.C
        elif self.rparts is not None:
            for part, r0 in self.rparts:
                part.add_grad(g.rows(r0, part.N, part.H, part.W))
        else:
            self.grads.append(g)

    def dense(self) -> torch.Tensor:
        """(N, C, H, W) fp32 copy — test/debug helper, not used on the hot path."""
        v = self.buf[:, self.off:self.off + self.C].float()
        if self.lazy is not None:   # the tensor this view stands for, as uz_bn_relu_apply would have stored it
            v = torch.relu(torch.addcmul(self.lazy[1], v, self.lazy[0])).to(self.buf.dtype).float()
        return v.reshape(self.N, self.H, self.W, self.C).permute(0, 3, 1, 2).contiguous()


def new_act(N: int, H: int, W: int, C: int, dtype: torch.dtype, device, needs_grad=True) -> Act:
    return Act(torch.empty((N * H * W, C), dtype=dtype, device=device), 0, C, N, H, W, needs_grad)


def act_from_nchw(x: torch.Tensor, dtype: torch.dtype) -> Act:
    """Test helper: (N,C,H,W) tensor -> NHWC Act."""
    N, C, H, W = x.shape
    buf = x.permute(0, 2, 3, 1).reshape(N * H * W, C).to(dtype).contiguous()
    return Act(buf, 0, C, N, H, W)


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# ------------------------------------------------------------------------------------------------
# Optional per-launch timing (bench.py): HIP events recorded on the launch stream around each
# kernel, grouped by kernel family, with the algorithmic flops / bytes of exactly those launches.
_prof_on = False
_prof_log: list = []
_prof_scope = None          # label of the model block being emitted (profile_scope), None outside
_prof_scopes: dict = {}     # {label: {"ms", "launches"}} of the last profile_end()


class profile_scope:
    """with profile_scope("doubleconv_l1"): ... -- the timed launches inside also count towards that label
    (profile_scopes() after profile_end()); bench.py's block-level figure for the north-star DoubleConv"""

    def __init__(self, label: str):
        self.label = label

    def __enter__(self):
        global _prof_scope
        self.prev, _prof_scope = _prof_scope, self.label
        return self

    def __exit__(self, *exc):
        global _prof_scope
        _prof_scope = self.prev
        return False


def profile_scopes() -> dict:
    return dict(_prof_scopes)


def profile_begin() -> None:
    global _prof_on
    _prof_log.clear()
    _prof_on = True


def profile_end() -> dict:
    """Returns {family: {"ms", "launches", "flops", "bytes"}}; synchronises the device."""
    global _prof_on
    _prof_on = False
    torch.cuda.synchronize()
    out: dict = {}
    _prof_scopes.clear()
    for name, e0, e1, fl, by, scope in _prof_log:
        d = out.setdefault(name, {"ms": 0.0, "launches": 0, "flops": 0.0, "bytes": 0.0})
        ms = e0.elapsed_time(e1)
        d["ms"] += ms
        d["launches"] += 1
        d["flops"] += fl
        d["bytes"] += by
        if scope is not None:
            sd = _prof_scopes.setdefault(scope, {"ms": 0.0, "launches": 0, "kernels": {}})
            sd["ms"] += ms
            sd["launches"] += 1
            sd["kernels"][name] = sd["kernels"].get(name, 0.0) + ms
    _prof_log.clear()
    return out


class _Timed:
    __slots__ = ("name", "flops", "bytes", "e0", "e1")

    def __init__(self, name: str, flops: float, nbytes: float):
        self.name, self.flops, self.bytes = name, flops, nbytes

    def __enter__(self):
        if _prof_on:
            # both events exist (recorded once) before the library re-records them at the kernels' own begin / end
            # (uz_profile_arm: hipExtLaunchKernelGGL start / stop events) -- the bracket is then first-kernel-begin to
            # last-kernel-end, as a kernel trace reports it, without the dispatch latency of an event recorded in front
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
            self.e1.record()
            L.check(L.load().uz_profile_arm(self.e0.cuda_event, self.e1.cuda_event), "uz_profile_arm")
        return self

    def __exit__(self, *exc):
        if _prof_on:
            n = L.load().uz_profile_disarm()
            if n == 0:          # nothing was launched through the library inside this scope: plain bracket
                self.e1.record()
            _prof_log.append((self.name, self.e0, self.e1, self.flops, self.bytes, _prof_scope))
        return False


def _tname(dt: torch.dtype) -> str:
    return "bf16" if dt == torch.bfloat16 else "f32"


# ------------------------------------------------------------------------------------------------
def pack_weights(w: torch.Tensor, mode: int, dtype: torch.dtype, kpad: int = 0) -> torch.Tensor:
    """fp32 master weights (reference layout) -> kernel layout in the run dtype."""
    L.require_cuda(w)
    assert w.dtype == torch.float32 and w.is_contiguous()
    d0, d1 = w.shape[0], w.shape[1]
    T = w.numel() // (d0 * d1)
    if mode in (L.PACK_CONV_FWD, L.PACK_CONV_DGRAD, L.PACK_IM2COL):
        Co, Ci = d0, d1
    else:  # ConvTranspose2d weight is (Cin, Cout, kh, kw)
        Ci, Co = d0, d1
    if mode == L.PACK_CONV_FWD:
        shape = (Co, T * Ci)
    elif mode == L.PACK_CONV_DGRAD:
        shape = (Ci, T * Co)
    elif mode == L.PACK_CONVT_FWD:
        shape = (T * Co, Ci)
    elif mode == L.PACK_CONVT_DGRAD:
        shape = (Ci, T * Co)
    else:
        shape = (Co, kpad)
    dst = torch.empty(shape, dtype=dtype, device=w.device)
    L.check(L.load().uz_pack_weights(L.dtype_code(dtype), mode, w.data_ptr(), Co, Ci, T, kpad,
                                     dst.data_ptr(), L.stream_ptr()), "uz_pack_weights")
    return dst


def im2col3x3_nchw(x: torch.Tensor, kpad: int, dtype: torch.dtype) -> Act:
    L.require_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
    N, C, H, W = x.shape
    out = new_act(N, H, W, kpad, dtype, x.device, needs_grad=False)
    with _Timed("im2col3x3_nchw", 0.0, 4.0 * x.numel() + out.buf.element_size() * out.buf.numel()):
        L.check(L.load().uz_im2col3x3_nchw(L.dtype_code(dtype), x.data_ptr(), N, C, H, W, kpad,
                                           out.buf.data_ptr(), L.stream_ptr()), "uz_im2col3x3_nchw")
    return out


def conv_first_supported(dtype: torch.dtype, C: int, Cout: int) -> bool:
    """the direct first-convolution kernels (uz_conv_first.hip) take this layer: bf16 run mode, C <= 3, Cout in {32, 64}"""
    return bool(L.load().uz_conv3x3_first_supported(L.dtype_code(dtype), C, Cout))


def conv_first_fwd(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], y: Act, want_stats: bool):
    """y = Conv2d(k3, p1)(x) for the fp32 NCHW network input x (no im2col buffer); returns the BatchNorm partial rows
    (rows, 2, Cout) of the stored values when want_stats"""
    L.require_cuda(x, w, y.buf)
    assert x.dtype == torch.float32 and x.is_contiguous() and w.dtype == torch.float32 and w.is_contiguous()
    N, C, H, W = x.shape
    Cout = w.shape[0]
    assert (y.N, y.H, y.W, y.C) == (N, H, W, Cout) and tuple(w.shape) == (Cout, C, 3, 3)
    lib = L.load()
    stats = None
    if want_stats:
        rows = L.check_count(lib.uz_conv3x3_first_rows(N, H, W), "uz_conv3x3_first_rows")
        stats = torch.empty((rows, 2, Cout), dtype=torch.float32, device=x.device)
    with _Timed("conv3x3_first_bf16", 2.0 * N * H * W * 9 * C * Cout, 4.0 * x.numel() + 2.0 * y.P * Cout):
        L.check(lib.uz_conv3x3_first_fwd(L.dtype_code(y.dtype), x.data_ptr(), N, C, H, W, w.data_ptr(), _p(bias), Cout,
                                         y.ptr(), y.ld, _p(stats), L.stream_ptr()), "uz_conv3x3_first_fwd")
    return stats


def conv_first_wgrad(x: torch.Tensor, dy: Act, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """d(loss)/d(weight) (Cout, C, 3, 3) fp32 of the first convolution from the fp32 NCHW input and the output gradient"""
    L.require_cuda(x, dy.buf)
    N, C, H, W = x.shape
    Cout = dy.C
    lib = L.load()
    ws_bytes = L.check_count(lib.uz_conv3x3_first_wgrad_workspace_bytes(N, H, W, Cout), "uz_conv3x3_first_wgrad_workspace_bytes")
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device)
    if out is None:
        out = torch.empty((Cout, C, 3, 3), dtype=torch.float32, device=x.device)
    assert out.numel() == Cout * C * 9 and out.is_contiguous() and out.dtype == torch.float32
    with _Timed("wgrad_first_bf16", 2.0 * N * H * W * 9 * C * Cout, 4.0 * x.numel() + 2.0 * dy.P * Cout):
        L.check(lib.uz_conv3x3_first_wgrad(L.dtype_code(dy.dtype), x.data_ptr(), N, C, H, W, dy.ptr(), dy.ld, Cout,
                                           out.data_ptr(), ws.data_ptr(), L.stream_ptr()), "uz_conv3x3_first_wgrad")
    return out


def conv_kernel_name(d, with_workspace: bool = False) -> str:
    """uz_conv_igemm_kernel_name(): the kernel family the library's plan picks for a ConvDesc (labels of the per-kernel
    timing of bench.py; tests use it to assert which generation they exercise)"""
    import ctypes
    buf = ctypes.create_string_buffer(96)
    L.check_count(L.load().uz_conv_igemm_kernel_name(byref(d), 1 if with_workspace else 0, buf, 96),
                  "uz_conv_igemm_kernel_name")
    return buf.value.decode()


def conv_igemm(x: Act, w_packed: torch.Tensor, bias: Optional[torch.Tensor], y: Act, *,
               ntaps: int, dil: int = 1, taps_mode: int = L.TAPS_CONV,
               store_mode: int = L.STORE_PLAIN, nout: Optional[int] = None, co: int = 0,
               want_stats: bool = False, res: Optional[Act] = None,
               bnred: Optional[tuple] = None, xform: Optional[tuple] = None) -> Optional[torch.Tensor]:
    """y = conv(x, w) + bias [+ res] on the matrix cores; returns the BN partial-sum rows if asked.  res: a tensor
    of y's shape added in the GEMM epilogue (uz_conv_igemm_res), or by a separate add where that kernel does not apply.
    bnred = (bn_y, vec): y is the gradient of relu(bn(bn_y)); the kernel's epilogue accumulates the two sums of that
    BatchNorm's backward (uz_conv_igemm_bnred) and their partial rows are returned -- or None when the problem is not
    one of the kernels that can (the caller then runs the stand-alone reduction).
    xform = (scale, shift): x is the RAW output of the preceding convolution and the kernel reads it through that layer's
    BatchNorm + ReLU, relu(x * scale + shift) per input channel (uz_conv_igemm_xf); the caller has asked
    conv_xform_supported() first."""
    L.require_cuda(x.buf, w_packed, y.buf)
    lib = L.load()
    if taps_mode == L.TAPS_CONV:
        N, H, W = x.N, x.H, x.W
    elif taps_mode == L.TAPS_CONV_UP2:       # x is the half-resolution tensor
        N, H, W = x.N, 2 * x.H, 2 * x.W
    elif taps_mode == L.TAPS_CONV_S2:        # Conv2d(k3, stride 2, padding 1): ceil(H / 2)
        N, H, W = x.N, (x.H + 1) // 2, (x.W + 1) // 2
    else:
        N, H, W = x.N, x.H // 2, x.W // 2
    shuffle = store_mode == L.STORE_SHUFFLE2X2   # destination grid may be one row / column larger (zero pad)
    d = L.ConvDesc(L.dtype_code(x.dtype), N, H, W, x.H, x.W, x.C, x.ld,
                   nout if nout is not None else y.C, y.ld, ntaps, taps_mode, dil, store_mode, co,
                   y.H if shuffle else 0, y.W if shuffle else 0)
    assert w_packed.dtype == x.dtype and y.dtype == x.dtype
    assert w_packed.shape == (d.Nout, ntaps * x.C), (tuple(w_packed.shape), d.Nout, ntaps, x.C)
    stats = None
    if want_stats:
        gm = L.check_count(lib.uz_conv_igemm_ws_grid_m(byref(d)), "uz_conv_igemm_ws_grid_m")
        stats = torch.empty((gm, 2, d.Nout), dtype=torch.float32, device=x.buf.device)
    wsb = L.check_count(lib.uz_conv_igemm_workspace_bytes(byref(d)), "uz_conv_igemm_workspace_bytes")
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=x.buf.device) if wsb > 0 else None
    M, K, es = N * H * W, ntaps * x.C, x.buf.element_size()
    kname = conv_kernel_name(d, ws is not None)   # the family the library's own plan launches for this descriptor
    if xform is not None:
        assert bnred is None and res is None
        with _Timed(kname + "_xf", 2.0 * M * d.Nout * K, es * (x.P * x.C + M * d.Nout + d.Nout * K)):
            L.check(lib.uz_conv_igemm_xf(byref(d), x.ptr(), xform[0].data_ptr(), xform[1].data_ptr(), w_packed.data_ptr(),
                                         _p(bias), y.ptr(), _p(stats), L.stream_ptr()), "uz_conv_igemm_xf")
        return stats
    if bnred is not None:
        assert bias is None and res is None and not want_stats
        bn_y, vec4 = bnred
        if not lib.uz_conv_igemm_bnred_supported(byref(d)) or (bn_y.P, bn_y.C) != (y.P, y.C):
            bnred = None
        else:
            gm = L.check_count(lib.uz_conv_igemm_grid_m(byref(d)), "uz_conv_igemm_grid_m")     # the unsplit launch's rows
            part = torch.empty((gm, 2, d.Nout), dtype=torch.float32, device=x.buf.device)
            with _Timed(kname + "_bnred", 2.0 * M * d.Nout * K, es * (x.P * x.C + 2 * M * d.Nout + d.Nout * K)):
                L.check(lib.uz_conv_igemm_bnred(byref(d), x.ptr(), w_packed.data_ptr(), y.ptr(), bn_y.ptr(), bn_y.ld,
                                                vec4[0].data_ptr(), vec4[1].data_ptr(), vec4[2].data_ptr(),
                                                vec4[3].data_ptr(), part.data_ptr(), L.stream_ptr()),
                        "uz_conv_igemm_bnred")
            return part
    if res is not None:
        assert (res.P, res.C) == (y.P, y.C) and res.dtype == y.dtype and not want_stats
        if kname.startswith("gemm_dma") and store_mode == L.STORE_PLAIN:
            with _Timed(kname, 2.0 * M * d.Nout * K, es * (x.P * x.C + 2 * M * d.Nout + d.Nout * K)):
                L.check(lib.uz_conv_igemm_res_ws(byref(d), x.ptr(), w_packed.data_ptr(), _p(bias), res.ptr(), res.ld, y.ptr(),
                                                 _p(ws), L.stream_ptr()), "uz_conv_igemm_res_ws")
            return None
    with _Timed(kname, 2.0 * M * d.Nout * K,
                        es * (x.P * x.C + M * d.Nout + d.Nout * K)):
        L.check(lib.uz_conv_igemm_ws(byref(d), x.ptr(), w_packed.data_ptr(), _p(bias), y.ptr(), _p(stats),
                                     _p(ws), L.stream_ptr()), "uz_conv_igemm_ws")
    if res is not None:
        add_acts(y, res, y)
    return stats


def conv_xform_supported(x: Act, nout: int, ldy: int, *, upsample: bool = False) -> bool:
    """whether uz_conv_igemm_xf takes the 3x3 convolution of x (channels, pixel grid, run dtype) to nout channels"""
    H, W = (2 * x.H, 2 * x.W) if upsample else (x.H, x.W)
    d = L.ConvDesc(L.dtype_code(x.dtype), x.N, H, W, x.H, x.W, x.C, x.ld, nout, ldy, 9,
                   L.TAPS_CONV_UP2 if upsample else L.TAPS_CONV, 1, L.STORE_PLAIN, 0, 0, 0)
    return bool(L.load().uz_conv_igemm_xf_supported(byref(d)))


def wgrad_kernel_name(d) -> str:
    """uz_wgrad_kernel_name(): the kernel family the library's plan picks for a WgradDesc (labels of bench.py's per-kernel
    timing; tests use it to assert which kernel they exercise)"""
    import ctypes
    buf = ctypes.create_string_buffer(96)
    L.check_count(L.load().uz_wgrad_kernel_name(byref(d), buf, 96), "uz_wgrad_kernel_name")
    return buf.value.decode()


def wgrad_xform_supported(Lt: Act, Rt: Act, ntaps: int, *, taps_mode: int = L.TAPS_CONV, dil: int = 1) -> bool:
    """whether uz_wgrad_xf takes this weight-gradient problem (R read through a BatchNorm + ReLU)"""
    d = L.WgradDesc(L.dtype_code(Lt.dtype), Lt.N, Lt.H, Lt.W, Rt.H, Rt.W, Lt.C, Lt.ld, Rt.C, Rt.ld, ntaps, taps_mode, dil)
    return bool(L.load().uz_wgrad_xf_supported(byref(d)))


def wgrad_xform_shapes_supported(N: int, H: int, W: int, Ci: int, ldl: int, Cj: int, ldr: int, dtype: torch.dtype) -> bool:
    """the same question before the output gradient exists: a (N, H, W, Ci) gradient against a (N, H, W, Cj) raw input"""
    d = L.WgradDesc(L.dtype_code(dtype), N, H, W, H, W, Ci, ldl, Cj, ldr, 9, L.TAPS_CONV, 1)
    return bool(L.load().uz_wgrad_xf_supported(byref(d)))


def wgrad(Lt: Act, Rt: Act, out_shape, *, ntaps: int, dil: int = 1,
          taps_mode: int = L.TAPS_CONV, out: Optional[torch.Tensor] = None, xform: Optional[tuple] = None) -> torch.Tensor:
    """out[i, j, tap] = sum_p L[p, i] * R[pix(p, tap), j]  (fp32, reference parameter layout).
    xform = (scale, shift): R is the RAW output of the convolution in front of this layer and is read through that
    layer's BatchNorm + ReLU (uz_wgrad_xf; ask wgrad_xform_supported first)."""
    L.require_cuda(Lt.buf, Rt.buf)
    lib = L.load()
    d = L.WgradDesc(L.dtype_code(Lt.dtype), Lt.N, Lt.H, Lt.W, Rt.H, Rt.W, Lt.C, Lt.ld, Rt.C, Rt.ld,
                    ntaps, taps_mode, dil)
    ws_bytes = L.check_count(lib.uz_wgrad_workspace_bytes(byref(d)), "uz_wgrad_workspace_bytes")
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=Lt.buf.device)
    if out is None:
        out = torch.empty(out_shape, dtype=torch.float32, device=Lt.buf.device)
    assert out.numel() == Lt.C * Rt.C * ntaps and out.is_contiguous() and out.dtype == torch.float32
    if xform is not None:
        def run(phase):
            L.check(lib.uz_wgrad_xf(byref(d), Lt.ptr(), Rt.ptr(), xform[0].data_ptr(), xform[1].data_ptr(), out.data_ptr(),
                                    ws.data_ptr(), L.stream_ptr(), phase), "uz_wgrad_xf")
        if not _prof_on:
            run(0)
            return out
        with _Timed(wgrad_kernel_name(d) + "_xf", 2.0 * Lt.P * Lt.C * Rt.C * ntaps,
                    Lt.buf.element_size() * (Lt.P * Lt.C + Rt.P * Rt.C) + 4.0 * out.numel()):
            run(1)
        with _Timed("wgrad_reduce", 0.0, float(ws_bytes) + 4.0 * out.numel()):
            run(2)
        return out
    if not _prof_on:
        L.check(lib.uz_wgrad(byref(d), Lt.ptr(), Rt.ptr(), out.data_ptr(), ws.data_ptr(), L.stream_ptr()), "uz_wgrad")
        return out
    # measured (bench.py's eager profile steps): the main kernel and the slab reduction in brackets of their own, so that
    # a family's average launch time is ONE kernel's, as a kernel trace reports it
    kname = wgrad_kernel_name(d)   # the family the library's own plan launches for this descriptor
    with _Timed(kname, 2.0 * Lt.P * Lt.C * Rt.C * ntaps,
                Lt.buf.element_size() * (Lt.P * Lt.C + Rt.P * Rt.C) + 4.0 * out.numel()):
        L.check(lib.uz_wgrad_phase(byref(d), Lt.ptr(), Rt.ptr(), out.data_ptr(), ws.data_ptr(), L.stream_ptr(), 1),
                "uz_wgrad_phase")
    with _Timed("wgrad_reduce", 0.0, float(ws_bytes) + 4.0 * out.numel()):
        L.check(lib.uz_wgrad_phase(byref(d), Lt.ptr(), Rt.ptr(), out.data_ptr(), ws.data_ptr(), L.stream_ptr(), 2),
                "uz_wgrad_phase")
    return out


def wgrad_multi(entries: Sequence) -> None:
    """uz_wgrad for every (L, R, out, ntaps, taps_mode, dil) entry, issued together (uz_wgrad_multi): the one-tap problems
    -- nn.Linear weight gradients -- share a few launches; out tensors are fp32, reference parameter layout"""
    if not entries:
        return
    lib = L.load()
    arr = (L.WgradItem * len(entries))()
    flops = nbytes = 0.0
    for i, (Lt, Rt, out, ntaps, taps_mode, dil) in enumerate(entries):
        L.require_cuda(Lt.buf, Rt.buf, out)
        assert out.numel() == Lt.C * Rt.C * ntaps and out.is_contiguous() and out.dtype == torch.float32
        d = L.WgradDesc(L.dtype_code(Lt.dtype), Lt.N, Lt.H, Lt.W, Rt.H, Rt.W, Lt.C, Lt.ld, Rt.C, Rt.ld, ntaps, taps_mode, dil)
        arr[i] = L.WgradItem(d, Lt.ptr(), Rt.ptr(), out.data_ptr())
        flops += 2.0 * Lt.P * Lt.C * Rt.C * ntaps
        nbytes += Lt.buf.element_size() * (Lt.P * Lt.C + Rt.P * Rt.C) + 4.0 * out.numel()
    ws_bytes = L.check_count(lib.uz_wgrad_multi_workspace_bytes(arr, len(entries)), "uz_wgrad_multi_workspace_bytes")
    ws = torch.empty(max(ws_bytes, 256) // 4, dtype=torch.float32, device=entries[0][0].buf.device)
    with _Timed("wgrad_multi", flops, nbytes + 2.0 * ws_bytes):
        L.check(lib.uz_wgrad_multi(arr, len(entries), ws.data_ptr(), L.stream_ptr()), "uz_wgrad_multi")


def bn_finalize(stats: torch.Tensor, count: int, gamma, beta, eps: float, momentum: float,
                running_mean, running_var):
    lib = L.load()
    C = stats.shape[2]
    dev = stats.device
    vec = torch.empty((4, C), dtype=torch.float32, device=dev)  # scale, shift, mean, invstd
    with _Timed("bn_finalize", 0.0, 4.0 * stats.numel()):
        L.check(lib.uz_bn_finalize(stats.data_ptr(), stats.shape[0], C, float(count), gamma.data_ptr(),
                                   beta.data_ptr(), eps, momentum, _p(running_mean), _p(running_var),
                                   vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(),
                                   vec[3].data_ptr(), L.stream_ptr()), "uz_bn_finalize")
    return vec


def colstats(x: Act) -> torch.Tensor:
    """partial rows (rows, 2, C) of per-channel sum / sum of squares of x, as bn_finalize() reads them"""
    lib = L.load()
    code = L.dtype_code(x.dtype)
    rows = L.check_count(lib.uz_colstats_rows(code, x.P, x.C), "uz_colstats_rows")
    part = torch.empty((rows, 2, x.C), dtype=torch.float32, device=x.buf.device)
    with _Timed("colstats", 0.0, x.buf.element_size() * x.P * x.C):
        L.check(lib.uz_colstats(code, x.ptr(), x.ld, x.P, x.C, part.data_ptr(), L.stream_ptr()), "uz_colstats")
    return part


def add_acts(a: Act, b: Act, out: Act) -> None:
    """out = a + b (the residual sums of ResidualConv, common_layers.py:199)"""
    pool_grad_combine(a, a, b, None, out)


def add_relu(a: Act, b: Optional[Act], out: Act) -> None:
    """out = relu(a + b) (multiresunet.py:79-80, 127-129)"""
    assert (a.P, a.C) == (out.P, out.C) and (b is None or (b.P, b.C) == (a.P, a.C))
    with _Timed("add_relu", 0.0, a.buf.element_size() * a.P * a.C * (2 + (b is not None))):
        L.check(L.load().uz_add_relu(L.dtype_code(a.dtype), a.ptr(), a.ld, b.ptr() if b is not None else None,
                                     b.ld if b is not None else 0, out.ptr(), out.ld, a.P, a.C, L.stream_ptr()), "uz_add_relu")


def relu_bwd(out: Act, g: Act, dx: Act) -> None:
    """dx = g * [out > 0]"""
    assert (out.P, out.C) == (g.P, g.C) == (dx.P, dx.C)
    with _Timed("relu_bwd", 0.0, out.buf.element_size() * out.P * out.C * 3):
        L.check(L.load().uz_relu_bwd(L.dtype_code(out.dtype), out.ptr(), out.ld, g.ptr(), g.ld, dx.ptr(), dx.ld, out.P,
                                     out.C, L.stream_ptr()), "uz_relu_bwd")


def bn_eval_scale(gamma, beta, running_mean, running_var, eps: float):
    lib = L.load()
    C = gamma.numel()
    vec = torch.empty((2, C), dtype=torch.float32, device=gamma.device)
    L.check(lib.uz_bn_eval_scale(C, gamma.data_ptr(), beta.data_ptr(), running_mean.data_ptr(),
                                 running_var.data_ptr(), eps, vec[0].data_ptr(), vec[1].data_ptr(),
                                 L.stream_ptr()), "uz_bn_eval_scale")
    return vec


def bn_relu_apply(y: Act, scale: torch.Tensor, shift: torch.Tensor, act: Act,
                  pooled: Optional[Act] = None, res: Optional[Act] = None, pool_ceil: bool = False,
                  relu: bool = True, reverse: bool = False) -> None:
    """act = relu(scale*y + shift) [+ res]; pooled = maxpool2x2(act) (floor or ceil output size);
    relu=False: plain BatchNorm (no pool).  reverse: the pass walks the tensor from its END -- the part the producing kernel
    wrote last and the 256 MB Infinity Cache still holds (tools/mall_order_probe.py: 46 -> 39 us after a convolution at
    64 channels x 1 M pixels, 65 us on a cold tensor); same values either way"""
    if pooled is not None:
        want = ((y.H + 1) // 2, (y.W + 1) // 2) if pool_ceil else (y.H // 2, y.W // 2)
        assert (pooled.H, pooled.W) == want, (pooled.H, pooled.W, want)
    lib = L.load()
    es = y.buf.element_size()
    with _Timed("bn_relu_apply", 0.0, es * y.P * y.C * ((2.25 if pooled is not None else 2.0) + (res is not None))):
        L.check(lib.uz_bn_relu_add_apply(L.dtype_code(y.dtype), y.ptr(), y.ld, scale.data_ptr(),
                                         shift.data_ptr(), y.N, y.H, y.W, y.C,
                                         res.ptr() if res is not None else None,
                                         res.ld if res is not None else 0, act.ptr(), act.ld,
                                         pooled.ptr() if pooled is not None else None,
                                         pooled.ld if pooled is not None else 0,
                                         int(pool_ceil) | (0 if relu else 2) | (4 if reverse else 0), L.stream_ptr()),
                "uz_bn_relu_add_apply")


def bn_relu_apply_fin(y: Act, stats: torch.Tensor, count: int, gamma, beta, eps: float, momentum: float, running_mean,
                      running_var, flag: torch.Tensor, act: Act, pooled: Optional[Act] = None, res: Optional[Act] = None,
                      pool_ceil: bool = False, relu: bool = True) -> torch.Tensor:
    """bn_finalize + bn_relu_apply in one launch (uz_bn_relu_add_apply_fin): returns vec = (scale, shift, mean, invstd).
    flag: one zeroed int32 element of the caller's per-step flag arena."""
    if pooled is not None:
        want = ((y.H + 1) // 2, (y.W + 1) // 2) if pool_ceil else (y.H // 2, y.W // 2)
        assert (pooled.H, pooled.W) == want, (pooled.H, pooled.W, want)
    C = stats.shape[2]
    assert C == y.C and flag.dtype == torch.int32 and flag.numel() == 1
    vec = torch.empty((4, C), dtype=torch.float32, device=stats.device)
    es = y.buf.element_size()
    with _Timed("bn_relu_apply", 0.0, es * y.P * y.C * ((2.25 if pooled is not None else 2.0) + (res is not None)) + 4.0 * stats.numel()):
        L.check(L.load().uz_bn_relu_add_apply_fin(
            L.dtype_code(y.dtype), y.ptr(), y.ld, stats.data_ptr(), stats.shape[0], float(count), gamma.data_ptr(), beta.data_ptr(),
            eps, momentum, _p(running_mean), _p(running_var), vec.data_ptr(), flag.data_ptr(), y.N, y.H, y.W, y.C,
            res.ptr() if res is not None else None, res.ld if res is not None else 0, act.ptr(), act.ld,
            pooled.ptr() if pooled is not None else None, pooled.ld if pooled is not None else 0,
            int(pool_ceil) | (0 if relu else 2), L.stream_ptr()), "uz_bn_relu_add_apply_fin")
    return vec


BN_BWD_ALTERNATE = True


def bn_relu_bwd(y: Act, vec: torch.Tensor, g0: Optional[Act], g1: Optional[Act],
                gpool: Optional[Act], sums: torch.Tensor, dy: Act, dgamma: torch.Tensor,
                dbeta: torch.Tensor, pool_ceil: bool = False, relu: bool = True,
                partials: Optional[torch.Tensor] = None, frozen: bool = False,
                fin_flag: Optional[torch.Tensor] = None, reverse: bool = False) -> None:
    """Two-pass backward of BN(train)+ReLU(+pool); `sums` is a float64 (2, C) scratch; relu=False: the
    forward was a plain BatchNorm.  partials: the rows of the first pass as left by the convolution that produced g0
    (conv_igemm(bnred=...)): only their fixed-order sum is launched instead of the reduction pass.
    frozen: the forward used RUNNING statistics (model.eval(); fine-tuning with frozen BatchNorm, which the reference
    allows): vec holds (scale, shift, running_mean, 1/sqrt(running_var + eps)); mean and variance are constants, so
    dy = scale * g * mask without the two batch-correction terms -- the same kernels with the sums zeroed between the
    passes; dgamma = sum g*mask*xhat and dbeta = sum g*mask are the first pass's results as they are."""
    lib = L.load()
    def desc(rev):
        return L.BnBwdDesc(L.dtype_code(y.dtype), y.N, y.H, y.W, y.C, y.ld,
                           g0.ld if g0 is not None else 0, g1.ld if g1 is not None else 0,
                           gpool.ld if gpool is not None else 0, dy.ld,
                           int(pool_ceil) | (0 if relu else 2) | (4 if rev else 0))   # bit 2: walk from the end (see bn_relu_apply)
    d = desc(reverse)
    # the apply pass after a reduce pass of its own starts where that pass ENDED (the opposite walk): what the reduce pass read
    # last is what the Infinity Cache holds (class-level switch for tools/ab_step.py)
    d_apply = desc(reverse != (partials is None and BN_BWD_ALTERNATE))
    args = (y.ptr(), vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(), vec[3].data_ptr(),
            g0.ptr() if g0 is not None else None, g1.ptr() if g1 is not None else None,
            gpool.ptr() if gpool is not None else None)
    s = L.stream_ptr()
    nsrc = (g0 is not None) + (g1 is not None) + 0.25 * (gpool is not None)
    es = y.buf.element_size()
    if fin_flag is not None and not frozen:
        # the finalize rides in the apply pass's launch (uz_bn_relu_bwd_apply_fin): fin_flag is one zeroed int32 of the
        # caller's per-step flag arena
        if partials is not None:
            assert relu and g1 is None and gpool is None and partials.shape[1:] == (2, y.C) and partials.is_contiguous()
            rows, pptr = partials.shape[0], partials.data_ptr()
        else:
            wsb = L.check_count(lib.uz_bn_relu_bwd_workspace_bytes(byref(d), int(gpool is not None)),
                                "uz_bn_relu_bwd_workspace_bytes")
            ws = torch.empty(wsb // 4, dtype=torch.float32, device=y.buf.device)
            rows, pptr = wsb // (8 * y.C), ws.data_ptr()
            with _Timed("bn_relu_bwd_reduce", 0.0, es * y.P * y.C * (1 + nsrc)):
                L.check(lib.uz_bn_relu_bwd_reduce_rows(byref(d), *args, pptr, s), "uz_bn_relu_bwd_reduce_rows")
        with _Timed("bn_relu_bwd_apply", 0.0, es * y.P * y.C * (2 + nsrc)):
            L.check(lib.uz_bn_relu_bwd_apply_fin(byref(d_apply), *args, pptr, rows, sums.data_ptr(), dgamma.data_ptr(),
                                                 dbeta.data_ptr(), fin_flag.data_ptr(), float(y.P), dy.ptr(), s),
                    "uz_bn_relu_bwd_apply_fin")
        return
    if partials is not None:
        assert relu and g1 is None and gpool is None and partials.shape[1:] == (2, y.C) and partials.is_contiguous()
        L.check(lib.uz_bn_bwd_finalize(partials.data_ptr(), partials.shape[0], y.C, sums.data_ptr(),
                                       dgamma.data_ptr(), dbeta.data_ptr(), s), "uz_bn_bwd_finalize")
    else:
        wsb = L.check_count(lib.uz_bn_relu_bwd_workspace_bytes(byref(d), int(gpool is not None)),
                            "uz_bn_relu_bwd_workspace_bytes")
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=y.buf.device)
        with _Timed("bn_relu_bwd_reduce", 0.0, es * y.P * y.C * (1 + nsrc)):
            L.check(lib.uz_bn_relu_bwd_reduce(byref(d), *args, ws.data_ptr(), sums.data_ptr(),
                                              dgamma.data_ptr(), dbeta.data_ptr(), s), "uz_bn_relu_bwd_reduce")
    if frozen:
        sums.zero_()
    with _Timed("bn_relu_bwd_apply", 0.0, es * y.P * y.C * (2 + nsrc)):
        L.check(lib.uz_bn_relu_bwd_apply(byref(d_apply), *args, sums.data_ptr(), float(y.P), dy.ptr(), s),
                "uz_bn_relu_bwd_apply")


def outconv_xform_supported(x: Act, K: int) -> bool:
    """uz_outconv_fwd_xf / uz_outconv_bwd_bnred(x = NULL) take this raw activation (bf16, <= 512 channels, <= 8 classes)"""
    return x.dtype == torch.bfloat16 and x.C % 8 == 0 and x.C // 8 <= 64 and 1 <= K <= 8 and x.ld % 8 == 0


def outconv_fwd(x: Act, w: torch.Tensor, b: torch.Tensor, xform: Optional[tuple] = None) -> torch.Tensor:
    """xform = (scale, shift): x is the RAW output of the convolution in front, read through its BatchNorm + ReLU"""
    lib = L.load()
    K = w.shape[0]
    out = torch.empty((x.N, K, x.H, x.W), dtype=torch.float32, device=x.buf.device)
    if xform is not None:
        assert outconv_xform_supported(x, K)
        L.check(lib.uz_outconv_fwd_xf(L.dtype_code(x.dtype), x.ptr(), x.ld, x.N, x.H * x.W, x.C, xform[0].data_ptr(),
                                      xform[1].data_ptr(), w.data_ptr(), b.data_ptr(), K, out.data_ptr(), L.stream_ptr()),
                "uz_outconv_fwd_xf")
        return out
    L.check(lib.uz_outconv_fwd(L.dtype_code(x.dtype), x.ptr(), x.ld, x.N, x.H * x.W, x.C,
                               w.data_ptr(), b.data_ptr(), K, out.data_ptr(), L.stream_ptr()),
            "uz_outconv_fwd")
    return out


def outconv_bwd(x: Act, w: torch.Tensor, g: torch.Tensor, dx: Optional[Act],
                dw: Optional[torch.Tensor] = None, db: Optional[torch.Tensor] = None,
                bnred: Optional[tuple] = None, lazy: bool = False):
    """bnred = (bn_y, vec): x = relu(bn(bn_y)) is read by this head only; the BatchNorm-backward sums of dx are taken in
    the same pass (uz_outconv_bwd_bnred) and their partial rows left in dx.bn_partials (bf16 only; else ignored).
    lazy: x was never written down (outconv_fwd(xform=...)): x is only a shape here, the kernel forms the activation from
    bn_y (bnred and dx are then required)"""
    lib = L.load()
    K = w.shape[0]
    assert g.dtype == torch.float32 and g.is_contiguous() and g.shape == (x.N, K, x.H, x.W)
    dev = x.buf.device
    if dw is None:
        dw = torch.empty((K, x.C), dtype=torch.float32, device=dev)
    if db is None:
        db = torch.empty(K, dtype=torch.float32, device=dev)
    code = L.dtype_code(x.dtype)
    wsb = L.check_count(lib.uz_outconv_bwd_workspace_bytes(code, x.N, x.H * x.W, x.C, K),
                        "uz_outconv_bwd_workspace_bytes")
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=dev)
    if lazy:
        assert bnred is not None and dx is not None and x.dtype == torch.bfloat16 and (bnred[0].P, bnred[0].C) == (x.P, x.C), \
            "a head on a lazy activation takes its backward through uz_outconv_bwd_bnred"
    if bnred is not None and dx is not None and x.dtype == torch.bfloat16 and (bnred[0].P, bnred[0].C) == (x.P, x.C):
        bn_y, vec4 = bnred
        rows = L.check_count(lib.uz_outconv_bwd_rows(code, x.N, x.H * x.W, x.C), "uz_outconv_bwd_rows")
        part = torch.empty((rows, 2, x.C), dtype=torch.float32, device=dev)
        L.check(lib.uz_outconv_bwd_bnred(code, None if lazy else x.ptr(), x.ld, x.N, x.H * x.W, x.C, w.data_ptr(), K, g.data_ptr(),
                                         dx.ptr(), dx.ld, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), bn_y.ptr(), bn_y.ld,
                                         vec4[0].data_ptr(), vec4[1].data_ptr(), vec4[2].data_ptr(), vec4[3].data_ptr(),
                                         part.data_ptr(), L.stream_ptr()), "uz_outconv_bwd_bnred")
        dx.bn_partials = part
        return dw.view(K, x.C), db
    L.check(lib.uz_outconv_bwd(code, x.ptr(), x.ld, x.N, x.H * x.W, x.C, w.data_ptr(), K, g.data_ptr(),
                               dx.ptr() if dx is not None else None,
                               dx.ld if dx is not None else 0, dw.data_ptr(), db.data_ptr(),
                               ws.data_ptr(), L.stream_ptr()), "uz_outconv_bwd")
    return dw.view(K, x.C), db


def colsum(x: Act, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """per-channel sums over the pixels / tokens (fp32), deterministic two-stage reduction"""
    lib = L.load()
    code = L.dtype_code(x.dtype)
    if out is None:
        out = torch.empty(x.C, dtype=torch.float32, device=x.buf.device)
    assert out.numel() == x.C and out.dtype == torch.float32 and out.is_contiguous()
    wsb = L.check_count(lib.uz_colsum_workspace_bytes(code, x.P, x.C), "uz_colsum_workspace_bytes")
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=x.buf.device)
    with _Timed("colsum", 0.0, x.buf.element_size() * x.P * x.C):
        L.check(lib.uz_colsum_ws(code, x.ptr(), x.ld, x.P, x.C, out.data_ptr(), ws.data_ptr(), L.stream_ptr()),
                "uz_colsum_ws")
    return out


def colsum_batched(pairs: Sequence) -> None:
    """out[c] = sum over the tokens of x[:, c] for every (x: Act, out: fp32 tensor of x.C) pair, all tensors of one
    run dtype per call group: two launches per 80 tensors (uz_colsum_batched)"""
    by_dtype = {}
    for x, out in pairs:
        assert out.numel() == x.C and out.dtype == torch.float32 and out.is_contiguous()
        by_dtype.setdefault(x.dtype, []).append((x, out))
    lib = L.load()
    for dt, group in by_dtype.items():
        arr = (L.ColsumItem * len(group))()
        nbytes = 0.0
        for i, (x, out) in enumerate(group):
            arr[i] = L.ColsumItem(x.ptr(), out.data_ptr(), x.P, x.C, x.ld, 0)
            nbytes += x.buf.element_size() * x.P * x.C
        code = L.dtype_code(dt)
        wsb = L.check_count(lib.uz_colsum_batched_workspace_bytes(code, arr, len(group)), "uz_colsum_batched_workspace_bytes")
        ws = torch.empty(max(wsb // 4, 1), dtype=torch.float32, device=group[0][0].buf.device)
        with _Timed("colsum_batched", 0.0, nbytes):
            L.check(lib.uz_colsum_batched(code, arr, len(group), ws.data_ptr(), L.stream_ptr()), "uz_colsum_batched")


# ------------------------------------------------------------------------------------------------
# attention gate (AttentionBlock, attention_unet.py:34-40) and nearest-upsample backward
def sum_rows(partial: torch.Tensor, rows: int, n: int) -> torch.Tensor:
    out = torch.empty(n, dtype=torch.float64, device=partial.device)
    L.check(L.load().uz_sum_rows(partial.data_ptr(), rows, n, out.data_ptr(), L.stream_ptr()), "uz_sum_rows")
    return out


def sum_rows_f32_batched(entries: Sequence) -> None:
    """sum_rows_f32 for every (partial, rows, out0, out1) entry in one launch per 80 buffers"""
    if not entries:
        return
    arr = (L.SumRowsItem * len(entries))()
    for i, (partial, rows, out0, out1) in enumerate(entries):
        n0 = out0.numel()
        n = n0 + (out1.numel() if out1 is not None else 0)
        assert partial.numel() == rows * n and partial.dtype == torch.float32
        arr[i] = L.SumRowsItem(partial.data_ptr(), out0.data_ptr(), out1.data_ptr() if out1 is not None else None, rows, n, n0, 0)
    with _Timed("sum_rows_batched", 0.0, 4.0 * sum(e[0].numel() for e in entries)):
        L.check(L.load().uz_sum_rows_f32_batched(arr, len(entries), L.stream_ptr()), "uz_sum_rows_f32_batched")


def sum_rows_f32(partial: torch.Tensor, rows: int, out0: torch.Tensor, out1: Optional[torch.Tensor] = None,
                 defer: Optional[list] = None) -> None:
    """fp32 row sums written in place: the first out0.numel() elements to out0, the rest to out1.
    defer: a list -> nothing runs now; the entry is appended for sum_rows_f32_batched()"""
    if defer is not None:
        defer.append((partial, rows, out0, out1))
        return
    n0 = out0.numel()
    n = n0 + (out1.numel() if out1 is not None else 0)
    assert partial.numel() == rows * n and partial.dtype == torch.float32
    for o in (out0, out1):
        assert o is None or (o.dtype == torch.float32 and o.is_contiguous())
    L.check(L.load().uz_sum_rows_f32(partial.data_ptr(), rows, n, out0.data_ptr(), n0,
                                     out1.data_ptr() if out1 is not None else None, L.stream_ptr()), "uz_sum_rows_f32")


def attn_grid(dtype: torch.dtype, P: int, channels: int) -> int:
    return L.check_count(L.load().uz_attn_grid(L.dtype_code(dtype), P, channels), "uz_attn_grid")


def attn_psi_fwd(g1: Act, x1: Act, vec_g, vec_x, wpsi: torch.Tensor, bpsi: Optional[torch.Tensor]):
    """q[p] = b + sum_c relu(bn(g1)+bn(x1))*w ; returns (q fp32 [P], partial [G,2,1])"""
    dev = g1.buf.device
    G = attn_grid(g1.dtype, g1.P, g1.C)
    q = torch.empty(g1.P, dtype=torch.float32, device=dev)
    part = torch.empty((G, 2, 1), dtype=torch.float32, device=dev)
    with _Timed("attn_psi_fwd", 0.0, g1.buf.element_size() * 2.0 * g1.P * g1.C):
        L.check(L.load().uz_attn_psi_fwd(L.dtype_code(g1.dtype), g1.ptr(), g1.ld, x1.ptr(), x1.ld,
                                         vec_g.data_ptr(), vec_x.data_ptr(), wpsi.data_ptr(), _p(bpsi),
                                         g1.P, g1.C, q.data_ptr(), part.data_ptr(), L.stream_ptr()),
                "uz_attn_psi_fwd")
    return q, part


def attn_gate_fwd(x: Act, q: torch.Tensor, vec_q: torch.Tensor, out: Act) -> None:
    with _Timed("attn_gate_fwd", 0.0, x.buf.element_size() * 2.0 * x.P * x.C):
        L.check(L.load().uz_attn_gate_fwd(L.dtype_code(x.dtype), x.ptr(), x.ld, q.data_ptr(), vec_q.data_ptr(),
                                          x.P, x.C, out.ptr(), out.ld, L.stream_ptr()), "uz_attn_gate_fwd")


def attn_bwd_psi(dout: Act, x: Act, q: torch.Tensor, vec_q: torch.Tensor, dxd: Act):
    dev = x.buf.device
    G = attn_grid(x.dtype, x.P, x.C)
    dz = torch.empty(x.P, dtype=torch.float32, device=dev)
    part = torch.empty((G, 2), dtype=torch.float32, device=dev)
    with _Timed("attn_bwd_psi", 0.0, x.buf.element_size() * 3.0 * x.P * x.C):
        L.check(L.load().uz_attn_bwd_psi(L.dtype_code(x.dtype), dout.ptr(), dout.ld, x.ptr(), x.ld, q.data_ptr(),
                                         vec_q.data_ptr(), x.P, x.C, dxd.ptr(), dxd.ld, dz.data_ptr(),
                                         part.data_ptr(), L.stream_ptr()), "uz_attn_bwd_psi")
    return dz, sum_rows(part, G, 2)


def attn_bwd_branches(g1: Act, x1: Act, q, dz, wpsi, vec_g, vec_x, vec_q, a01, dg1: Act, dx1: Act,
                      frozen: bool = False) -> torch.Tensor:
    """reduce + apply passes; returns the float64 totals [4F+1] = (B0, B1, D1, dw_psi, sum dq).
    frozen: the three BatchNorms run on their running statistics (model.eval() + backward()): the batch-statistics terms
    of their input gradients vanish -- the kernels receive zeros for the sums they would subtract -- while the sums
    themselves still are the gradients of gamma / beta and are returned unchanged"""
    lib = L.load()
    F_, P = g1.C, g1.P
    G = attn_grid(g1.dtype, P, F_)
    part = torch.empty((G, 4 * F_ + 1), dtype=torch.float32, device=g1.buf.device)
    code = L.dtype_code(g1.dtype)
    a01_k = torch.zeros_like(a01) if frozen else a01
    common = (code, g1.ptr(), g1.ld, x1.ptr(), x1.ld, q.data_ptr(), dz.data_ptr(), wpsi.data_ptr(),
              vec_g.data_ptr(), vec_x.data_ptr(), vec_q.data_ptr(), a01_k.data_ptr())
    es = g1.buf.element_size()
    with _Timed("attn_bwd_reduce", 0.0, es * 2.0 * P * F_):
        L.check(lib.uz_attn_bwd_reduce(*common, P, F_, part.data_ptr(), L.stream_ptr()), "uz_attn_bwd_reduce")
    tot = sum_rows(part, G, 4 * F_ + 1)
    tot_k = tot
    if frozen:
        tot_k = tot.clone()
        tot_k[:3 * F_].zero_()
    with _Timed("attn_bwd_apply", 0.0, es * 4.0 * P * F_):
        L.check(lib.uz_attn_bwd_apply(*common, tot_k.data_ptr(), P, F_, dg1.ptr(), dg1.ld, dx1.ptr(), dx1.ld,
                                      L.stream_ptr()), "uz_attn_bwd_apply")
    return tot


def sum2x2(du: Act, dx: Act) -> None:
    """backward of nearest x2 upsampling: dx (coarse) = sum of the 2x2 fine pixels of du"""
    with _Timed("sum2x2", 0.0, du.buf.element_size() * 1.25 * du.P * du.C):
        L.check(L.load().uz_sum2x2(L.dtype_code(du.dtype), du.ptr(), du.ld, dx.N, dx.H, dx.W, dx.C, dx.ptr(),
                                   dx.ld, L.stream_ptr()), "uz_sum2x2")


# ------------------------------------------------------------------------------------------------
# U^2-Net pieces (u2net.py): bilinear resize, residual/pool gradient merge, side heads, fuse conv
def bilinear_fwd(x: Act, out: Act, align_corners: bool = False) -> None:
    """out = F.interpolate(x, size=(out.H, out.W), mode='bilinear', align_corners=align_corners)"""
    assert x.N == out.N and x.C == out.C and x.dtype == out.dtype
    es = x.buf.element_size()
    with _Timed("bilinear_fwd", 0.0, es * (x.P + out.P) * x.C):
        L.check(L.load().uz_resize_bilinear_fwd(L.dtype_code(x.dtype), x.ptr(), x.ld, x.H * x.W * x.ld, x.N, x.H, x.W,
                                                x.C, out.ptr(), out.ld, out.H * out.W * out.ld, out.H, out.W,
                                                int(align_corners), L.stream_ptr()), "uz_resize_bilinear_fwd")


def bilinear_bwd(g: Act, dx: Act, align_corners: bool = False) -> None:
    """dx (at the resize's input resolution) from g (at its output resolution)"""
    assert g.N == dx.N and g.C == dx.C and g.dtype == dx.dtype
    es = g.buf.element_size()
    with _Timed("bilinear_bwd", 0.0, es * (g.P + dx.P) * g.C):
        L.check(L.load().uz_resize_bilinear_bwd(L.dtype_code(g.dtype), g.ptr(), g.ld, g.H * g.W * g.ld, g.N, dx.H, dx.W,
                                                g.C, dx.ptr(), dx.ld, dx.H * dx.W * dx.ld, g.H, g.W,
                                                int(align_corners), L.stream_ptr()), "uz_resize_bilinear_bwd")


RESAMPLE_COPY, RESAMPLE_SUBSAMPLE, RESAMPLE_ZERO_INSERT = 0, 1, 2


def resample2(src: Act, dst: Act, mode: int) -> None:
    """copy / keep every second pixel / spread between zeros (uz_resample2)"""
    assert src.N == dst.N and src.C == dst.C and src.dtype == dst.dtype
    es = src.buf.element_size()
    with _Timed("resample2", 0.0, es * (min(src.P, dst.P) + dst.P) * src.C):
        L.check(L.load().uz_resample2(L.dtype_code(src.dtype), src.ptr(), src.ld, src.N, src.H, src.W, src.C, dst.ptr(),
                                      dst.ld, dst.H, dst.W, mode, L.stream_ptr()), "uz_resample2")


def bilinear_planes(src_ptr: int, src_img: int, hi: int, wi: int, dst_ptr: int, dst_img: int, ho: int, wo: int,
                    n: int, backward: bool = False) -> None:
    """fp32 single-channel planes (the logit maps); strides in elements.  backward: src is the
    gradient at (ho, wo), dst the gradient at (hi, wi)."""
    lib = L.load()
    fn = lib.uz_bilinear_bwd if backward else lib.uz_bilinear_fwd
    with _Timed("bilinear_planes_bwd" if backward else "bilinear_planes_fwd", 0.0, 4.0 * n * (hi * wi + ho * wo)):
        L.check(fn(L.UZ_F32, src_ptr, 1, src_img, n, hi, wi, 1, dst_ptr, 1, dst_img, ho, wo, L.stream_ptr()),
                "uz_bilinear(planes)")


def pool_grad_combine(act: Act, g0: Optional[Act], g1: Optional[Act], gp: Optional[Act], out: Act,
                      pool_ceil: bool = False) -> None:
    if g0 is None:
        g0, g1 = g1, None
    es = act.buf.element_size()
    n = 1 + (g0 is not None) + (g1 is not None) + (1.25 if gp is not None else 0)
    with _Timed("pool_grad_combine", 0.0, es * act.P * act.C * n):
        L.check(L.load().uz_pool_grad_combine(
            L.dtype_code(act.dtype), act.N, act.H, act.W, act.C, act.ptr(), act.ld,
            g0.ptr() if g0 is not None else None, g0.ld if g0 is not None else 0,
            g1.ptr() if g1 is not None else None, g1.ld if g1 is not None else 0,
            gp.ptr() if gp is not None else None, gp.ld if gp is not None else 0,
            out.ptr(), out.ld, int(pool_ceil), L.stream_ptr()), "uz_pool_grad_combine")


def sideconv_fwd(x: Act, w_ptr: int, b_ptr: Optional[int], taps_ws: torch.Tensor, out_ptr: int, out_img: int) -> None:
    assert taps_ws.numel() >= x.N * 9 * x.H * x.W and taps_ws.dtype == torch.float32
    with _Timed("sideconv3x3_fwd", 18.0 * x.P * x.C, x.buf.element_size() * x.P * x.C + 4.0 * 19 * x.P):
        L.check(L.load().uz_sideconv3x3_fwd(L.dtype_code(x.dtype), x.ptr(), x.ld, x.N, x.H, x.W, x.C, w_ptr, b_ptr,
                                            taps_ws.data_ptr(), out_ptr, out_img, L.stream_ptr()),
                "uz_sideconv3x3_fwd")


def sideconv_bwd(x: Act, w_ptr: int, g_ptr: int, g_img: int, dx: Optional[Act], dw_ptr: int,
                 db_ptr: Optional[int]) -> None:
    lib = L.load()
    code = L.dtype_code(x.dtype)
    wsb = L.check_count(lib.uz_sideconv3x3_bwd_workspace_bytes(code, x.N, x.H, x.W, x.C),
                        "uz_sideconv3x3_bwd_workspace_bytes")
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=x.buf.device)
    with _Timed("sideconv3x3_bwd", 36.0 * x.P * x.C, x.buf.element_size() * 2.0 * x.P * x.C):
        L.check(lib.uz_sideconv3x3_bwd(code, x.ptr(), x.ld, x.N, x.H, x.W, x.C, w_ptr, g_ptr, g_img,
                                       dx.ptr() if dx is not None else None, dx.ld if dx is not None else 0,
                                       dw_ptr, db_ptr, ws.data_ptr(), L.stream_ptr()), "uz_sideconv3x3_bwd")


def fuse1x1_fwd(d: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    N, Cc, H, W = d.shape
    K = w.shape[0]
    out = torch.empty((N, K, H, W), dtype=torch.float32, device=d.device)
    L.check(L.load().uz_fuse1x1_fwd(d.data_ptr(), N, H * W, Cc, K, w.data_ptr(), _p(b), out.data_ptr(),
                                    L.stream_ptr()), "uz_fuse1x1_fwd")
    return out


def fuse1x1_bwd(d: torch.Tensor, w: torch.Tensor, g: Optional[torch.Tensor], extras: Sequence[Optional[torch.Tensor]],
                dw: torch.Tensor, db: Optional[torch.Tensor]) -> torch.Tensor:
    import ctypes
    lib = L.load()
    N, Cc, H, W = d.shape
    K = w.shape[0]
    assert len(extras) == Cc // K
    dcat = torch.empty_like(d)
    wsb = L.check_count(lib.uz_fuse1x1_bwd_workspace_bytes(N, H * W, Cc, K), "uz_fuse1x1_bwd_workspace_bytes")
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=d.device)
    arr = (ctypes.c_void_p * len(extras))(*[(e.data_ptr() if e is not None else None) for e in extras])
    L.check(lib.uz_fuse1x1_bwd(d.data_ptr(), N, H * W, Cc, K, w.data_ptr(), _p(g), arr, len(extras),
                               dcat.data_ptr(), dw.data_ptr(), _p(db), ws.data_ptr(), L.stream_ptr()),
            "uz_fuse1x1_bwd")
    return dcat


# ------------------------------------------------------------------------------------------------
# Swin-UNet V2 pieces (swin_unet_v2.py): patch extraction, LayerNorm with folded permutations, window attention
def patchify(x: torch.Tensor, patch: int, kpad: int, dtype: torch.dtype) -> Act:
    L.require_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
    N, C, H, W = x.shape
    out = new_act(N, H // patch, W // patch, kpad, dtype, x.device, needs_grad=False)
    L.check(L.load().uz_patchify(L.dtype_code(dtype), x.data_ptr(), N, C, H, W, patch, kpad, out.buf.data_ptr(),
                                 L.stream_ptr()), "uz_patchify")
    return out


def _ln_desc(x: Act, N: int, Ho: int, Wo: int, C: int, mode: int, r: int, eps: float, ldy=0, ldr=0, ldg=0, lddx=0, act=0):
    return L.LnDesc(L.dtype_code(x.dtype), N, Ho, Wo, C, x.ld, ldy, ldr, ldg, lddx, mode, r, eps, act)


def layernorm_fwd(x: Act, gamma: torch.Tensor, beta: torch.Tensor, out: Act, *, mode: int = L.LN_PLAIN, r: int = 1,
                  eps: float = 1e-5, res: Optional[Act] = None, image_scale: Optional[torch.Tensor] = None,
                  gelu: bool = False) -> torch.Tensor:
    """out = [res +] [image_scale[b] *] LayerNorm(x rows addressed by `mode`), or GELU(LayerNorm(x)); returns the
    (P, 2) mean/rstd"""
    d = _ln_desc(x, out.N, out.H, out.W, out.C, mode, r, eps, ldy=out.ld, ldr=res.ld if res is not None else 0,
                 act=1 if gelu else 0)
    stats = torch.empty((out.P, 2), dtype=torch.float32, device=x.buf.device)
    es = x.buf.element_size()
    with _Timed("layernorm_fwd", 0.0, es * out.P * out.C * (2 + (res is not None))):
        L.check(L.load().uz_layernorm_fwd(byref(d), x.ptr(), gamma.data_ptr(), beta.data_ptr(),
                                          res.ptr() if res is not None else None, _p(image_scale), out.ptr(),
                                          stats.data_ptr(), L.stream_ptr()), "uz_layernorm_fwd")
    return stats


def layernorm_bwd(x: Act, gamma: torch.Tensor, stats: torch.Tensor, g: Act, dx: Act, *, mode: int = L.LN_PLAIN,
                  r: int = 1, eps: float = 1e-5, image_scale: Optional[torch.Tensor] = None,
                  dgamma: Optional[torch.Tensor] = None, dbeta: Optional[torch.Tensor] = None,
                  gelu_beta: Optional[torch.Tensor] = None, defer: Optional[list] = None):
    """returns (dgamma, dbeta) fp32 (written into the given tensors when passed); dx is written with x's addressing.
    gelu_beta: the forward was GELU(LayerNorm(x)) with this beta"""
    lib = L.load()
    d = _ln_desc(x, g.N, g.H, g.W, g.C, mode, r, eps, ldg=g.ld, lddx=dx.ld, act=0 if gelu_beta is None else 1)
    rows = L.check_count(lib.uz_layernorm_bwd_rows(byref(d)), "uz_layernorm_bwd_rows")
    part = torch.empty((rows, 2, g.C), dtype=torch.float32, device=x.buf.device)
    es = x.buf.element_size()
    with _Timed("layernorm_bwd", 0.0, es * g.P * g.C * 3):
        if gelu_beta is not None:
            assert image_scale is None
            L.check(lib.uz_layernorm_act_bwd(byref(d), x.ptr(), gamma.data_ptr(), gelu_beta.data_ptr(), stats.data_ptr(),
                                             g.ptr(), dx.ptr(), part.data_ptr(), L.stream_ptr()), "uz_layernorm_act_bwd")
        else:
            L.check(lib.uz_layernorm_bwd(byref(d), x.ptr(), gamma.data_ptr(), stats.data_ptr(), g.ptr(), _p(image_scale),
                                         dx.ptr(), part.data_ptr(), L.stream_ptr()), "uz_layernorm_bwd")
    if dgamma is None:
        dgamma = torch.empty(g.C, dtype=torch.float32, device=x.buf.device)
    if dbeta is None:
        dbeta = torch.empty(g.C, dtype=torch.float32, device=x.buf.device)
    sum_rows_f32(part, rows, dgamma, dbeta, defer=defer)   # deferred: valid after sum_rows_f32_batched(defer)
    return dgamma, dbeta


LN_HEAD_MAX_CLASSES = 4


def ln_head_supported(C: int, K: int, dtype: torch.dtype) -> bool:
    """shapes the fused LayerNorm + 1x1 head kernels take (<= 4 classes; one class: three 16-byte chunks per lane)"""
    vec = 8 if dtype == torch.bfloat16 else 4
    return C % vec == 0 and 1 <= K <= LN_HEAD_MAX_CLASSES and C // vec <= (192 if K == 1 else 64)


def ln_head_fwd(x: Act, gamma: torch.Tensor, beta: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor],
                N: int, Ho: int, Wo: int, C: int, *, mode: int = L.LN_PLAIN, r: int = 1, eps: float = 1e-5):
    """logits (N, K, Ho, Wo) fp32 = b + w . LayerNorm(x rows addressed by `mode`), and the (P, 2) mean/rstd"""
    K = w.shape[0]
    assert w.shape == (K, C) and w.is_contiguous() and w.dtype == torch.float32
    d = _ln_desc(x, N, Ho, Wo, C, mode, r, eps)
    dev = x.buf.device
    logits = torch.empty((N, K, Ho, Wo), dtype=torch.float32, device=dev)
    stats = torch.empty((N * Ho * Wo, 2), dtype=torch.float32, device=dev)
    with _Timed("ln_head_fwd", 2.0 * N * Ho * Wo * C * K, x.buf.element_size() * N * Ho * Wo * C):
        L.check(L.load().uz_ln_head_fwd(byref(d), x.ptr(), gamma.data_ptr(), beta.data_ptr(), w.data_ptr(), _p(b), K,
                                        logits.data_ptr(), stats.data_ptr(), L.stream_ptr()), "uz_ln_head_fwd")
    return logits, stats


def ln_head_bwd(x: Act, gamma: torch.Tensor, beta: torch.Tensor, w: torch.Tensor, stats: torch.Tensor,
                dlogits: torch.Tensor, dx: Act, *, mode: int = L.LN_PLAIN, r: int = 1, eps: float = 1e-5,
                dgamma=None, dbeta=None, dw=None, db=None):
    """dx (x's addressing) and (dgamma, dbeta, dw (K, C), db (K)) fp32, written into the given tensors when passed"""
    lib = L.load()
    N, K, Ho, Wo = dlogits.shape
    C = w.shape[1]
    assert dlogits.dtype == torch.float32 and dlogits.is_contiguous() and w.shape[0] == K
    d = _ln_desc(x, N, Ho, Wo, C, mode, r, eps, lddx=dx.ld)
    wsb = L.check_count(lib.uz_ln_head_bwd_workspace_bytes(byref(d), K), "uz_ln_head_bwd_workspace_bytes")
    dev = x.buf.device
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=dev)
    outs = []
    for t, shape in ((dgamma, (C,)), (dbeta, (C,)), (dw, (K, C)), (db, (K,))):
        if t is None:
            t = torch.empty(shape, dtype=torch.float32, device=dev)
        assert t.numel() == math.prod(shape) and t.dtype == torch.float32 and t.is_contiguous()
        outs.append(t)
    with _Timed("ln_head_bwd", 4.0 * N * Ho * Wo * C * K, x.buf.element_size() * N * Ho * Wo * C * 2):
        L.check(lib.uz_ln_head_bwd(byref(d), x.ptr(), gamma.data_ptr(), beta.data_ptr(), w.data_ptr(), K,
                                   stats.data_ptr(), dlogits.data_ptr(), dx.ptr(), outs[0].data_ptr(), outs[1].data_ptr(),
                                   outs[2].data_ptr(), outs[3].data_ptr(), ws.data_ptr(), L.stream_ptr()),
                "uz_ln_head_bwd")
    return tuple(outs)


def _attn_desc(qkv: Act, heads: int, ws: int, shift: int, Nt: int, ldo: int, scale: Optional[float] = None):
    C = qkv.C // 3
    return L.WinAttnDesc(L.dtype_code(qkv.dtype), qkv.N, qkv.H, qkv.W, C, heads, ws, shift, Nt, qkv.ld, ldo,
                         float((C // heads) ** -0.5 if scale is None else scale))


def winattn_fwd(qkv: Act, tau: torch.Tensor, bias: torch.Tensor, out: Act, heads: int, ws: int, shift: int,
                scale: Optional[float] = None) -> torch.Tensor:
    d = _attn_desc(qkv, heads, ws, shift, tau.shape[1], out.ld, scale)
    nwin = qkv.N * (qkv.H // ws) * (qkv.W // ws)
    lse = torch.empty((nwin, heads, ws * ws), dtype=torch.float32, device=qkv.buf.device)
    assert tau.dtype == torch.float32 and tau.is_contiguous() and bias.shape == (heads, ws * ws, ws * ws)
    with _Timed("winattn_fwd", 4.0 * qkv.P * ws * ws * out.C, qkv.buf.element_size() * qkv.P * out.C * 4):
        L.check(L.load().uz_winattn_fwd(byref(d), qkv.ptr(), tau.data_ptr(), bias.data_ptr(), out.ptr(), lse.data_ptr(),
                                        L.stream_ptr()), "uz_winattn_fwd")
    return lse


def winattn_bwd(qkv: Act, tau: torch.Tensor, bias: torch.Tensor, out: Act, lse: torch.Tensor, dout: Act, dqkv: Act,
                heads: int, ws: int, shift: int, dtau: Optional[torch.Tensor] = None, scale: Optional[float] = None,
                defer: Optional[list] = None):
    """returns (dbias, dtau) as (heads, N, N) fp32; dtau is written into the given tensor when passed.
    defer: a list -> the row sums that finish dbias / dtau are appended to it instead of launched (sum_rows_f32_batched):
    the two tensors are valid once the caller has flushed the list"""
    lib = L.load()
    d = _attn_desc(qkv, heads, ws, shift, tau.shape[1], out.ld, scale)
    rows = L.check_count(lib.uz_winattn_bwd_rows(byref(d)), "uz_winattn_bwd_rows")
    N = ws * ws
    part = torch.empty((rows, 2, heads, N, N), dtype=torch.float32, device=qkv.buf.device)
    with _Timed("winattn_bwd", 10.0 * qkv.P * N * out.C, qkv.buf.element_size() * qkv.P * out.C * 8):
        L.check(lib.uz_winattn_bwd(byref(d), qkv.ptr(), tau.data_ptr(), bias.data_ptr(), out.ptr(), lse.data_ptr(),
                                   dout.ptr(), dout.ld, dqkv.ptr(), dqkv.ld, part.data_ptr(), L.stream_ptr()),
                "uz_winattn_bwd")
    dbias = torch.empty((heads, N, N), dtype=torch.float32, device=qkv.buf.device)
    if dtau is None:
        dtau = torch.empty((heads, N, N), dtype=torch.float32, device=qkv.buf.device)
    assert dtau.shape == (heads, N, N)
    sum_rows_f32(part, rows, dbias, dtau, defer=defer)
    return dbias, dtau


def cpb_fwd(idx: torch.Tensor, w1, b1, w2, b2) -> torch.Tensor:
    """continuous position bias (heads, R) from the (R, 2) log-spaced offsets"""
    R, heads, hidden = idx.shape[0], w2.shape[0], w1.shape[0]
    assert idx.is_contiguous() and idx.dtype == torch.float32
    bias = torch.empty((heads, R), dtype=torch.float32, device=idx.device)
    L.check(L.load().uz_cpb_fwd(idx.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), R, hidden,
                                heads, bias.data_ptr(), L.stream_ptr()), "uz_cpb_fwd")
    return bias


def cpb_bwd(idx: torch.Tensor, w1, b1, w2, G: torch.Tensor, dw1, db1, dw2, db2) -> None:
    R, heads, hidden = idx.shape[0], w2.shape[0], w1.shape[0]
    assert G.is_contiguous() and G.shape == (heads, R) and G.dtype == torch.float32
    L.check(L.load().uz_cpb_bwd(idx.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), G.data_ptr(), R, hidden,
                                heads, dw1.data_ptr(), db1.data_ptr(), dw2.data_ptr(), db2.data_ptr(), L.stream_ptr()),
            "uz_cpb_bwd")

def _cpb_items(mods):
    """mods: dicts with idx, w1, b1, w2 and (forward) b2, bias or (backward) G, dw1, db1, dw2, db2 tensors"""
    arr = (L.CpbItem * len(mods))()
    for it, m in zip(arr, mods):
        idx, w1, w2 = m["idx"], m["w1"], m["w2"]
        assert idx.is_contiguous() and idx.dtype == torch.float32 and idx.shape[1] == 2
        it.R, it.hidden, it.heads = idx.shape[0], w1.shape[0], w2.shape[0]
        for name in ("idx", "w1", "b1", "w2", "b2", "bias", "G", "dw1", "db1", "dw2", "db2"):
            t = m.get(name)
            if t is not None:
                assert t.dtype == torch.float32 and t.is_contiguous(), name
                setattr(it, name, t.data_ptr())
    return arr


def cpb_fwd_batched(mods) -> None:
    """every module's bias (heads, R) in one launch; see _cpb_items for the fields"""
    arr = _cpb_items(mods)
    L.check(L.load().uz_cpb_fwd_batched(arr, len(mods), L.stream_ptr()), "uz_cpb_fwd_batched")


def cpb_bwd_batched(mods) -> None:
    lib = L.load()
    arr = _cpb_items(mods)
    wsb = L.check_count(lib.uz_cpb_bwd_batched_workspace_bytes(arr, len(mods)), "uz_cpb_bwd_batched_workspace_bytes")
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=mods[0]["idx"].device)
    L.check(lib.uz_cpb_bwd_batched(arr, len(mods), ws.data_ptr(), L.stream_ptr()), "uz_cpb_bwd_batched")



# ------------------------------------------------------------------------------------------------
# MISSFormer / MiT blocks (uz_mit.hip)
def gelu_fwd(x: Act, y: Act) -> None:
    assert (x.P, x.C) == (y.P, y.C) and x.dtype == y.dtype
    with _Timed("gelu_fwd", 0.0, 2 * x.buf.element_size() * x.P * x.C):
        L.check(L.load().uz_gelu_fwd(L.dtype_code(x.dtype), x.ptr(), x.ld, y.ptr(), y.ld, x.P, x.C, L.stream_ptr()),
                "uz_gelu_fwd")


def gelu_bwd(x: Act, g: Act, dx: Act) -> None:
    assert (x.P, x.C) == (g.P, g.C) == (dx.P, dx.C) and x.dtype == g.dtype == dx.dtype
    with _Timed("gelu_bwd", 0.0, 3 * x.buf.element_size() * x.P * x.C):
        L.check(L.load().uz_gelu_bwd(L.dtype_code(x.dtype), x.ptr(), x.ld, g.ptr(), g.ld, dx.ptr(), dx.ld, x.P, x.C,
                                     L.stream_ptr()), "uz_gelu_bwd")


def dwconv3x3(x: Act, w_taps: torch.Tensor, bias: Optional[torch.Tensor], y: Act, *, skip: bool = False,
              flip: bool = False) -> None:
    """depthwise 3x3 (padding 1): w_taps [9, C] fp32; skip adds x; flip = gradient wrt the input"""
    assert (x.N, x.H, x.W, x.C) == (y.N, y.H, y.W, y.C) and x.dtype == y.dtype
    assert w_taps.shape == (9, x.C) and w_taps.dtype == torch.float32 and w_taps.is_contiguous()
    assert bias is None or (bias.numel() == x.C and bias.dtype == torch.float32)
    with _Timed("dwconv3x3", 18.0 * x.P * x.C, 2 * x.buf.element_size() * x.P * x.C):
        L.check(L.load().uz_dwconv3x3(L.dtype_code(x.dtype), x.ptr(), x.ld, w_taps.data_ptr(),
                                      bias.data_ptr() if bias is not None else None, y.ptr(), y.ld, x.N, x.H, x.W, x.C,
                                      (1 if skip else 0) | (2 if flip else 0), L.stream_ptr()), "uz_dwconv3x3")


def dwconv3x3_wgrad(x: Act, g: Act, defer: Optional[list] = None) -> torch.Tensor:
    """[10, C] fp32: rows 0..8 the tap gradients, row 9 the bias gradient (defer: see sum_rows_f32)"""
    assert (x.N, x.H, x.W, x.C) == (g.N, g.H, g.W, g.C) and x.dtype == g.dtype
    lib = L.load()
    code = L.dtype_code(x.dtype)
    rows = L.check_count(lib.uz_dwconv3x3_wgrad_rows(code, x.N, x.H, x.W, x.C), "uz_dwconv3x3_wgrad_rows")
    part = torch.empty(rows, 10 * x.C, dtype=torch.float32, device=x.buf.device)
    out = torch.empty(10, x.C, dtype=torch.float32, device=x.buf.device)
    with _Timed("dwconv3x3_wgrad", 20.0 * x.P * x.C, 2 * x.buf.element_size() * x.P * x.C):
        L.check(lib.uz_dwconv3x3_wgrad(code, x.ptr(), x.ld, g.ptr(), g.ld, part.data_ptr(), x.N, x.H, x.W, x.C,
                                       L.stream_ptr()), "uz_dwconv3x3_wgrad")
    sum_rows_f32(part, rows, out, defer=defer)
    return out


def space_to_depth(src: Act, dst: Act, r: int, inverse: bool = False) -> None:
    """dst[n, ho, wo, (ty*r+tx)*C + c] = src[n, ho*r+ty, wo*r+tx, c]; inverse: the scatter back"""
    fine, coarse = (dst, src) if inverse else (src, dst)
    assert fine.N == coarse.N and fine.H == coarse.H * r and fine.W == coarse.W * r and coarse.C == r * r * fine.C
    assert src.dtype == dst.dtype
    with _Timed("space_to_depth", 0.0, 2 * src.buf.element_size() * fine.P * fine.C):
        L.check(L.load().uz_space_to_depth(L.dtype_code(src.dtype), src.ptr(), src.ld, dst.ptr(), dst.ld, coarse.N,
                                           coarse.H, coarse.W, fine.C, r, 1 if inverse else 0, L.stream_ptr()),
                "uz_space_to_depth")


def im2col_nchw(x: torch.Tensor, k: int, stride: int, pad: int, kpad: int, dtype: torch.dtype) -> Act:
    L.require_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
    N, C, H, W = x.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = new_act(N, Ho, Wo, kpad, dtype, x.device, needs_grad=False)
    L.check(L.load().uz_im2col_nchw(L.dtype_code(dtype), x.data_ptr(), N, C, H, W, k, stride, pad, kpad,
                                    out.buf.data_ptr(), L.stream_ptr()), "uz_im2col_nchw")
    return out


def _sra_desc(q: Act, kv: Act, B: int, heads: int, kps: int, scale: float, ldo: int) -> "L.SraDesc":
    C = heads * 64
    assert q.C == C and kv.C == 2 * C and q.P % B == 0 and kv.P % B == 0 and q.dtype == kv.dtype
    return L.SraDesc(L.dtype_code(q.dtype), B, q.P // B, kv.P // B, heads, 64, kps, q.ld, kv.ld, kv.ld, ldo, scale)


def sra_fwd(q: Act, kv: Act, out: Act, B: int, heads: int, kps: int, scale: float) -> torch.Tensor:
    """out = softmax(q k^T * scale) v per (image, head); kv = [k | v] rows (see uz_sra_fwd); returns lse"""
    d = _sra_desc(q, kv, B, heads, kps, scale, out.ld)
    assert out.P == q.P and out.C == q.C and out.dtype == q.dtype
    lse = torch.empty(B * heads * d.N, dtype=torch.float32, device=q.buf.device)
    es = q.buf.element_size()
    with _Timed("sra_fwd", 4.0 * B * d.N * d.NK * q.C, es * (2 * q.P * q.C + kv.P * kv.C)):
        L.check(L.load().uz_sra_fwd(byref(d), q.ptr(), kv.ptr(), kv.ptr() + q.C * es, out.ptr(),
                                    lse.data_ptr(), L.stream_ptr()), "uz_sra_fwd")
    return lse


def sra_bwd(q: Act, kv: Act, out: Act, lse: torch.Tensor, go: Act, dq: Act, dkv: Act, B: int, heads: int, kps: int,
            scale: float) -> None:
    d = _sra_desc(q, kv, B, heads, kps, scale, out.ld)
    assert dkv.P == kv.P and dkv.C == kv.C and dkv.ld == kv.C and dkv.off == 0 and dq.P == q.P and dq.C == q.C
    lib = L.load()
    wsb = L.check_count(lib.uz_sra_bwd_workspace_bytes(byref(d)), "uz_sra_bwd_workspace_bytes")
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=q.buf.device)
    es = q.buf.element_size()
    with _Timed("sra_bwd", 10.0 * B * d.N * d.NK * q.C, es * (4 * q.P * q.C + 2 * kv.P * kv.C) + wsb):
        L.check(lib.uz_sra_bwd(byref(d), q.ptr(), kv.ptr(), kv.ptr() + q.C * es, out.ptr(), lse.data_ptr(),
                               go.ptr(), go.ld, dq.ptr(), dq.ld, dkv.ptr(), dkv.ld, ws.data_ptr(), L.stream_ptr()),
                "uz_sra_bwd")


# ------------------------------------------------------------------------------------------------
# Dense token attention (unet_transformer.py:126-137, :190-213; transatt_unet.py:41-49, :91-107): batched NT products on
# the LDS-DMA GEMM, softmax over either axis, F.adaptive_avg_pool2d -- include/unetzoo_hip.h "Dense token attention".
def gemm_nt(dtype: torch.dtype, batch: int, M: int, N: int, K: int, x_ptr: int, ldx: int, xb: int, w_ptr: int, ldw: int,
            wb: int, y_ptr: int, ldy: int, yb: int, *, bias: Optional[torch.Tensor] = None, res_ptr: Optional[int] = None,
            ldres: int = 0, resb: int = 0, batch2: int = 1, xb2: int = 0, wb2: int = 0, yb2: int = 0, resb2: int = 0) -> None:
    """y_b[m][n] = sum_k x_b[m][k] w_b[n][k] (+ bias[n]) (+ res_b[m][n]); strides in elements, 0 = shared operand.
    batch2 > 1: a second batch level, matrix (b, h) at + b * xb + h * xb2 ..."""
    d = L.GemmDesc(L.dtype_code(dtype), batch, M, N, K, ldx, ldw, ldy, ldres, xb, wb, yb, resb, batch2, xb2, wb2, yb2, resb2)
    es = 2 if dtype == torch.bfloat16 else 4
    nm = batch * max(batch2, 1)
    with _Timed(f"gemm_nt_{_tname(dtype)}", 2.0 * nm * M * N * K, es * nm * (M * K + N * K + M * N)):
        L.check(L.load().uz_gemm_nt(byref(d), x_ptr, w_ptr, _p(bias), res_ptr, y_ptr, L.stream_ptr()), "uz_gemm_nt")


def wgrad_heads(Lt: Act, Rt: Act, heads: int) -> torch.Tensor:
    """scores[b][h] = L_{b,h}^T R_{b,h} (fp32, (N, heads, Lt.C / heads, Rt.C / heads)): the heads sit side by side in the
    channels of the two token maps; ONE launch for all (image, head) pairs"""
    lib = L.load()
    B, P, H = Lt.N, Lt.H * Lt.W, heads
    C, KV = Lt.C // H, Rt.C // H
    assert (Rt.N, Rt.H * Rt.W) == (B, P) and Lt.dtype == Rt.dtype and Lt.C == H * C and Rt.C == H * KV
    d = L.WgradDesc(L.dtype_code(Lt.dtype), 1, 1, P, 1, P, C, Lt.ld, KV, Rt.ld, 1, L.TAPS_CONV, 1)
    ws_bytes = L.check_count(lib.uz_wgrad_batched_workspace_bytes(byref(d), B * H), "uz_wgrad_batched_workspace_bytes")
    ws = torch.empty(max(ws_bytes // 4, 1), dtype=torch.float32, device=Lt.buf.device)
    out = torch.empty((B, H, C, KV), dtype=torch.float32, device=Lt.buf.device)
    with _Timed(f"wgrad_batched_{_tname(Lt.dtype)}_1tap", 2.0 * B * H * P * C * KV,
                Lt.buf.element_size() * B * P * (Lt.C + Rt.C) + 4.0 * out.numel()):
        L.check(lib.uz_wgrad_batched2(byref(d), B, H, Lt.ptr(), P * Lt.ld, C, Rt.ptr(), P * Rt.ld, KV, out.data_ptr(), C * KV,
                                      ws.data_ptr(), L.stream_ptr()), "uz_wgrad_batched2")
    return out


def wgrad_batched(Lt: Act, Rt: Act, out: Optional[torch.Tensor] = None, out_off: int = 0, ob: Optional[int] = None) -> torch.Tensor:
    """out[b] = L_b^T R_b (fp32, (N, Lt.C, Rt.C)) for the N images of two token maps: one launch pair for the batch.
    `out` / `out_off` / `ob` (floats): write result b at out.flat[out_off + b * ob] instead (a head's block of a
    (B, heads, C, KV) tensor)"""
    L.require_cuda(Lt.buf, Rt.buf)
    lib = L.load()
    B, P = Lt.N, Lt.H * Lt.W
    assert (Rt.N, Rt.H * Rt.W) == (B, P) and Lt.dtype == Rt.dtype
    d = L.WgradDesc(L.dtype_code(Lt.dtype), 1, 1, P, 1, P, Lt.C, Lt.ld, Rt.C, Rt.ld, 1, L.TAPS_CONV, 1)
    ws_bytes = L.check_count(lib.uz_wgrad_batched_workspace_bytes(byref(d), B), "uz_wgrad_batched_workspace_bytes")
    ws = torch.empty(max(ws_bytes // 4, 1), dtype=torch.float32, device=Lt.buf.device)
    if out is None:
        out = torch.empty((B, Lt.C, Rt.C), dtype=torch.float32, device=Lt.buf.device)
        ob = Lt.C * Rt.C
    assert out.dtype == torch.float32 and out.is_contiguous() and out_off + (B - 1) * ob + Lt.C * Rt.C <= out.numel()
    with _Timed(f"wgrad_batched_{_tname(Lt.dtype)}_1tap", 2.0 * B * P * Lt.C * Rt.C,
                Lt.buf.element_size() * B * P * (Lt.C + Rt.C) + 4.0 * B * Lt.C * Rt.C):
        L.check(lib.uz_wgrad_batched(byref(d), B, Lt.ptr(), P * Lt.ld, Rt.ptr(), P * Rt.ld, out.data_ptr() + 4 * out_off, ob,
                                     ws.data_ptr(), L.stream_ptr()), "uz_wgrad_batched")
    return out


def chanattn_probs_fwd(scores: torch.Tensor, scale: float, eps: float, dtype: torch.dtype):
    """scores fp32 (B, H, C, KV) -> (pcat (B, C, H*KV), pcat_t (B, H*KV, C)) in `dtype`: softmax_kv(InstanceNorm(scale * s)) / H"""
    assert scores.dtype == torch.float32 and scores.is_contiguous() and scores.dim() == 4
    B, H, C, KV = scores.shape
    pcat = torch.empty((B, C, H * KV), dtype=dtype, device=scores.device)
    pcat_t = torch.empty((B, H * KV, C), dtype=dtype, device=scores.device)
    with _Timed("chanattn_probs_fwd", 0.0, 12.0 * scores.numel()):
        L.check(L.load().uz_chanattn_probs_fwd(L.dtype_code(dtype), scores.data_ptr(), B, H, C, KV, scale, eps, pcat.data_ptr(),
                                               pcat_t.data_ptr(), L.stream_ptr()), "uz_chanattn_probs_fwd")
    return pcat, pcat_t


def chanattn_probs_bwd(scores: torch.Tensor, dpc: torch.Tensor, scale: float, eps: float, dtype: torch.dtype):
    """d(loss)/d(pcat) fp32 (B, C, H*KV) -> (ds (B, H, C, KV), ds_t (B, H, KV, C)) in `dtype` = d(loss)/d(scores)"""
    B, H, C, KV = scores.shape
    assert dpc.dtype == torch.float32 and dpc.is_contiguous() and tuple(dpc.shape) == (B, C, H * KV)
    ds = torch.empty((B, H, C, KV), dtype=dtype, device=scores.device)
    ds_t = torch.empty((B, H, KV, C), dtype=dtype, device=scores.device)
    with _Timed("chanattn_probs_bwd", 0.0, 24.0 * scores.numel()):
        L.check(L.load().uz_chanattn_probs_bwd(L.dtype_code(dtype), scores.data_ptr(), dpc.data_ptr(), B, H, C, KV, scale, eps,
                                               ds.data_ptr(), ds_t.data_ptr(), L.stream_ptr()), "uz_chanattn_probs_bwd")
    return ds, ds_t


def softmax_fwd(s: torch.Tensor, axis: int, scale: float) -> None:
    """s (batch, rows, cols) <- softmax(scale * s) over axis 0 (of each matrix: columns sum to one) or 1, in place"""
    assert s.dim() == 3 and s.is_contiguous()
    B, R, C = s.shape
    lib = L.load()
    ws_bytes = L.check_count(lib.uz_softmax_workspace_bytes(B, R, C, axis), "uz_softmax_workspace_bytes")
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=s.device) if ws_bytes else None
    with _Timed(f"softmax_axis{axis}_fwd", 0.0, 3.0 * s.numel() * s.element_size()):
        L.check(lib.uz_softmax_fwd(L.dtype_code(s.dtype), s.data_ptr(), C, R * C, B, R, C, axis, scale, _p(ws), L.stream_ptr()),
                "uz_softmax_fwd")


def softmax_bwd(a: torch.Tensor, g: torch.Tensor, axis: int, scale: float, dot: Optional[torch.Tensor] = None) -> None:
    """g <- a * (g - sum_axis(a * g)) * scale in place; `dot` (batch, cols) fp32 = the sums when the caller has them (axis 0)"""
    assert a.shape == g.shape and a.is_contiguous() and g.is_contiguous() and a.dtype == g.dtype
    B, R, C = a.shape
    given = dot is not None
    if axis == 0 and dot is None:
        dot = torch.empty((B, C), dtype=torch.float32, device=a.device)
    with _Timed(f"softmax_axis{axis}_bwd", 0.0, 3.0 * a.numel() * a.element_size()):
        L.check(L.load().uz_softmax_bwd(L.dtype_code(a.dtype), a.data_ptr(), g.data_ptr(), C, R * C, B, R, C, axis, scale,
                                        _p(dot), 1 if given else 0, L.stream_ptr()), "uz_softmax_bwd")


def adaptive_avgpool_fwd(x: Act, out: Act) -> None:
    with _Timed("adaptive_avgpool_fwd", 0.0, x.buf.element_size() * (x.P * x.C + out.P * out.C)):
        L.check(L.load().uz_adaptive_avgpool_fwd(L.dtype_code(x.dtype), x.ptr(), x.ld, x.N, x.H, x.W, x.C, out.ptr(), out.ld,
                                                 out.H, out.W, L.stream_ptr()), "uz_adaptive_avgpool_fwd")


def adaptive_avgpool_bwd(g: Act, dx: Act, accumulate: bool = False) -> None:
    with _Timed("adaptive_avgpool_bwd", 0.0, g.buf.element_size() * (g.P * g.C + dx.P * dx.C)):
        L.check(L.load().uz_adaptive_avgpool_bwd(L.dtype_code(g.dtype), g.ptr(), g.ld, dx.N, dx.H, dx.W, dx.C, dx.ptr(), dx.ld,
                                                 g.H, g.W, 1 if accumulate else 0, L.stream_ptr()), "uz_adaptive_avgpool_bwd")


def rowdot_f32(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """out[r] = sum_c a[r][c] * b[r][c]; a fp32 (rows, C), b run dtype (rows, C)"""
    assert a.dtype == torch.float32 and a.dim() == 2 and b.dim() == 2 and a.shape == b.shape
    assert a.is_contiguous() and b.is_contiguous()
    out = torch.empty(a.shape[0], dtype=torch.float32, device=a.device)
    with _Timed("rowdot_f32", 2.0 * a.numel(), a.numel() * (4 + b.element_size())):
        L.check(L.load().uz_rowdot_f32(L.dtype_code(b.dtype), a.data_ptr(), a.shape[1], b.data_ptr(), b.shape[1], a.shape[0],
                                       a.shape[1], out.data_ptr(), L.stream_ptr()), "uz_rowdot_f32")
    return out


def cast_rows(src: torch.Tensor, dst: Act, accumulate: bool = False) -> None:
    """dst (run dtype) = src (fp32, (P, C) contiguous), or dst += src"""
    assert src.dtype == torch.float32 and src.is_contiguous() and tuple(src.shape) == (dst.P, dst.C)
    with _Timed("cast_rows", 0.0, src.numel() * (4 + dst.buf.element_size())):
        L.check(L.load().uz_cast_rows(L.dtype_code(dst.dtype), src.data_ptr(), dst.C, dst.ptr(), dst.ld, dst.P, dst.C,
                                      1 if accumulate else 0, L.stream_ptr()), "uz_cast_rows")


def add_map(x: Act, map_f32: torch.Tensor, out: Act) -> None:
    """out = x + map broadcast over the batch; map fp32 (H*W, C)"""
    assert map_f32.dtype == torch.float32 and map_f32.is_contiguous() and tuple(map_f32.shape) == (x.H * x.W, x.C)
    assert (out.P, out.C) == (x.P, x.C)
    with _Timed("add_map", 0.0, 2.0 * x.P * x.C * x.buf.element_size()):
        L.check(L.load().uz_add_map(L.dtype_code(x.dtype), x.ptr(), x.ld, map_f32.data_ptr(), out.ptr(), out.ld, x.P, x.H * x.W,
                                    x.C, L.stream_ptr()), "uz_add_map")


def dropout(x: Act, u: torch.Tensor, p: float, out: Act) -> None:
    """out = x * [u >= p] / (1 - p); u fp32 (P, C) uniform draws (the same call masks the gradient)"""
    assert u.dtype == torch.float32 and u.is_contiguous() and tuple(u.shape) == (x.P, x.C) and (out.P, out.C) == (x.P, x.C)
    with _Timed("dropout", 0.0, x.P * x.C * (4.0 + 2 * x.buf.element_size())):
        L.check(L.load().uz_dropout(L.dtype_code(x.dtype), x.ptr(), x.ld, u.data_ptr(), p, out.ptr(), out.ld, x.P, x.C,
                                    L.stream_ptr()), "uz_dropout")


def chanscale_relu(mode: int, g: Optional[Act], x: Act, s: Optional[torch.Tensor], a: Optional[torch.Tensor], out: Act) -> None:
    """mode 2: out = relu(x * s[n][c]); mode 0: out = g * [x > 0] * x; mode 1: out = g * [x > 0] * s + a[n][c]  (s, a: (N, C) fp32)"""
    for t in (s, a):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == (x.N, x.C))
    with _Timed("chanscale_relu", 0.0, (2.0 if mode == 2 else 3.0) * x.P * x.C * x.buf.element_size()):
        L.check(L.load().uz_chanscale_relu(L.dtype_code(x.dtype), mode, g.ptr() if g is not None else None, g.ld if g is not None else 0,
                                           x.ptr(), x.ld, _p(s), _p(a), x.N, x.H * x.W, x.C, out.ptr(), out.ld, L.stream_ptr()),
                "uz_chanscale_relu")
