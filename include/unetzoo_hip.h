/*
 * unetzoo_hip.h — C ABI of libunetzoo_hip.so, the MI355X (gfx950) kernel library behind the
 * unet_zoo encoder/decoder hot path.
 *
 * The reference (irfanfadhullah/unet_zoo) has no FFI: every primitive below replaces an ATen op
 * that the reference reaches through torch.nn.  Each entry point cites
 * the reference call site it stands in for.  Conventions:
 *   - plain pointers + ints, no torch types, no allocation inside, no exceptions;
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (caller passes torch's current stream);
 *   - activations are NHWC ("pixel-major"): element (pixel p, channel c) of a tensor lives at
 *     base[p * ld + c]; `ld` (elements) lets a tensor be a channel slice of a wider buffer, which
 *     is how skip-concats are never materialised (reference: torch.cat, common_layers.py:115);
 *   - dtype: UZ_F32 = IEEE fp32 storage + exact-fp32 MFMA (parity mode),
 *            UZ_BF16 = bf16 storage + bf16 MFMA with fp32 accumulate (throughput mode);
 *   - return 0 on success, otherwise a negative UZ_E* code or a positive hipError_t;
 *     uz_last_error_string() describes the last failure on the calling thread.
 * All functions are re-entrant.  Process-global mutable state: the thread-local error string and ONE process-wide
 * setting, uz_set_cu_reserve() (an atomic the plans read when they size their grids; set it before planning or capturing).
 */
#ifndef UNETZOO_HIP_H
#define UNETZOO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UZ_ABI_VERSION 1

enum { UZ_F32 = 0, UZ_BF16 = 1 };

enum {
  UZ_OK = 0,
  UZ_EINVAL = -1,   /* bad shape / alignment / unsupported combination */
  UZ_ENOTIMPL = -2
};

/* tap geometry of an implicit-GEMM convolution */
enum {
  UZ_TAPS_CONV = 0,     /* ntaps = 1 (1x1) or 9 (3x3, dilation `dil`, zero padding `dil`) */
  UZ_TAPS_GATHER2X2 = 1, /* ntaps = 4: input pixel (2h+a, 2w+b), tap = 2a+b (ConvTranspose k2s2 dgrad) */
  UZ_TAPS_CONV_UP2 = 2,  /* 3x3 taps on the nearest-neighbour x2 upsampling of the input: the input tensor
                            lives at (H/2, W/2) and pixel (h, w) reads (h>>1, w>>1)
                            (nn.Upsample(scale_factor=2) + Conv2d, common_layers.py:69-72) */
  UZ_TAPS_CONV_S2 = 3    /* ntaps = 9: Conv2d(k3, stride 2, padding 1): output pixel (h, w) reads input pixel
                            (2h + ty - 1, 2w + tx - 1) of the (Hin, Win) grid, H = ceil(Hin / 2); zero outside
                            (ResidualConv, common_layers.py:188).  uz_conv_igemm: forward; uz_wgrad: weight gradient
                            (L = dy at (H, W), R = x at (Hr, Wr) = (Hin, Win)) */
};
enum {
  UZ_STORE_PLAIN = 0,    /* y[p*ldy + n] */
  UZ_STORE_SHUFFLE2X2 = 1 /* n = (2a+b)*Co + co  ->  y[pix(2h+a,2w+b)*ldy + co] (ConvTranspose k2s2 fwd) */
};

int uz_abi_version(void);
const char* uz_last_error_string(void);
/* Measurement hook (bench.py's per-launch profile, tools/): between uz_profile_arm(e0, e1) and uz_profile_disarm() the
 * kernel launches made by THIS thread through the library record the two hipEvent_t handles at the kernels' own begin
 * and end timestamps (first launch: both; later launches: the end event only), so hipEventElapsedTime(e0, e1) is the
 * duration from the first kernel's begin to the last kernel's end -- what a kernel trace reports, without the dispatch
 * latency that events recorded around a launch include.  Not to be armed during stream capture.  uz_profile_disarm()
 * returns the number of launches seen. */
int uz_profile_arm(void* start_event, void* stop_event);
int uz_profile_disarm(void);
/* 1 when the library was built with -DUZ_ABLATE (the measurement build of tools/kbench.py, whose kernels honour the
 * UZ_TUNE / UZ_ATTN_GX / UZ_WG_SPLIT environment switches), 0 for the shipped build, which never reads the
 * environment.  bench.py refuses to time an ablation build. */
int uz_build_ablate(void);
/* sha256 (hex) of the kernel sources the library was built from (csrc/Makefile: $(HASHED), concatenated in that order):
 * lets a test tell a stale build -- the library is a build artefact outside the repository's history. */
const char* uz_source_hash(void);
/* Hold `n` of the 256 CUs back from the library's persistent grids (0 <= n <= 128; default 0): the convolution, GEMM and
 * weight-gradient kernels run one 160 KB workgroup per CU, so a collective launched beside the backward (RCCL all-reduce
 * of a finished gradient span; reference seam unet_zoo/utils/multi_gpu.py:20-31) finds no CU to start on
 * until a kernel boundary.  With a reserve every plan sizes its grid -- and its statistics / slab partition, so call it
 * BEFORE querying *_grid_m / *_workspace_bytes -- for 256 - n CUs.  Process-wide; results stay equal up to the fp32
 * order of the per-workgroup partial sums.  uz_get_cu_reserve() returns the current value. */
/* Measurement: the shader clock this device holds under a dense bf16 MFMA stream (the quantity that sets every
 * MFMA-bound kernel's time and differs between devices -- MI355X_MICROARCH.md "DVFS give-back").  `workgroups` blocks of
 * 512 threads issue 8 * iters v_mfma_f32_16x16x32_bf16 per wave on random operands; block b writes
 * out_pairs[2 b] = shader cycles (s_memtime), out_pairs[2 b + 1] = 100 MHz ticks (s_memrealtime) of its loop
 * (uint64, device memory).  clock [GHz] = cycles / ticks / 10.  bench.py reports the median after a settling run. */
int uz_clock_probe(int iters, void* out_pairs, int workgroups, void* stream);
int uz_set_cu_reserve(int n);
int uz_get_cu_reserve(void);

/* ---------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on the matrix cores.
 *   y[p, n] = bias[n] + sum_{tap, c} x[pix(p, tap), c] * w[n, tap*Cin + c]
 * Replaces: nn.Conv2d k3 p1 (common_layers.py:28,31,47,52,71; u2net.py:10 with dilation),
 *           nn.Conv2d k1 (attention_unet.py:11,18,25), their input-gradients (autograd, a19),
 *           nn.ConvTranspose2d k2 s2 forward and input-gradient (common_layers.py:104).
 * Optionally emits per-channel partial sums (sum, sum of squares) of the stored value for the
 * train-mode BatchNorm that follows (common_layers.py:29,32): stats_partial[g][0][n], [g][1][n]
 * for g < *grid_m as returned by uz_conv_igemm_grid_m(); reduced by uz_bn_finalize().
 * Requirements: Cin % (16/sizeof(T)*8) == 0 (bf16: 64, fp32: 32), K = ntaps*Cin,
 *               x, w, y 16-byte aligned, ldx % (16/sizeof(T)) == 0.
 * ------------------------------------------------------------------------------------------- */
typedef struct uz_conv_desc {
  int dtype;
  int N, H, W;     /* output pixel grid */
  int Hin, Win;    /* input pixel grid (== H, W for UZ_TAPS_CONV; 2H..2H+1, 2W..2W+1 for GATHER2X2) */
  int Cin, ldx;    /* channels per tap, input pixel stride */
  int Nout, ldy;   /* GEMM N (rows of w), output pixel stride */
  int ntaps, taps_mode, dil;
  int store_mode, Co; /* Co: channels per sub-pixel for UZ_STORE_SHUFFLE2X2 (Nout = 4*Co) */
  int Hout, Wout;     /* UZ_STORE_SHUFFLE2X2: pixel grid of the destination, 2H..2H+1 x 2W..2W+1 (0 = 2H, 2W).
                       * The 2H x 2W result lands at its top-left; UpSample_UNet's F.pad of an odd skip size
                       * (common_layers.py:110-113) leaves the last row / column to the caller (zeros). */
} uz_conv_desc;

int uz_conv_igemm_grid_m(const uz_conv_desc* d); /* number of stats partial rows; <0 on error */
/* Name of the kernel family the library's own plan picks for this descriptor (what uz_conv_igemm_ws() will launch;
 * `with_workspace` != 0: called with a workspace), e.g. "conv3x3_pp512_bf16", "conv3x3_pp256_bf16", "conv3x3_direct_bf16_bn128",
 * "conv3x3_direct_bf16_bn64_resident", "conv3x3_res64_bf16", "gemm_dma_bf16", "igemm_f32_128x64_tapsplit"; a "_up2"
 * suffix marks the nearest-upsampled input.  Measurement code (bench.py's per-family roofline) labels launches with
 * it instead of mirroring the plan.  Writes at most cap - 1 characters + NUL, returns the full length, < 0 on error. */
int uz_conv_igemm_kernel_name(const uz_conv_desc* d, int with_workspace, char* buf, int cap);
int uz_conv_igemm(const uz_conv_desc* d, const void* x, const void* w_packed, const float* bias,
                  void* y, float* stats_partial, void* stream);
/* y = conv(x) + bias + res: the same with an (M, ldres) tensor of the run dtype added in the epilogue (the result is
 * rounded to the run dtype before the addition, as a separate add of the stored tensor would): the residual sums
 * x + proj(.), tx + fc2(.) of a transformer block (missformer.py:266-267) and, in the backward, the sum of a tensor's
 * gradients when the last contribution is an input-gradient GEMM.  Problems of the LDS-DMA GEMM (1x1 / Linear, the
 * gather modes) with UZ_STORE_PLAIN; others: UZ_ENOTIMPL. */
int uz_conv_igemm_res(const uz_conv_desc* d, const void* x, const void* w_packed, const float* bias,
                      const void* res, int ldres, void* y, void* stream);
/* The same with the workspace of uz_conv_igemm_workspace_bytes() (may be NULL): products with few tiles and a long K loop
 * (Linear layers on the 8 x 8 / 16 x 16 token maps) then run split over K -- fp32 partial tiles, a fixed-order reduce pass
 * that adds bias and residual -- as uz_conv_igemm_ws does for the form without a residual. */
int uz_conv_igemm_res_ws(const uz_conv_desc* d, const void* x, const void* w_packed, const float* bias,
                         const void* res, int ldres, void* y, void* workspace, void* stream);
/* Same with a scratch buffer: small-M 3x3 problems on the generic kernel (u2net's dilated layers at
 * <= 32x32 maps, u2net.py:196-201) split their nine taps across workgroups into fp32 partial tiles in
 * `workspace` and finish with a fixed-order reduce + bias + statistics pass.  workspace_bytes() == 0:
 * identical to uz_conv_igemm (workspace may be NULL).  The statistics rows of THIS entry point are
 * counted by uz_conv_igemm_ws_grid_m().  The same entry serves the split-K form of the direct bf16 3x3 kernel: the
 * 16 x 16 / 32 x 32 bottleneck maps of a 256 x 256 input have 32 ... 128 tiles of 16 ... 32 channel slabs x nine taps, so
 * with a workspace the slabs of a tile are dealt to 2 ... 8 workgroups (fp32 partial tiles, the same reduce pass;
 * uz_conv_igemm_kernel_name(d, 1) then ends in "_splitk").  The split is sized by the hardware's CU count. */
long long uz_conv_igemm_workspace_bytes(const uz_conv_desc* d);
int uz_conv_igemm_ws_grid_m(const uz_conv_desc* d);
int uz_conv_igemm_ws(const uz_conv_desc* d, const void* x, const void* w_packed, const float* bias,
                     void* y, float* stats_partial, void* workspace, void* stream);
/* Input-gradient convolution with the first pass of the BatchNorm backward fused into its epilogue.
 * y = conv(x, w) is the gradient of an activation a = relu(bn(bn_y)) (common_layers.py:28-33: the Conv -> BN -> ReLU
 * whose output feeds only this convolution's forward).  Instead of the statistics of y the kernel writes, per
 * workgroup row of `partial` ([uz_conv_igemm_grid_m()][2][Nout], the layout of uz_bn_relu_bwd_reduce's
 * workspace), sum(dz) and sum(dz * xhat) over its pixels with dz = y * [scale * bn_y + shift > 0] (y as stored,
 * i.e. rounded to the tensor dtype) and xhat = (bn_y - mean) * invstd: uz_bn_bwd_finalize() then stands in for
 * uz_bn_relu_bwd_reduce() and the activation gradient is not read a second time.  bf16 problems of the direct 3x3
 * kernels whose epilogue is staged through LDS and of the LDS-DMA GEMM with UZ_STORE_PLAIN (the 2x2-gather input
 * gradient of ConvTranspose2d k2 s2, common_layers.py:104: the decoder blocks' last BatchNorm)
 * (uz_conv_igemm_bnred_supported() == 1); others: UZ_ENOTIMPL. */
int uz_conv_igemm_bnred_supported(const uz_conv_desc* d);
int uz_conv_igemm_bnred(const uz_conv_desc* d, const void* x, const void* w_packed, void* y, const void* bn_y,
                        int ld_bny, const float* scale, const float* shift, const float* mean, const float* invstd,
                        float* partial, void* stream);

/* Convolution that reads its input THROUGH the BatchNorm + ReLU in front of it (round 5): x holds the raw output of the
 * preceding convolution, the kernel computes y = conv(a, w) + bias with a = relu(x * in_scale[c] + in_shift[c]) formed
 * inside LDS on the halo patch (fp32 fma, rounded to the tensor dtype exactly as uz_bn_relu_apply would store it; zero
 * padding stays zero), so the normalised activation of a DoubleConv's first half (common_layers.py:28-33: Conv -> BN ->
 * ReLU -> Conv) is never written to or read from HBM.  Statistics rows as uz_conv_igemm (uz_conv_igemm_grid_m() rows).
 * bf16 3x3 problems of the direct ping-pong kernel whose LDS image leaves room for the channel table
 * (uz_conv_igemm_xf_supported() == 1); others: UZ_ENOTIMPL -- the caller then materialises the activation. */
int uz_conv_igemm_xf_supported(const uz_conv_desc* d);
int uz_conv_igemm_xf(const uz_conv_desc* d, const void* x, const float* in_scale, const float* in_shift,
                     const void* w_packed, const float* bias, void* y, float* stats_partial, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Weight gradient (reduction over pixels), fp32 output in the reference's parameter layout.
 *   out[i, j, tap] (+)= sum_p L[p, i] * R[pix(p, tap), j]
 * conv k3/k1:  L = dy (i = c_out), R = x (j = c_in)   -> out is OIHW        (nn.Conv2d.weight)
 * convT k2s2:  L = x  (i = c_in),  R = dy (j = c_out) -> out is (Cin,Cout,2,2) (ConvTranspose2d.weight)
 * Replaces the weight-gradient half of autograd for the call sites above.
 * Two kernels: the pixel range is split over uz_wgrad_split() workgroup slices that each write an
 * fp32 partial slab into `workspace` (uz_wgrad_workspace_bytes() bytes, no initialisation
 * needed); a second kernel adds the slabs in fixed order and transposes into `out`
 * (deterministic, no atomics).
 * ------------------------------------------------------------------------------------------- */
typedef struct uz_wgrad_desc {
  int dtype;
  int N, H, W;     /* pixel grid of L */
  int Hr, Wr;      /* pixel grid of R */
  int Ci, ldl;     /* channels / pixel stride of L */
  int Cj, ldr;     /* channels / pixel stride of R */
  int ntaps, taps_mode, dil;
} uz_wgrad_desc;

int uz_wgrad_split(const uz_wgrad_desc* d);
long long uz_wgrad_workspace_bytes(const uz_wgrad_desc* d); /* <0 on error */
int uz_wgrad(const uz_wgrad_desc* d, const void* L, const void* R, float* out, void* workspace,
             void* stream);
/* The kernel family the library's plan launches for this descriptor, written to buf (labels of per-kernel measurements;
 * e.g. "wgrad9_bf16_128x64_rowwalk": nine taps per workgroup, uz_wgrad9.hip).  Returns the length, <0 on error. */
int uz_wgrad_kernel_name(const uz_wgrad_desc* d, char* buf, int cap);
/* The weight gradient of a convolution whose input was read through the BatchNorm + ReLU in front of it (uz_conv_igemm_xf):
 * R holds the RAW output of the preceding convolution; the kernel's loader waves form relu(R * r_scale[c] + r_shift[c]),
 * rounded to the tensor dtype as uz_bn_relu_apply would store it, inside the LDS ring (zero padding stays zero), so the
 * normalised activation of a DoubleConv's first half (common_layers.py:28-33) is not needed in the backward either.
 * Nine-tap bf16 problems of the row-walk kernel (uz_wgrad_xf_supported() == 1); others: UZ_ENOTIMPL.  phase: 0 = uz_wgrad's
 * two launches, 1 / 2 = as uz_wgrad_phase.  Workspace: uz_wgrad_workspace_bytes(). */
int uz_wgrad_xf_supported(const uz_wgrad_desc* d);
int uz_wgrad_xf(const uz_wgrad_desc* d, const void* L, const void* R, const float* r_scale, const float* r_shift,
                float* out, void* workspace, void* stream, int phase);
/* uz_wgrad in two calls, for per-kernel measurements (bench.py brackets each with its own event pair): phase 1 = the main
 * kernel (partial slabs into the workspace), phase 2 = the fixed-order slab reduction into `out`.  uz_wgrad == 1 then 2. */
int uz_wgrad_phase(const uz_wgrad_desc* d, const void* L, const void* R, float* out, void* workspace, void* stream,
                   int phase);
/* n independent problems, each exactly uz_wgrad(&items[i].desc, L, R, out, ...), issued together: the one-tap (nn.Linear)
 * problems of the bf16 run mode share launches -- one per tile shape and 24 problems, plus one reduction per 64 -- with the
 * pixel split planned for the whole set; every other problem runs through uz_wgrad.  The weight gradients of a model's
 * Linear layers (swin_unet_v2.py:205-240, missformer.py:148-214) are not needed before the optimizer step, so the host
 * defers them to the end of the backward pass.  Deterministic for a given set of items: each problem's pixel split -- and so
 * the order of its fp32 additions -- is a function of the set, not of timing.  A problem that is not split writes straight
 * into `out`.  Workspace: uz_wgrad_multi_workspace_bytes(), 256-byte aligned. */
typedef struct {
  uz_wgrad_desc desc;
  const void* L;
  const void* R;
  float* out;
} uz_wgrad_item;
long long uz_wgrad_multi_workspace_bytes(const uz_wgrad_item* items, int n); /* <0 on error */
int uz_wgrad_multi(const uz_wgrad_item* items, int n, void* workspace, void* stream);
/* `batch` independent one-tap problems of the shape `d` in one launch pair: problem b reads L + b * lb, R + b * rb
 * (strides in elements, multiples of 16 bytes) and writes out + b * ob floats -- the per-image products of the token
 * attention that contract over the rows of both operands (dV_b = A_b^T dO_b, dK_b = dS_b^T Q_b,
 * unet_transformer.py:133-136; the channel Gram matrix x_b^T x_b of transatt_unet.py:93-101).  The pixel split is
 * planned for the whole batch. */
long long uz_wgrad_batched_workspace_bytes(const uz_wgrad_desc* d, int batch);
int uz_wgrad_batched(const uz_wgrad_desc* d, int batch, const void* L, long long lb, const void* R, long long rb,
                     float* out, long long ob, void* workspace, void* stream);
/* The same with a second batch level: batch * batch2 problems, problem (b, h) reads L + b * lb + h * lb2, R + b * rb +
 * h * rb2 and writes out + (b * batch2 + h) * ob (the scores Q_h^T K_h of every (image, head), uctransnet.py:160-168). */
int uz_wgrad_batched2(const uz_wgrad_desc* d, int batch, int batch2, const void* L, long long lb, long long lb2, const void* R,
                      long long rb, long long rb2, float* out, long long ob, void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The network's first convolution as direct kernels (bf16 run mode; uz_conv_first.hip): nn.Conv2d(C <= 3, Cout in {32, 64},
 * k3, p1) on the fp32 NCHW input image (reference: the first layer of every UNet-family model, common_layers.py:28 reached
 * from unet.py:15; attention_unet.py:11; u2net.py:30) -- no im2col buffer.
 *   fwd:   y[p][co] (bf16 NHWC, pixel stride ldy) = bias[co] + sum_{c,ty,tx} w[co][c][ty][tx] x[n][c][h+ty-1][w+tx-1], x and w
 *          rounded to bf16 (what the im2col path stored), fp32 accumulation, one rounding; stats (nullable):
 *          uz_conv3x3_first_rows(N, H, W) partial rows [row][2][Cout] of sum / sum of squares of the STORED values
 *          (uz_bn_finalize reads them).
 *   wgrad: dw[co][c][ty][tx] (fp32, the parameter's own layout) = sum_p dy[p][co] x[n][c][h+ty-1][w+tx-1]; workspace of
 *          uz_conv3x3_first_wgrad_workspace_bytes() for the workgroups' partial slabs (fixed-order reduction).
 * ------------------------------------------------------------------------------------------- */
int uz_conv3x3_first_supported(int dtype, int C, int Cout);
int uz_conv3x3_first_rows(int N, int H, int W);
int uz_conv3x3_first_fwd(int dtype, const float* x, int N, int C, int H, int W, const float* w, const float* bias, int Cout,
                         void* y, int ldy, float* stats, void* stream);
long long uz_conv3x3_first_wgrad_workspace_bytes(int N, int H, int W, int Cout);
int uz_conv3x3_first_wgrad(int dtype, const float* x, int N, int C, int H, int W, const void* dy, int lddy, int Cout, float* dw,
                           void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Weight re-packing: fp32 master parameters in the reference layout -> kernel layout in run dtype.
 *   UZ_PACK_CONV_FWD : w[Co][Ci][T] (OIHW)     -> dst[Co][t*Ci + ci]
 *   UZ_PACK_CONV_DGRAD: w[Co][Ci][T]           -> dst[Ci][(T-1-t)*Co + co]   (flipped taps)
 *   UZ_PACK_CONVT_FWD : w[Ci][Co][4]           -> dst[t*Co + co][ci]
 *   UZ_PACK_CONVT_DGRAD: w[Ci][Co][4]          -> dst[ci][t*Co + co]
 *   UZ_PACK_IM2COL    : w[Co][Ci][T], K=T*Ci   -> dst[Co][Kpad], k = t*Ci + ci, zero padded
 *   UZ_PACK_VEC_REPEAT: v[Co] (Ci = 1)         -> dst[t*Co + c] = v[c], t < T, dst stays FP32 whatever `dtype` (the bias of
 *                       nn.ConvTranspose2d k2 s2 for each of the four sub-pixels of the pixel-shuffle store,
 *                       common_layers.py:104; batched form only: a model's four repeats ride in the weights' launch)
 * ------------------------------------------------------------------------------------------- */
enum { UZ_PACK_CONV_FWD = 0, UZ_PACK_CONV_DGRAD = 1, UZ_PACK_CONVT_FWD = 2, UZ_PACK_CONVT_DGRAD = 3,
       UZ_PACK_IM2COL = 4, UZ_PACK_VEC_REPEAT = 5 };
int uz_pack_weights(int dtype, int mode, const float* w, int Co, int Ci, int T, int Kpad, void* dst,
                    void* stream);
/* The same for a whole model in one launch: `items_device` is a device array of n_items entries
 * sorted by `begin` = number of destination elements of all earlier entries; total_elements = sum. */
typedef struct uz_pack_item {
  const float* src;
  void* dst;
  long long begin;
  int mode, Co, Ci, T, Kpad;
  int pad_;
} uz_pack_item;
int uz_pack_weights_batched(int dtype, const uz_pack_item* items_device, int n_items,
                            long long total_elements, void* stream);
/* 3x3 convolution weights (Co, Ci multiples of 32): forward AND input-gradient layouts from one
 * coalesced read of the OIHW tensor; either destination may be NULL. */
typedef struct uz_pack3x3_item {
  const float* src;
  void* dst_fwd;    /* [Co][t*Ci + ci]        (UZ_PACK_CONV_FWD)   */
  void* dst_dgrad;  /* [Ci][(8-t)*Co + co]    (UZ_PACK_CONV_DGRAD) */
  int Co, Ci;
  int tile_begin;   /* number of 32x32 (co, ci) tiles of all earlier items */
  int pad_;
} uz_pack3x3_item;
int uz_pack_conv3x3_batched(int dtype, const uz_pack3x3_item* items_device, int n_items, int total_tiles,
                            void* stream);

/* im2col of a small-channel NCHW fp32 input (the network input, unet.py:31 first conv):
 *   dst[p][t*C + c] = x[n, c, h+dy, w+dx] (zero padded), dst row length Kpad, k >= 9C zero. */
int uz_im2col3x3_nchw(int dtype, const float* x_nchw, int N, int C, int H, int W, int Kpad, void* dst,
                      void* stream);

/* ---------------------------------------------------------------------------------------------
 * BatchNorm2d (train): finalize statistics.  nn.BatchNorm2d, common_layers.py:29,32.
 *   mean = S1/count, var = S2/count - mean^2 (biased), invstd = 1/sqrt(var+eps)
 *   scale = gamma*invstd, shift = beta - mean*scale
 *   running_mean = (1-m)*rm + m*mean ; running_var = (1-m)*rv + m*var*count/(count-1)
 * stats_partial as written by uz_conv_igemm ([grid_m][2][C]).
 * ------------------------------------------------------------------------------------------- */
int uz_bn_finalize(const float* stats_partial, int grid_m, int C, double count, const float* gamma,
                   const float* beta, float eps, float momentum, float* running_mean,
                   float* running_var, float* scale, float* shift, float* mean, float* invstd,
                   void* stream);
/* eval mode: scale/shift from running statistics (SURVEY §2.3 "BatchNorm2d (eval)") */
int uz_bn_eval_scale(int C, const float* gamma, const float* beta, const float* running_mean,
                     const float* running_var, float eps, float* scale, float* shift, void* stream);

/* act = relu(scale*y + shift) written to `act` (ld = lda), and optionally its MaxPool2d(2,2)
 * (floor mode; common_layers.py:90) to `pooled`.  BN-apply + ReLU + pool in one pass. */
int uz_bn_relu_apply(int dtype, const void* y, int ldy, const float* scale, const float* shift,
                     int N, int H, int W, int C, void* act, int lda, void* pooled, int ldp,
                     void* stream);
/* Same with a residual: act = relu(scale*y + shift) + res, pooled = MaxPool2d(2,2)(act).
 * Replaces the RSU block tail `hx1d + hxin` (u2net.py:74,119,157,188,213) and the stage pools
 * (u2net.py:221-229).  res == NULL is uz_bn_relu_apply. */
int uz_bn_relu_add_apply(int dtype, const void* y, int ldy, const float* scale, const float* shift,
                         int N, int H, int W, int C, const void* res, int ldr, void* act, int lda,
                         void* pooled, int ldp, int pool_ceil, void* stream);
/* uz_bn_finalize + uz_bn_relu_add_apply in ONE launch (training forward of Conv -> BatchNorm -> ReLU, common_layers.py:28-33):
 * the first workgroups of the element pass sum the `rows` partial rows (same order as uz_bn_finalize: the same bits), write
 * vec = [scale | shift | mean | invstd] (4 C floats) and the running statistics, and release a flag the other workgroups wait
 * for before they read scale / shift.  *flag must be 0 at launch (one int per call, the caller clears its flag arena once per
 * step) and is left at the number of finalizing workgroups.  Where the grid is smaller than the finalize (a few pixels, many
 * channels) the two launches are issued instead and the flag is not touched. */
int uz_bn_relu_add_apply_fin(int dtype, const void* y, int ldy, const float* stats_partial, int rows, double count,
                             const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                             float* running_var, float* vec, int* flag, int N, int H, int W, int C, const void* res, int ldr,
                             void* act, int lda, void* pooled, int ldp, int pool_ceil, void* stream);
/* pool_ceil bit 0 = 1: pooled is (N, ceil(H/2), ceil(W/2), C), border windows clipped (ceil_mode=True, u2net.py:30);
 * bit 0 = 0: pooled is (N, H/2, W/2, C), an odd last row / column is not pooled (MaxPool2d default); bit 1: BatchNorm without
 * the ReLU; bit 2: the pass walks the tensor from its end (see uz_bnbwd_desc.pool_ceil; identical output). */

/* Backward of (BN train -> ReLU [-> MaxPool2d(2,2)]) in two passes.
 * The gradient arriving at the activation is
 *     g = g0[p] + g1[p] + (p is the first max of its 2x2 window ? gpool[window] : 0)
 * (any of g0/g1/gpool may be NULL).  Pass 1 (uz_bn_relu_bwd_reduce) writes one partial row per
 * workgroup into `workspace` (uz_bn_relu_bwd_workspace_bytes() bytes), then a finalize kernel
 * sums the rows in fixed order into sums[0][c] = sum g*mask, sums[1][c] = sum g*mask*xhat
 * (double) and dbeta = sums[0], dgamma = sums[1] (fp32; may be NULL).  Pass 2 writes
 *     dy = scale * (g*mask - sums0/count - xhat*sums1/count). */
typedef struct uz_bnbwd_desc {
  int dtype;
  int N, H, W, C;
  int ldy, ldg0, ldg1, ldgp, lddy;
  int pool_ceil; /* bit 0: geometry of gpool: 1 = (ceil(H/2), ceil(W/2)) with clipped border windows, 0 = (H/2, W/2);
                  * bit 1: the forward was BatchNorm WITHOUT ReLU (uz_bn_relu_add_apply with the same bit); with unit
                  * scale and zero shift that pair is a stand-alone MaxPool2d(2) of an arbitrary tensor
                  * (unet_transformer.py:151);
                  * bit 2: walk the tensors from their END (same values; the part the producing kernel wrote last is what
                  * the 256 MB Infinity Cache still holds -- a pass after a convolution runs 15 % faster that way; the
                  * partial rows of pass 1 then belong to other pixel groups, so the fp32 totals differ in the last bits
                  * between the two walks, each walk being deterministic) */
} uz_bnbwd_desc;
/* Statistics of a BatchNorm whose input is not a convolution output (pre-activation blocks, `ResidualConv`,
 * common_layers.py:186-187): per-channel sum and sum of squares of x (P pixels, C channels, row stride ld) as
 * partial rows [uz_colstats_rows()][2][C], the layout uz_bn_finalize() reads. */
int uz_colstats_rows(int dtype, int P, int C);
int uz_colstats(int dtype, const void* x, int ld, int P, int C, float* partial, void* stream);
long long uz_bn_relu_bwd_workspace_bytes(const uz_bnbwd_desc* d, int has_pool_grad);
int uz_bn_relu_bwd_reduce(const uz_bnbwd_desc* d, const void* y, const float* scale,
                          const float* shift, const float* mean, const float* invstd, const void* g0,
                          const void* g1, const void* gpool, void* workspace, double* sums,
                          float* dgamma, float* dbeta, void* stream);
int uz_bn_relu_bwd_apply(const uz_bnbwd_desc* d, const void* y, const float* scale,
                         const float* shift, const float* mean, const float* invstd, const void* g0,
                         const void* g1, const void* gpool, const double* sums, double count,
                         void* dy, void* stream);
/* Pass 1 alone: the partial rows [rows][2][C] into `workspace`, rows = uz_bn_relu_bwd_workspace_bytes() / (8 C); and
 * uz_bn_bwd_finalize + uz_bn_relu_bwd_apply in ONE launch over such rows (from uz_bn_relu_bwd_reduce_rows, uz_conv_igemm_bnred
 * or uz_outconv_bwd_bnred): the flag protocol of uz_bn_relu_add_apply_fin; sums / dgamma / dbeta are written as by
 * uz_bn_bwd_finalize (same order of additions). */
int uz_bn_relu_bwd_reduce_rows(const uz_bnbwd_desc* d, const void* y, const float* scale, const float* shift,
                               const float* mean, const float* invstd, const void* g0, const void* g1, const void* gpool,
                               void* workspace, void* stream);
int uz_bn_relu_bwd_apply_fin(const uz_bnbwd_desc* d, const void* y, const float* scale, const float* shift,
                             const float* mean, const float* invstd, const void* g0, const void* g1, const void* gpool,
                             const float* partial, int rows, double* sums, float* dgamma, float* dbeta, int* flag,
                             double count, void* dy, void* stream);
/* The finalize half of uz_bn_relu_bwd_reduce() alone: sums[2][C] (double), dbeta = sums[0], dgamma = sums[1] from
 * `rows` partial rows [rows][2][C] written by uz_conv_igemm_bnred(). */
int uz_bn_bwd_finalize(const float* partial, int rows, int C, double* sums, float* dgamma, float* dbeta, void* stream);

/* 1x1 convolution with few outputs (OutConv, common_layers.py:125), NCHW fp32 logits.
 *   out[n, k, h, w] = b[k] + sum_c x[p, c] * w[k, c],  k < Kout <= 8 */
int uz_outconv_fwd(int dtype, const void* x, int ldx, int N, int HW, int C, const float* w,
                   const float* b, int Kout, float* out_nchw, void* stream);
/* uz_outconv_fwd on the RAW output y of the convolution in front, read through that layer's BatchNorm + ReLU (bf16):
 *   x[p, c] = bf16(max(fma(y[p, c], scale[c], shift[c]), 0))   -- the value uz_bn_relu_apply() would have stored --
 * so the normalised activation of the last decoder block (DoubleConv's second half, common_layers.py:31-33, feeding OutConv,
 * :125) is never written down; its backward is uz_outconv_bwd_bnred(x = NULL). */
int uz_outconv_fwd_xf(int dtype, const void* y, int ldy, int N, int HW, int C, const float* scale, const float* shift,
                      const float* w, const float* b, int Kout, float* out_nchw, void* stream);
/* dx[p,c] = sum_k g[n,k,hw] * w[k,c];  dw[k,c] = sum_p g*x;  db[k] = sum_p g.
 * Two kernels (per-workgroup partial rows in `workspace`, then a fixed-order sum): deterministic. */
long long uz_outconv_bwd_workspace_bytes(int dtype, int N, int HW, int C, int Kout);
int uz_outconv_bwd(int dtype, const void* x, int ldx, int N, int HW, int C, const float* w, int Kout,
                   const float* g_nchw, void* dx, int lddx, float* dw, float* db, void* workspace,
                   void* stream);
/* uz_outconv_bwd with the first pass of a BatchNorm backward in the same pass over the pixels (bf16): x = relu(bn(bn_y))
 * feeds only this head, so the gradient dx it writes is the whole gradient of that activation; bn_partial receives
 * [uz_outconv_bwd_rows()][2][C] partial rows for uz_bn_bwd_finalize() (see uz_conv_igemm_bnred).  x == NULL: the
 * activation was never written down (uz_outconv_fwd_xf); the kernel forms it from bn_y, scale, shift (ldx unused). */
int uz_outconv_bwd_rows(int dtype, int N, int HW, int C);
int uz_outconv_bwd_bnred(int dtype, const void* x, int ldx, int N, int HW, int C, const float* w, int Kout,
                         const float* g_nchw, void* dx, int lddx, float* dw, float* db, void* workspace,
                         const void* bn_y, int ld_bny, const float* scale, const float* shift, const float* mean,
                         const float* invstd, float* bn_partial, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Additive attention gate (AttentionBlock.forward, attention_unet.py:34-40), everything except the
 * two 1x1 convolutions W_g / W_x (those are uz_conv_igemm with statistics):
 *   q[p]   = b_psi + sum_c relu(bn_g(g1raw) + bn_x(x1raw))[p,c] * w_psi[c]      uz_attn_psi_fwd
 *   out    = x * sigmoid(bn_q(q))                                               uz_attn_gate_fwd
 * and the backward chain (see uz_attn.hip).  vec_* are the [4][channels] (scale, shift, mean,
 * invstd) rows produced by uz_bn_finalize; partial buffers hold uz_attn_grid() rows.
 * ------------------------------------------------------------------------------------------- */
int uz_attn_grid(int dtype, int P, int channels);
int uz_attn_psi_fwd(int dtype, const void* g1raw, int ldg, const void* x1raw, int ldx, const float* vec_g,
                    const float* vec_x, const float* wpsi, const float* bpsi /* device, may be NULL */,
                    int P, int F, float* q, float* partial /* [grid(F)][2] */, void* stream);
int uz_attn_gate_fwd(int dtype, const void* x, int ldx, const float* q, const float* vec_q, int P, int C,
                     void* out, int ldo, void* stream);
int uz_attn_bwd_psi(int dtype, const void* dout, int ldd, const void* x, int ldx, const float* q,
                    const float* vec_q, int P, int C, void* dx_direct, int lddx, float* dz,
                    float* partial /* [grid(C)][2] */, void* stream);
int uz_attn_bwd_reduce(int dtype, const void* g1raw, int ldg, const void* x1raw, int ldx, const float* q,
                       const float* dz, const float* wpsi, const float* vec_g, const float* vec_x,
                       const float* vec_q, const double* a01, int P, int F,
                       float* partial /* [grid(F)][4F+1] */, void* stream);
int uz_attn_bwd_apply(int dtype, const void* g1raw, int ldg, const void* x1raw, int ldx, const float* q,
                      const float* dz, const float* wpsi, const float* vec_g, const float* vec_x,
                      const float* vec_q, const double* a01, const double* totals, int P, int F,
                      void* dg1raw, int lddg, void* dx1raw, int lddx, void* stream);
/* out[e] (double) = sum over rows of partial[row][e] */
int uz_sum_rows(const float* partial, int rows, int n, double* out, void* stream);
/* the same sums (accumulated in double) rounded to fp32 and written where they are wanted: elements
 * [0, n0) to out0, [n0, n) to out1 (NULL when n0 == n) -- e.g. the two halves of uz_layernorm_bwd()'s
 * rows straight into the gradient tensors of weight and bias */
int uz_sum_rows_f32(const float* partial, int rows, int n, float* out0, int n0, float* out1, void* stream);
/* ... of the first n columns of rows that are ld >= n floats apart (a column window of wider partial rows) */
int uz_sum_rows_f32_ld(const float* partial, int ld, int rows, int n, float* out0, int n0, float* out1,
                       void* stream);
/* backward of nearest x2 upsampling: dx[coarse pixel] = sum of its 2x2 fine pixels (H, W coarse) */
int uz_sum2x2(int dtype, const void* du, int ldu, int N, int H, int W, int C, void* dx, int lddx,
              void* stream);

/* ---------------------------------------------------------------------------------------------
 * U^2-Net pieces (reference: unet_zoo/models/u2net.py).
 * ------------------------------------------------------------------------------------------- */
/* F.interpolate(mode='bilinear', align_corners=False, size=(Ho, Wo)) (u2net.py:19-22).
 * Element (n, h, w, c) of x lives at x[n*x_img_stride + (h*Wi + w)*ldx + c] (strides in elements),
 * likewise y: NHWC activations (ld = channel count of the holding buffer) and NCHW 1-channel logit
 * planes (ld = 1, C = 1) go through the same entry point. */
int uz_bilinear_fwd(int dtype, const void* x, int ldx, long long x_img_stride, int N, int Hi, int Wi,
                    int C, void* y, int ldy, long long y_img_stride, int Ho, int Wo, void* stream);
/* its backward: g at (Ho, Wo) -> dx at (Hi, Wi), overwritten (gather form, deterministic) */
int uz_bilinear_bwd(int dtype, const void* g, int ldg, long long g_img_stride, int N, int Hi, int Wi,
                    int C, void* dx, int lddx, long long dx_img_stride, int Ho, int Wo, void* stream);
/* the same pair with the align_corners switch of F.interpolate / nn.Upsample (align_corners != 0:
 * nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True), nested_unet.py:32) */
int uz_resize_bilinear_fwd(int dtype, const void* x, int ldx, long long x_img_stride, int N, int Hi, int Wi,
                           int C, void* y, int ldy, long long y_img_stride, int Ho, int Wo, int align_corners,
                           void* stream);
int uz_resize_bilinear_bwd(int dtype, const void* g, int ldg, long long g_img_stride, int N, int Hi, int Wi,
                           int C, void* dx, int lddx, long long dx_img_stride, int Ho, int Wo, int align_corners,
                           void* stream);
/* Pixel-grid moves of NHWC tensors (row strides lds / ldd in elements, C channels):
 *   mode 0 copy (a tensor into its slot of a concat buffer: UNet++'s dense skips, nested_unet.py:80-93),
 *   mode 1 dst[h, w] = src[2h, 2w]  (Hd = ceil(Hs/2): what a stride-2 convolution keeps or reads),
 *   mode 2 dst[h, w] = (h, w even) ? src[h/2, w/2] : 0  (Hs = ceil(Hd/2): that selection's gradient). */
int uz_resample2(int dtype, const void* src, int lds, int N, int Hs, int Ws, int C, void* dst, int ldd, int Hd,
                 int Wd, int mode, void* stream);
/* out = g0 + g1 + unpool(gp): total gradient of `act` consumed directly (g0, g1 may be NULL) and through
 * MaxPool2d(2,2) (gp at (H/2, W/2), routed to the first maximum of each window as ATen does). */
int uz_pool_grad_combine(int dtype, int N, int H, int W, int C, const void* act, int lda, const void* g0,
                         int ldg0, const void* g1, int ldg1, const void* gp, int ldgp, void* out, int ldo,
                         int pool_ceil, void* stream);
/* Side head Conv2d(C, 1, 3, padding=1) (u2net.py:238-243): x NHWC, w = the (1, C, 3, 3) fp32 parameter,
 * bias 1 value or NULL, out[n*out_img_stride + h*W + w] fp32; taps_ws: N*9*H*W floats of scratch. */
int uz_sideconv3x3_fwd(int dtype, const void* x, int ldx, int N, int H, int W, int C, const float* w,
                       const float* bias, float* taps_ws, float* out, long long out_img_stride,
                       void* stream);
long long uz_sideconv3x3_bwd_workspace_bytes(int dtype, int N, int H, int W, int C);
/* g[n*g_img_stride + hw] = d(loss)/d(out); dx (NHWC, may be NULL), dw (1, C, 3, 3), db (1 value or NULL) */
int uz_sideconv3x3_bwd(int dtype, const void* x, int ldx, int N, int H, int W, int C, const float* w,
                       const float* g, long long g_img_stride, void* dx, int lddx, float* dw, float* db,
                       void* workspace, void* stream);
/* Fuse conv Conv2d(Cc, K, 1) on the NCHW fp32 concat d (N, Cc, HW) of the side maps (u2net.py:244,288) */
int uz_fuse1x1_fwd(const float* d, int N, int HW, int Cc, int K, const float* w, const float* b,
                   float* out, void* stream);
long long uz_fuse1x1_bwd_workspace_bytes(int N, int HW, int Cc, int K);
/* dcat = W^T g (+ g_extra[s] added onto channels [s*K, (s+1)*K): the gradients that arrive at the
 * side outputs themselves; host array of n_extra = Cc/K device pointers, entries may be NULL; n_extra
 * may be 0), dw (K, Cc), db (K).  g == NULL: only the extras flow, dw = db = 0. */
int uz_fuse1x1_bwd(const float* d, int N, int HW, int Cc, int K, const float* w, const float* g,
                   const float* const* g_extra, int n_extra, float* dcat, float* dw, float* db,
                   void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Swin-UNet V2 pieces (reference: unet_zoo/models/swin_unet_v2.py).  Tokens are NHWC activations:
 * row = (image, h, w) on the token grid, C channels.  nn.Linear runs on uz_conv_igemm (ntaps = 1).
 * ------------------------------------------------------------------------------------------- */
/* PatchEmbed's Conv2d(kernel = stride = patch) input (swin_unet_v2.py:548-556):
 * out[(b,i,j)][(kh*patch + kw)*C + c] = x[b][c][i*patch+kh][j*patch+kw], zero up to Kpad. */
int uz_patchify(int dtype, const float* x_nchw, int N, int C, int H, int W, int patch, int Kpad,
                void* out, void* stream);

/* nn.LayerNorm over C of an (N, Ho, Wo, C) token tensor.  mode selects how an output token's input
 * row is addressed (the reference's permutations, never materialised):
 *   UZ_LN_PLAIN   x[token][c]
 *   UZ_LN_MERGE   PatchMerging (:315-332): input grid (2Ho, 2Wo) with C/4 channels; channel segment
 *                 s = c/(C/4) of output token (i, j) is input token (2i + (s&1), 2j + (s>>1))
 *   UZ_LN_EXPAND  PatchExpand / FinalPatchExpand_X4 (:352-362, :375-387): input grid (Ho/r, Wo/r) with
 *                 r*r*C channels; output token (h*r+p1, w*r+p2) reads channels (p1*r+p2)*C + c
 * y = [res +] [image_scale[image] *] LN(x): the block tail shortcut + drop_path(norm1(.)) (:264-267).
 * stats: (P_out, 2) fp32 mean / rstd, kept for the backward. */
enum { UZ_LN_PLAIN = 0, UZ_LN_MERGE = 1, UZ_LN_EXPAND = 2 };
typedef struct uz_ln_desc {
  int dtype, N, Ho, Wo, C;
  int ldx, ldy, ldr, ldg, lddx; /* row strides (elements) of x, y, res, g (grad of y), dx */
  int mode, r;
  float eps;
  int act; /* 0: none; 1: exact GELU applied to the result (MixFFN_skip's act(norm1(.)), missformer.py:206;
              no residual / image scale; backward through uz_layernorm_act_bwd) */
} uz_ln_desc;
/* (every entry of this family: gamma / beta are read with 16-byte loads, stats -- (mean, rstd) pairs -- with 8-byte accesses:
 * 16- / 8-byte aligned pointers) */
int uz_layernorm_fwd(const uz_ln_desc* d, const void* x, const float* gamma, const float* beta,
                     const void* res, const float* image_scale, void* y, float* stats, void* stream);
int uz_layernorm_bwd_rows(const uz_ln_desc* d); /* rows of `partial`; <0 on error */
/* dx (addressed like x; every input element is written exactly once) and per-workgroup partial rows
 * partial[row][2][C] = sums of g*xhat (-> dgamma) and g (-> dbeta), g already scaled by image_scale;
 * add the rows with uz_sum_rows().  The residual branch's gradient is g itself. */
int uz_layernorm_bwd(const uz_ln_desc* d, const void* x, const float* gamma, const float* stats,
                     const void* g, const float* image_scale, void* dx, float* partial, void* stream);
/* the same for act = 1: g is the gradient of GELU(LayerNorm(x)); beta rebuilds the pre-activation */
int uz_layernorm_act_bwd(const uz_ln_desc* d, const void* x, const float* gamma, const float* beta,
                         const float* stats, const void* g, void* dx, float* partial, void* stream);
/* LayerNorm followed by a 1x1 head (FinalPatchExpand_X4's norm + the `output` convolution of swin_unet_v2,
 * swin_unet_v2.py:385, :690, :753) without materialising the normalised tensor: logits (N, K, Ho, Wo) fp32 =
 * b[k] + sum_c w[k][c] LN(x)[token][c] with x addressed as in uz_layernorm_fwd (d->ldy / ldr / ldg unused),
 * K <= 4 (K = 1: C <= 192 sixteen-byte chunks, else 64).  Backward from dlogits: dx with x's addressing (lddx)
 * and the gradients of gamma, beta (C), w (K, C), b (K; may be NULL), overwritten. */
int uz_ln_head_fwd(const uz_ln_desc* d, const void* x, const float* gamma, const float* beta, const float* w,
                   const float* b /* or NULL */, int K, float* logits, float* stats, void* stream);
long long uz_ln_head_bwd_workspace_bytes(const uz_ln_desc* d, int K);
int uz_ln_head_bwd(const uz_ln_desc* d, const void* x, const float* gamma, const float* beta, const float* w,
                   int K, const float* stats, const float* dlogits, void* dx, float* dgamma, float* dbeta,
                   float* dw, float* db /* or NULL */, float* workspace, void* stream);

/* WindowAttention core (:127-159) with window_partition / roll / window_reverse (:30-56, :246-262)
 * as index arithmetic: qkv (P, 3C) = [3][heads][32] per token, out (P, C).  For window tokens i, j
 *   S_ij = (scale q_i . k_j) / max(|scale q_i| |k_j|, 1e-6) / max(tau[h][i][j], 0.01) + bias[h][i][j]
 *          (- 100 when the shifted-window region ids differ, :214-236);  out_i = softmax_j(S) v
 * tau: (heads, Nt, Nt) parameter (Nt >= ws*ws), bias: (heads, ws*ws, ws*ws) fp32 = cpb MLP output.
 * lse: (B*nW, heads, ws*ws) row log-sum-exp kept for the backward.  head_dim must be 32, ws <= 8. */
typedef struct uz_winattn_desc {
  int dtype, B, H, W, C, heads, ws, shift, Nt;
  int ldq, ldo;
  float scale;
} uz_winattn_desc;
int uz_winattn_fwd(const uz_winattn_desc* d, const void* qkv, const float* tau, const float* bias,
                   void* out, float* lse, void* stream);
/* Continuous position bias (:121-125, Mlp_Relu :58-72): bias[h][r] = fc2(relu(fc1(idx[r]))) over the
 * R = N*N log-spaced offsets idx (R, 2); w1 (hidden, 2), b1 (hidden), w2 (heads, hidden), b2 (heads), fp32.
 * Backward from G = d bias (heads, R): gradients of the four parameter tensors, overwritten. */
int uz_cpb_fwd(const float* idx, const float* w1, const float* b1, const float* w2, const float* b2, int R,
               int hidden, int heads, float* bias, void* stream);
int uz_cpb_bwd(const float* idx, const float* w1, const float* b1, const float* w2, const float* G, int R,
               int hidden, int heads, float* dw1, float* db1, float* dw2, float* db2, void* stream);
/* All position-bias MLPs of a model in one launch each way (they depend on parameters only: forward at the
 * start of the step, backward once every d bias is known).  Forward reads idx, w1, b1, w2, b2 and writes
 * bias; backward reads idx, w1, b1, w2, G and overwrites dw1, db1, dw2, db2 (unused pointers may be NULL). */
typedef struct {
  const float* idx;      /* (R, 2) */
  const float* w1;       /* (hidden, 2) */
  const float* b1;       /* (hidden) */
  const float* w2;       /* (heads, hidden) */
  const float* b2;       /* (heads) */
  int R, hidden, heads, reserved;
  float* bias;           /* forward out (heads, R) */
  const float* G;        /* backward in (heads, R) */
  float* dw1;
  float* db1;
  float* dw2;
  float* db2;
} uz_cpb_item;
int uz_cpb_fwd_batched(const uz_cpb_item* items, int n, void* stream);
long long uz_cpb_bwd_batched_workspace_bytes(const uz_cpb_item* items, int n);
int uz_cpb_bwd_batched(const uz_cpb_item* items, int n, float* workspace, void* stream);
int uz_winattn_bwd_rows(const uz_winattn_desc* d); /* rows of `partial`; <0 on error */
/* dqkv (P, 3C) fully written; partial[row][2][heads][N][N]: sums over the row's windows of dS (-> d bias)
 * and of d tau (zero where tau < 0.01); add the rows with uz_sum_rows() / uz_sum_rows_f32(). */
int uz_winattn_bwd(const uz_winattn_desc* d, const void* qkv, const float* tau, const float* bias,
                   const void* out, const float* lse, const void* dout, int lddo, void* dqkv, int lddq,
                   float* partial, void* stream);

/* out[c] = sum_p x[p*ld + c] (fp32; out zeroed by caller). ConvTranspose2d bias gradient. */
int uz_colsum(int dtype, const void* x, int ld, int P, int C, float* out, void* stream);
/* Deterministic form (per-workgroup partial rows in `workspace`, fixed-order finalize; `out` is
 * overwritten, no zeroing needed): nn.Linear / PatchEmbed bias gradients (swin_unet_v2.py:120,123,546). */
long long uz_colsum_workspace_bytes(int dtype, int P, int C);
int uz_colsum_ws(int dtype, const void* x, int ld, int P, int C, float* out, void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Step tail: torch.nn.utils.clip_grad_norm_(params, max_norm) followed by torch.optim.AdamW.step()
 * (unet_zoo/utils/training_loop.py:119-121) over FLAT fp32 buffers p, g, m (exp_avg), v (exp_avg_sq) of n
 * elements.  The clipped gradient is consumed, not written back.  `step` is a device scalar (float),
 * incremented by this call; max_norm <= 0 disables clipping.  After the call workspace holds, at byte
 * offset 8192, four floats: clip coefficient, 1-beta1^t, 1-beta2^t, total gradient norm.
 * ------------------------------------------------------------------------------------------- */
long long uz_clip_adamw_workspace_bytes(void);
int uz_clip_adamw(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, float max_norm, float* step, void* workspace,
                  void* stream);

/* ---------------------------------------------------------------------------------------------
 * MISSFormer / MiT blocks (unet_zoo/models/missformer.py).  uz_mit.hip
 * ------------------------------------------------------------------------------------------- */
/* nn.GELU() (erf form) of MixFFN_skip (missformer.py:196,206) on a (P, C) token tensor */
int uz_gelu_fwd(int dtype, const void* x, int ldx, void* y, int ldy, long long P, int C, void* stream);
int uz_gelu_bwd(int dtype, const void* x, int ldx, const void* g, int ldg, void* dx, int lddx, long long P,
                int C, void* stream);
/* DWConv: Conv2d(C, C, 3, 1, 1, groups=C) on NHWC tokens (missformer.py:168-177).  w_taps = the (C,1,3,3)
 * parameter transposed to [9][C] fp32.  flags bit 0: y += x (the "dwconv(fc1) + fc1" of MixFFN_skip :205),
 * bit 1: flipped taps (the gradient with respect to the input; bias NULL). */
int uz_dwconv3x3(int dtype, const void* x, int ldx, const float* w_taps, const float* bias, void* y, int ldy,
                 int N, int H, int W, int C, int flags, void* stream);
/* partial sums part[rows][10][C] (taps 0..8, then the bias) of d(loss)/d(w_taps), d(loss)/d(bias); rows from
 * uz_dwconv3x3_wgrad_rows(); reduce over rows with uz_sum_rows_f32. */
int uz_dwconv3x3_wgrad_rows(int dtype, int N, int H, int W, int C);
int uz_dwconv3x3_wgrad(int dtype, const void* x, int ldx, const void* g, int ldg, float* part, int N, int H,
                       int W, int C, void* stream);
/* dst[n, ho, wo, (ty*r + tx)*C + c] = src[n, ho*r + ty, wo*r + tx, c]: the input of a Conv2d(C, C', r, r)
 * (EfficientSelfAtten.sr, missformer.py:17-18,26-27) as GEMM rows; inverse != 0 scatters such rows back. */
int uz_space_to_depth(int dtype, const void* src, int lds, void* dst, int ldd, int N, int Ho, int Wo, int C, int r,
                      int inverse, void* stream);
/* im2col of the NCHW fp32 input for OverlapPatchEmbeddings' Conv2d(3, 64, 7, 4, 3) (missformer.py:242,312):
 * out[(n, ho, wo)][(kh*k + kw)*C + c] in the run dtype, zero beyond k*k*C up to Kpad. */
int uz_im2col_nchw(int dtype, const float* x_nchw, int N, int C, int H, int W, int k, int stride, int pad, int Kpad,
                   void* out, void* stream);
/* softmax(q k^T * scale) v per (image, head), head_dim 64 (missformer.py:21-39, :113-128).
 * q / out: (B*N, ld) token tensors, head h in columns [64h, 64h+64).  Key j of image b is row
 * ((j / kps) * B + b) * kps + j % kps of k / v (kps = NK: one [B][NK] block; the bridge attends to four
 * blocks of kps rows, missformer.py:81-100).  lse: B*heads*N floats kept for the backward: log2 of the sum of
 * exp(scaled scores), i.e. logsumexp / ln 2. */
typedef struct uz_sra_desc {
  int dtype, B, N, NK, heads, head_dim, kps;
  int ldq, ldk, ldv, ldo;
  float scale;
} uz_sra_desc;
int uz_sra_fwd(const uz_sra_desc* d, const void* q, const void* k, const void* v, void* out, float* lse,
               void* stream);
long long uz_sra_bwd_workspace_bytes(const uz_sra_desc* d);
/* go = d(loss)/d(out); dq like q; dkv: dense (B*NK, 2*heads*64) rows laid out like the kv tensor, dK in the
 * first heads*64 columns and dV in the rest (k = kv, v = kv + heads*64, ldk = ldv = 2*heads*64). */
int uz_sra_bwd(const uz_sra_desc* d, const void* q, const void* k, const void* v, const void* o, const float* lse,
               const void* go, int ldgo, void* dq, int lddq, void* dkv, int lddkv, void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Loss and metric of the training step on the device.  uz_loss.hip
 * out2[0] = BCEWithLogitsLoss(logits, target) with mean reduction (scripts/train.py:135, training_loop.py:113-119),
 * out2[1] = dice_coefficient(logits, target) of the thresholded prediction (utils/metrics.py:7-24: sigmoid > 0.5,
 * epsilon 1e-7, 1.0 for an empty union); dlogits (may be NULL) = d(loss)/d(logits) = (sigmoid(x) - t) / n.
 * n fp32 elements each; two launches, fixed summation order, no host synchronisation (the reference reads
 * loss.item() / dc.item() every step, training_loop.py:123-124). */
long long uz_bce_dice_workspace_bytes(long long n);
int uz_bce_dice(const float* logits, const float* target, long long n, float* dlogits, float* out2, void* workspace,
                void* stream);

/* ---------------------------------------------------------------------------------------------
 * Bias gradients of a whole backward pass (or phase) in two launches.  uz_colsum.hip
 * out_i[c] = sum_p x_i[p*ld + c] (fp32) for n tensors of the run dtype: what `uz_colsum_ws` computes for one
 * nn.Linear / Conv2d bias (db = sum over tokens of the output gradient), for all of them at once; `items` is a
 * HOST array, copied into the kernel arguments (80 tensors per launch pair).  Deterministic. */
typedef struct uz_colsum_item {
  const void* x; /* (P, ld) rows, C <= ld columns used */
  float* out;    /* C floats */
  int P, C, ld, reserved;
} uz_colsum_item;
/* uz_sum_rows_f32 for n partial buffers at once (the parameter gradients that kernels leave as per-workgroup rows:
 * LayerNorm dgamma | dbeta, depthwise weight gradients): out0[e] / out1[e - n0] = sum_r partial[r*n + e]. */
typedef struct uz_sum_rows_item {
  const float* partial;
  float* out0;
  float* out1; /* may be NULL when n0 == n */
  int rows, n, n0, reserved;
} uz_sum_rows_item;
int uz_sum_rows_f32_batched(const uz_sum_rows_item* items, int n, void* stream);
long long uz_colsum_batched_workspace_bytes(int dtype, const uz_colsum_item* items, int n);
int uz_colsum_batched(int dtype, const uz_colsum_item* items, int n, void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Residual sums with ReLU: out = relu(a + b) (b may be NULL) and its gradient dx = g * [out > 0], which both
 * summands receive.  Replaces `x = x + temp; x = F.relu(x)` of Multiresblock.forward and `x = x + shortcut;
 * x = F.relu(x)` of Respath.forward (unet_zoo/models/multiresunet.py:79-80, 127-129, 133-135).
 * ------------------------------------------------------------------------------------------- */
int uz_add_relu(int dtype, const void* a, int lda, const void* b, int ldb, void* out, int ldo, long long P, int C,
                void* stream);
int uz_relu_bwd(int dtype, const void* out, int ldo, const void* g, int ldg, void* dx, int lddx, long long P, int C,
                void* stream);

/* ---------------------------------------------------------------------------------------------
 * Input pipeline on the GPU (BoneDataset.__getitem__, unet_zoo/data/datasets.py:40-59):
 *   transforms.Resize((512, 512)) on a PIL image = Pillow's antialiased two-pass BILINEAR resample of 8-bit pixels in
 *   22-bit fixed point (Pillow src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc,
 *   ImagingResampleHorizontal_8bpc / Vertical_8bpc), then ToTensor (p / 255) and Normalize((v - mean) / std) for the
 *   image, `> 0.5` for the mask.  Bit-exact with Pillow + torch on the CPU.
 * uz_pil_resample_h_u8: dst[h][x][c] (uint8, HWC) for x < Wout from src (H, Win, C) uint8; `bounds` = Wout pairs
 *   (first tap, tap count), `kk` = Wout rows of `ksize` fixed-point coefficients (device pointers, built on the host).
 * uz_pil_resample_v_f32: the vertical pass over (Hin, W, C) uint8 fused with the conversion: mode 0 writes
 *   out[c][y][x] = (p / 255 - mean[c]) / std[c] (fp32, CHW), mode 1 (C = 1) out = (p / 255 > 0.5).  mean / std: HOST
 *   pointers to C floats.
 * ------------------------------------------------------------------------------------------- */
int uz_pil_resample_h_u8(const void* src, int H, int Win, int C, const int* bounds, const int* kk, int ksize, int Wout,
                         void* dst, void* stream);
int uz_pil_resample_v_f32(const void* src, int Hin, int W, int C, const int* bounds, const int* kk, int ksize, int Hout,
                          const float* mean_host, const float* std_host, int mode, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Dense token attention (U-Transformer's MultiHeadSelfAttention / MultiHeadCrossAttention,
 * unet_zoo/models/unet_transformer.py:126-137, :190-213; TransAttUNet's PAM_Module and ScaledDotProductAttention,
 * unet_zoo/models/transatt_unet.py:41-49, :91-107): torch.bmm + nn.Softmax on (b, tokens, tokens) matrices.
 *
 * uz_gemm_nt: y_b[m][n] = sum_k x_b[m][k] * w_b[n][k] (+ bias[n]) (+ res_b[m][n]) for b < batch -- the LDS-DMA GEMM of
 *   uz_conv_igemm's 1x1 path with a matrix index on the grid.  Strides (ldx, ldw, ldy, ldres: rows; xb, wb, yb, resb:
 *   matrices; 0 = one operand shared by the batch) in ELEMENTS, all multiples of 16 bytes; one matrix < 2 GB.
 *   Replaces torch.bmm(Q, K^T), torch.bmm(A, V) and the MultiHeadDense products x @ W (:24-27) and their gradients
 *   with respect to the left operand; the gradients that contract over the ROWS of both operands (dV = A^T dO,
 *   dK = dS^T Q, dW = X^T dQ) are uz_wgrad with ntaps = 1.
 * uz_softmax_fwd: s_b <- softmax(scale * s_b) in place over axis 0 (every COLUMN sums to one: nn.Softmax(dim=1) of a
 *   (b, rows, cols) tensor, unet_transformer.py:123; three streaming launches: column statistics per 512-row slice,
 *   their combination, the elementwise apply -- `workspace` of uz_softmax_workspace_bytes()) or axis 1 (rows; at most
 *   2048 bf16 / 1024 fp32 columns; no workspace).
 * uz_softmax_bwd: g_b <- a_b * (g_b - dot) * scale in place, a = the softmax output, dot = sum of a * g along the
 *   normalised axis.  Axis 0: `dot` is a (batch, cols) fp32 buffer, computed by a first pass unless dot_given != 0
 *   (attention knows it more cheaply: sum_q A[q][k] dA[q][k] = dV[k] . V[k]); axis 1: computed in registers.
 * uz_adaptive_avgpool_fwd / _bwd: F.adaptive_avg_pool2d on NHWC maps (:196-198) and its gradient (accumulate != 0: added
 *   to dx).  uz_rowdot_f32: out[r] = sum_c a[r][c] * b[r][c] (a fp32, b run dtype).  uz_cast_rows: dst = (run dtype) src
 *   row by row (accumulate != 0: dst += src): the fp32 results of uz_wgrad back into activations.
 * ------------------------------------------------------------------------------------------- */
typedef struct uz_gemm_desc {
  int dtype, batch, M, N, K;
  int ldx, ldw, ldy, ldres;
  long long xb, wb, yb, resb;
  /* optional second batch level (0 / 1: none): batch * batch2 matrices, matrix (b, h) at + b * xb + h * xb2 ... --
   * UCTransNet's heads, which sit side by side in the channels of one token map (uctransnet.py:140-158) */
  int batch2;
  long long xb2, wb2, yb2, resb2;
} uz_gemm_desc;
int uz_gemm_nt(const uz_gemm_desc* d, const void* x, const void* w, const float* bias, const void* res, void* y,
               void* stream);
long long uz_softmax_workspace_bytes(int batch, int rows, int cols, int axis);   /* axis 0: partial column statistics */
int uz_softmax_fwd(int dtype, void* s, int ld, long long sb, int batch, int rows, int cols, int axis, float scale,
                   void* workspace, void* stream);
int uz_softmax_bwd(int dtype, const void* a, void* g, int ld, long long sb, int batch, int rows, int cols, int axis,
                   float scale, float* dot, int dot_given, void* stream);
int uz_adaptive_avgpool_fwd(int dtype, const void* x, int ldx, int N, int Hi, int Wi, int C, void* y, int ldy, int Ho,
                            int Wo, void* stream);
int uz_adaptive_avgpool_bwd(int dtype, const void* g, int ldg, int N, int Hi, int Wi, int C, void* dx, int lddx, int Ho,
                            int Wo, int accumulate, void* stream);
int uz_rowdot_f32(int dtype, const float* a, int lda, const void* b, int ldb, long long rows, int C, float* out,
                  void* stream);
int uz_cast_rows(int dtype, const float* src, int lds, void* dst, int ldd, long long rows, int C, int accumulate,
                 void* stream);
/* out[p][c] = x[p][c] + map[p % HW][c], map fp32 (HW, C): `x + self.pe(x)` (unet_transformer.py:128-129, :181-185), the
 * learned embeddings of transatt_unet.py:144-145 and swin_unet_v2.py:714-715, broadcast over the batch. */
int uz_add_map(int dtype, const void* x, int ldx, const float* map, void* out, int ldo, long long P, int HW, int C,
               void* stream);

/* nn.Dropout(p) in training mode (uctransnet.py:54, :222-223; swin_unet_v2.py:158, :716): out = x * [u >= p] / (1 - p) with
 * u the caller's uniform draw (P x C fp32, contiguous; torch.rand: the generator stays torch's, a seed reproduces a run).
 * The backward is the same call on the gradient with the same u. */
int uz_dropout(int dtype, const void* x, int ldx, const float* u, float p, void* out, int ldo, long long P, int C,
               void* stream);

/* The gate of UCTransNet's CCA (uctransnet.py:417-427): with s[n][c] = sigmoid(...) > 0 per (image, channel),
 *   mode 2 (forward):  out = relu(x * s)
 *   mode 0 (backward): out = g * [x > 0] * x            -- its per-image column sums are d(loss)/d(s)
 *   mode 1 (backward): out = g * [x > 0] * s + a[n][c]   -- a: the gradient reaching x through the global average behind s
 * x, g, out: (N * HW, C) activations; s, a: (N, C) fp32. */
int uz_chanscale_relu(int dtype, int mode, const void* g, int ldg, const void* x, int ldx, const float* s, const float* a,
                      int N, int HW, int C, void* out, int ldo, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Channel-wise cross attention of UCTransNet (Attention_org.forward, unet_zoo/models/uctransnet.py:160-216): the part
 * between the matrix products.  `scores` fp32 (B, H, C, KV) = Q^T K per (image, head) (uz_wgrad_batched); per plane:
 * x = scale * scores, InstanceNorm2d (mean / biased variance over the (C, KV) plane, eps, no affine; :176), softmax over KV
 * (:177).  fwd writes pcat[b][c][h * KV + kv] = P / H and its transpose pcat_t[b][h * KV + kv][c] in the run dtype: the
 * context product over K = H * KV with V (B, tokens, H * KV) then is `context_layer.mean(dim=3)` of all heads (:195-199) in
 * one uz_gemm_nt.  bwd takes d(loss)/d(pcat) (fp32, (B, C, H * KV)) and writes d(loss)/d(scores) as ds[b][h][c][kv] and
 * ds_t[b][h][kv][c] (run dtype), recomputing the statistics and probabilities from `scores`.  KV <= 1024.
 * ------------------------------------------------------------------------------------------- */
int uz_chanattn_probs_fwd(int dtype, const float* scores, int B, int H, int C, int KV, float scale, float eps, void* pcat,
                          void* pcat_t, void* stream);
int uz_chanattn_probs_bwd(int dtype, const float* scores, const float* dpc, int B, int H, int C, int KV, float scale,
                          float eps, void* ds, void* ds_t, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UNETZOO_HIP_H */
