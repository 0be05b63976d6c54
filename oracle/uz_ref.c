/* TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the arithmetic of libunetzoo_hip.so's entry points, one `<name>_ref` per entry with the SAME
 * signature, compiled from the same header (include/unetzoo_hip.h; SURVEY.md 8b, last sentence).  Host pointers instead of
 * device pointers, `stream` and `workspace` ignored, sums in double precision, results rounded once to the tensor's type
 * (fp32, or bf16 round-to-nearest-even).  Every function cites the reference line whose operator it restates; the CPU
 * tests (tests/test_c_ref.py) pin each of them against the torch operator the reference calls, the GPU tests
 * (tests/test_c_ref_gpu.py) hold the kernels against them on the same bytes.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's library
 * (oracle/libuz_ref.so, built by oracle/Makefile); the product never does.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/unetzoo_hip.h"

/* a `_ref` must have exactly the signature the header declares for the entry it restates */
#define UZ_SAME_SIGNATURE(name) static __typeof__(name)* const uz_sigcheck_##name __attribute__((unused)) = name##_ref

/* ---- element types ------------------------------------------------------------------------------------------------ */
static inline float bf16_to_f32(uint16_t v) {
  uint32_t u = (uint32_t)v << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static inline uint16_t f32_to_bf16(float f) { /* round to nearest even; NaN stays NaN */
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline double ld(int dtype, const void* p, long long i) {
  return dtype == UZ_BF16 ? (double)bf16_to_f32(((const uint16_t*)p)[i]) : (double)((const float*)p)[i];
}
/* store v rounded to the tensor type; returns the stored value */
static inline double st(int dtype, void* p, long long i, double v) {
  if (dtype == UZ_BF16) {
    const uint16_t b = f32_to_bf16((float)v);
    ((uint16_t*)p)[i] = b;
    return (double)bf16_to_f32(b);
  }
  ((float*)p)[i] = (float)v;
  return (double)(float)v;
}

/* ---- tap geometry (include/unetzoo_hip.h: UZ_TAPS_*) --------------------------------------------------------------- */
/* input pixel index (within image n's grid of the operand) read by output pixel (h, w) for tap t, or -1 (zero) */
static long long tap_pixel(int mode, int ntaps, int dil, int t, int n, int h, int w, int H, int W, int Hin, int Win) {
  if (mode == UZ_TAPS_CONV) {
    if (ntaps == 1) return ((long long)n * H + h) * W + w;
    const int hh = h + (t / 3 - 1) * dil, ww = w + (t % 3 - 1) * dil;
    if (hh < 0 || hh >= H || ww < 0 || ww >= W) return -1;
    return ((long long)n * H + hh) * W + ww;
  }
  if (mode == UZ_TAPS_CONV_UP2) { /* 3x3 on the nearest x2 upsampling of an (H/2, W/2) tensor, common_layers.py:69-72 */
    const int hh = h + t / 3 - 1, ww = w + t % 3 - 1;
    if (hh < 0 || hh >= H || ww < 0 || ww >= W) return -1;
    return ((long long)n * (H / 2) + (hh >> 1)) * (W / 2) + (ww >> 1);
  }
  if (mode == UZ_TAPS_GATHER2X2) { /* ConvTranspose2d k2 s2 input gradient, common_layers.py:104 */
    const int hh = 2 * h + (t >> 1), ww = 2 * w + (t & 1);
    if (hh >= Hin || ww >= Win) return -1;
    return ((long long)n * Hin + hh) * Win + ww;
  }
  /* UZ_TAPS_CONV_S2: Conv2d(k3, stride 2, padding 1), common_layers.py:188 */
  const int hh = 2 * h + t / 3 - 1, ww = 2 * w + t % 3 - 1;
  if (hh < 0 || hh >= Hin || ww < 0 || ww >= Win) return -1;
  return ((long long)n * Hin + hh) * Win + ww;
}

/* ---- uz_pack_weights: fp32 master parameters -> kernel layout -------------------------------------------------------- */
int uz_pack_weights_ref(int dtype, int mode, const float* w, int Co, int Ci, int T, int Kpad, void* dst, void* stream) {
  (void)stream;
  for (int co = 0; co < Co; ++co)
    for (int ci = 0; ci < Ci; ++ci)
      for (int t = 0; t < T; ++t) {
        if (mode == UZ_PACK_CONV_FWD) st(dtype, dst, (long long)co * T * Ci + (long long)t * Ci + ci, w[((long long)co * Ci + ci) * T + t]);
        else if (mode == UZ_PACK_CONV_DGRAD) st(dtype, dst, (long long)ci * T * Co + (long long)(T - 1 - t) * Co + co, w[((long long)co * Ci + ci) * T + t]);
        else if (mode == UZ_PACK_CONVT_FWD) st(dtype, dst, ((long long)t * Co + co) * Ci + ci, w[((long long)ci * Co + co) * T + t]);
        else if (mode == UZ_PACK_CONVT_DGRAD) st(dtype, dst, (long long)ci * T * Co + (long long)t * Co + co, w[((long long)ci * Co + co) * T + t]);
        else st(dtype, dst, (long long)co * Kpad + (long long)t * Ci + ci, w[((long long)co * Ci + ci) * T + t]);
      }
  if (mode == UZ_PACK_IM2COL)
    for (int co = 0; co < Co; ++co)
      for (int k = T * Ci; k < Kpad; ++k) st(dtype, dst, (long long)co * Kpad + k, 0.0);
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_pack_weights);

/* ---- uz_conv_igemm: nn.Conv2d k3 / k1 (common_layers.py:28,31,47,52,71; u2net.py:10), their input gradients,
 *      nn.ConvTranspose2d k2 s2 forward (pixel-shuffle store) and input gradient (gather taps) (common_layers.py:104).
 *      One statistics row (uz_conv_igemm_grid_m_ref() == 1): sum and sum of squares of the STORED values. ------------ */
int uz_conv_igemm_grid_m_ref(const uz_conv_desc* d) {
  (void)d;
  return 1;
}
UZ_SAME_SIGNATURE(uz_conv_igemm_grid_m);

int uz_conv_igemm_ref(const uz_conv_desc* d, const void* x, const void* w, const float* bias, void* y, float* stats,
                      void* stream) {
  (void)stream;
  const int K = d->ntaps * d->Cin;
  const int Hout = d->Hout ? d->Hout : 2 * d->H, Wout = d->Wout ? d->Wout : 2 * d->W;
  double* s1 = stats ? (double*)calloc(2 * (size_t)d->Nout, sizeof(double)) : NULL;
  for (int n = 0; n < d->N; ++n)
    for (int h = 0; h < d->H; ++h)
      for (int wq = 0; wq < d->W; ++wq) {
        const long long p = ((long long)n * d->H + h) * d->W + wq;
        for (int no = 0; no < d->Nout; ++no) {
          double acc = 0.0;
          for (int t = 0; t < d->ntaps; ++t) {
            const long long q = tap_pixel(d->taps_mode, d->ntaps, d->dil, t, n, h, wq, d->H, d->W, d->Hin, d->Win);
            if (q < 0) continue;
            for (int c = 0; c < d->Cin; ++c)
              acc += ld(d->dtype, x, q * d->ldx + c) * ld(d->dtype, w, (long long)no * K + (long long)t * d->Cin + c);
          }
          if (bias) acc += bias[no];
          long long o;
          if (d->store_mode == UZ_STORE_PLAIN) {
            o = p * d->ldy + no;
          } else { /* n = (2a + b) * Co + co -> pixel (2h + a, 2w + b), channel co */
            const int ab = no / d->Co, co = no % d->Co;
            o = (((long long)n * Hout + 2 * h + (ab >> 1)) * Wout + 2 * wq + (ab & 1)) * d->ldy + co;
          }
          const double v = st(d->dtype, y, o, acc);
          if (s1) {
            s1[no] += v;
            s1[d->Nout + no] += v * v;
          }
        }
      }
  if (s1) {
    for (int i = 0; i < 2 * d->Nout; ++i) stats[i] = (float)s1[i];
    free(s1);
  }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_conv_igemm);

/* ---- uz_wgrad: the weight gradients autograd produces for those layers (training_loop.py:119) ---------------------- */
int uz_wgrad_ref(const uz_wgrad_desc* d, const void* L, const void* R, float* out, void* workspace, void* stream) {
  (void)workspace;
  (void)stream;
  const long long n_out = (long long)d->Ci * d->Cj * d->ntaps;
  double* acc = (double*)calloc((size_t)n_out, sizeof(double));
  for (int n = 0; n < d->N; ++n)
    for (int h = 0; h < d->H; ++h)
      for (int w = 0; w < d->W; ++w) {
        const long long p = ((long long)n * d->H + h) * d->W + w;
        for (int t = 0; t < d->ntaps; ++t) {
          const long long q = tap_pixel(d->taps_mode, d->ntaps, d->dil, t, n, h, w, d->H, d->W, d->Hr, d->Wr);
          if (q < 0) continue;
          for (int i = 0; i < d->Ci; ++i) {
            const double l = ld(d->dtype, L, p * d->ldl + i);
            if (l == 0.0) continue;
            for (int j = 0; j < d->Cj; ++j) acc[((long long)i * d->Cj + j) * d->ntaps + t] += l * ld(d->dtype, R, q * d->ldr + j);
          }
        }
      }
  for (long long i = 0; i < n_out; ++i) out[i] = (float)acc[i];
  free(acc);
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_wgrad);

/* ---- nn.BatchNorm2d (train) + nn.ReLU + nn.MaxPool2d(2) (common_layers.py:29-33, :90) ------------------------------- */
int uz_bn_finalize_ref(const float* stats_partial, int grid_m, int C, double count, const float* gamma, const float* beta,
                       float eps, float momentum, float* running_mean, float* running_var, float* scale, float* shift,
                       float* mean, float* invstd, void* stream) {
  (void)stream;
  for (int c = 0; c < C; ++c) {
    double s1 = 0.0, s2 = 0.0;
    for (int g = 0; g < grid_m; ++g) {
      s1 += stats_partial[((long long)g * 2 + 0) * C + c];
      s2 += stats_partial[((long long)g * 2 + 1) * C + c];
    }
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0.0) var = 0.0;
    const double is = 1.0 / sqrt(var + (double)eps);
    const double sc = (gamma ? gamma[c] : 1.0) * is;
    if (scale) scale[c] = (float)sc;
    if (shift) shift[c] = (float)((beta ? beta[c] : 0.0) - m * sc);
    if (mean) mean[c] = (float)m;
    if (invstd) invstd[c] = (float)is;
    if (running_mean) running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
    if (running_var) running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * var * count / (count - 1.0));
  }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_bn_finalize);

int uz_bn_eval_scale_ref(int C, const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                         float eps, float* scale, float* shift, void* stream) {
  (void)stream;
  for (int c = 0; c < C; ++c) {
    const double sc = (gamma ? gamma[c] : 1.0) / sqrt((double)running_var[c] + (double)eps);
    scale[c] = (float)sc;
    shift[c] = (float)((beta ? beta[c] : 0.0) - running_mean[c] * sc);
  }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_bn_eval_scale);

int uz_bn_relu_apply_ref(int dtype, const void* y, int ldy, const float* scale, const float* shift, int N, int H, int W, int C,
                         void* act, int lda, void* pooled, int ldp, void* stream) {
  (void)stream;
  for (long long p = 0; p < (long long)N * H * W; ++p)
    for (int c = 0; c < C; ++c) {
      const double v = (double)(float)((float)ld(dtype, y, p * ldy + c) * scale[c] + shift[c]);   /* one fp32 fma-free evaluation */
      st(dtype, act, p * lda + c, v > 0.0 ? v : 0.0);
    }
  if (pooled) {
    const int Hp = H / 2, Wp = W / 2;
    for (int n = 0; n < N; ++n)
      for (int h = 0; h < Hp; ++h)
        for (int w = 0; w < Wp; ++w)
          for (int c = 0; c < C; ++c) {
            double m = -INFINITY;
            for (int a = 0; a < 2; ++a)
              for (int b = 0; b < 2; ++b) {
                const double v = ld(dtype, act, (((long long)n * H + 2 * h + a) * W + 2 * w + b) * lda + c);
                if (v > m) m = v;
              }
            st(dtype, pooled, (((long long)n * Hp + h) * Wp + w) * ldp + c, m);
          }
  }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_bn_relu_apply);

/* gradient arriving at activation pixel (n, h, w), channel c: g0 + g1 + (first maximum of its 2x2 window ? gpool : 0) */
static double bn_bwd_g(const uz_bnbwd_desc* d, const void* y, const float* scale, const float* shift, const void* g0, const void* g1,
                       const void* gpool, int n, int h, int w, int c) {
  const long long p = ((long long)n * d->H + h) * d->W + w;
  double g = 0.0;
  if (g0) g += ld(d->dtype, g0, p * d->ldg0 + c);
  if (g1) g += ld(d->dtype, g1, p * d->ldg1 + c);
  if (gpool) {
    const int ceil_mode = d->pool_ceil & 1, relu = !(d->pool_ceil & 2);
    const int Hp = ceil_mode ? (d->H + 1) / 2 : d->H / 2, Wp = ceil_mode ? (d->W + 1) / 2 : d->W / 2;
    const int ph = h / 2, pw = w / 2;
    if (ph < Hp && pw < Wp) {
      /* first maximum of the window in raster order, on the activation as it was stored */
      double best = -INFINITY;
      int bh = -1, bw = -1;
      for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) {
          const int hh = 2 * ph + a, ww = 2 * pw + b;
          if (hh >= d->H || ww >= d->W) continue;
          const long long q = ((long long)n * d->H + hh) * d->W + ww;
          float v = (float)ld(d->dtype, y, q * d->ldy + c) * scale[c] + shift[c];
          if (relu && v < 0.f) v = 0.f;
          const double vs = d->dtype == UZ_BF16 ? (double)bf16_to_f32(f32_to_bf16(v)) : (double)v;
          if (vs > best) {
            best = vs;
            bh = hh;
            bw = ww;
          }
        }
      if (bh == h && bw == w) g += ld(d->dtype, gpool, (((long long)n * Hp + ph) * Wp + pw) * d->ldgp + c);
    }
  }
  return g;
}

int uz_bn_relu_bwd_reduce_ref(const uz_bnbwd_desc* d, const void* y, const float* scale, const float* shift, const float* mean,
                              const float* invstd, const void* g0, const void* g1, const void* gpool, void* workspace, double* sums,
                              float* dgamma, float* dbeta, void* stream) {
  (void)workspace;
  (void)stream;
  const int relu = !(d->pool_ceil & 2);
  for (int c = 0; c < d->C; ++c) {
    double s0 = 0.0, s1 = 0.0;
    for (int n = 0; n < d->N; ++n)
      for (int h = 0; h < d->H; ++h)
        for (int w = 0; w < d->W; ++w) {
          const long long p = ((long long)n * d->H + h) * d->W + w;
          const float yv = (float)ld(d->dtype, y, p * d->ldy + c);
          if (relu && !(yv * scale[c] + shift[c] > 0.f)) continue;
          const double g = bn_bwd_g(d, y, scale, shift, g0, g1, gpool, n, h, w, c);
          s0 += g;
          s1 += g * ((double)yv - mean[c]) * invstd[c];
        }
    sums[c] = s0;
    sums[d->C + c] = s1;
    if (dbeta) dbeta[c] = (float)s0;
    if (dgamma) dgamma[c] = (float)s1;
  }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_bn_relu_bwd_reduce);

int uz_bn_relu_bwd_apply_ref(const uz_bnbwd_desc* d, const void* y, const float* scale, const float* shift, const float* mean,
                             const float* invstd, const void* g0, const void* g1, const void* gpool, const double* sums, double count,
                             void* dy, void* stream) {
  (void)stream;
  const int relu = !(d->pool_ceil & 2);
  for (int n = 0; n < d->N; ++n)
    for (int h = 0; h < d->H; ++h)
      for (int w = 0; w < d->W; ++w)
        for (int c = 0; c < d->C; ++c) {
          const long long p = ((long long)n * d->H + h) * d->W + w;
          const float yv = (float)ld(d->dtype, y, p * d->ldy + c);
          const int on = !relu || (yv * scale[c] + shift[c] > 0.f);
          const double g = on ? bn_bwd_g(d, y, scale, shift, g0, g1, gpool, n, h, w, c) : 0.0;
          const double xhat = ((double)yv - mean[c]) * invstd[c];
          st(d->dtype, dy, p * d->lddy + c, scale[c] * (g - sums[c] / count - xhat * sums[d->C + c] / count));
        }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_bn_relu_bwd_apply);

/* ---- OutConv (common_layers.py:118-128) ---------------------------------------------------------------------------- */
int uz_outconv_fwd_ref(int dtype, const void* x, int ldx, int N, int HW, int C, const float* w, const float* b, int Kout,
                       float* out_nchw, void* stream) {
  (void)stream;
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < Kout; ++k)
      for (int q = 0; q < HW; ++q) {
        double acc = b ? b[k] : 0.0;
        for (int c = 0; c < C; ++c) acc += ld(dtype, x, ((long long)n * HW + q) * ldx + c) * w[(long long)k * C + c];
        out_nchw[((long long)n * Kout + k) * HW + q] = (float)acc;
      }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_outconv_fwd);

/* the head on the RAW output of the last convolution, read through BatchNorm + ReLU (common_layers.py:31-33 then :125): the
 * activation is the value uz_bn_relu_apply would have stored -- fp32 fma, max with 0, rounded to bf16 -- then the head as above */
int uz_outconv_fwd_xf_ref(int dtype, const void* y, int ldy, int N, int HW, int C, const float* scale, const float* shift,
                          const float* w, const float* b, int Kout, float* out_nchw, void* stream) {
  (void)stream;
  if (dtype != UZ_BF16) return UZ_ENOTIMPL;
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < Kout; ++k)
      for (int q = 0; q < HW; ++q) {
        double acc = b ? b[k] : 0.0;
        for (int c = 0; c < C; ++c) {
          const float raw = (float)ld(dtype, y, ((long long)n * HW + q) * ldy + c);
          float a = fmaf(raw, scale[c], shift[c]);
          a = a > 0.f ? a : 0.f;
          acc += (double)bf16_to_f32(f32_to_bf16(a)) * w[(long long)k * C + c];
        }
        out_nchw[((long long)n * Kout + k) * HW + q] = (float)acc;
      }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_outconv_fwd_xf);

/* ---- dense token attention (unet_transformer.py:126-137, :200-213; transatt_unet.py:41-49, :91-107) ------------------ */
int uz_gemm_nt_ref(const uz_gemm_desc* d, const void* x, const void* w, const float* bias, const void* res, void* y, void* stream) {
  (void)stream;
  const int nb2 = d->batch2 > 1 ? d->batch2 : 1;
  for (int b = 0; b < d->batch; ++b)
    for (int h = 0; h < nb2; ++h) {
      const long long xo = b * d->xb + h * d->xb2, wo = b * d->wb + h * d->wb2, yo = b * d->yb + h * d->yb2, ro = b * d->resb + h * d->resb2;
      for (int m = 0; m < d->M; ++m)
        for (int n = 0; n < d->N; ++n) {
          double acc = 0.0;
          for (int k = 0; k < d->K; ++k) acc += ld(d->dtype, x, xo + (long long)m * d->ldx + k) * ld(d->dtype, w, wo + (long long)n * d->ldw + k);
          if (bias) acc += bias[n];
          const long long o = yo + (long long)m * d->ldy + n;
          if (res) { /* the product is rounded to the tensor type before the residual is added (uz_conv_igemm_res) */
            const double v = st(d->dtype, y, o, acc);
            st(d->dtype, y, o, v + ld(d->dtype, res, ro + (long long)m * d->ldres + n));
          } else {
            st(d->dtype, y, o, acc);
          }
        }
    }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_gemm_nt);

long long uz_wgrad_batched_workspace_bytes_ref(const uz_wgrad_desc* d, int batch) {
  (void)d;
  (void)batch;
  return 0;
}
UZ_SAME_SIGNATURE(uz_wgrad_batched_workspace_bytes);

int uz_wgrad_batched2_ref(const uz_wgrad_desc* d, int batch, int batch2, const void* L, long long lb, long long lb2, const void* R,
                          long long rb, long long rb2, float* out, long long ob, void* workspace, void* stream) {
  const int es = d->dtype == UZ_BF16 ? 2 : 4;
  for (int b = 0; b < batch; ++b)
    for (int h = 0; h < batch2; ++h)
      uz_wgrad_ref(d, (const char*)L + (b * lb + h * lb2) * es, (const char*)R + (b * rb + h * rb2) * es,
                   out + ((long long)b * batch2 + h) * ob, workspace, stream);
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_wgrad_batched2);

int uz_wgrad_batched_ref(const uz_wgrad_desc* d, int batch, const void* L, long long lb, const void* R, long long rb, float* out,
                         long long ob, void* workspace, void* stream) {
  return uz_wgrad_batched2_ref(d, batch, 1, L, lb, 0, R, rb, 0, out, ob, workspace, stream);
}
UZ_SAME_SIGNATURE(uz_wgrad_batched);

long long uz_softmax_workspace_bytes_ref(int batch, int rows, int cols, int axis) {
  (void)batch;
  (void)rows;
  (void)cols;
  (void)axis;
  return 0;
}
UZ_SAME_SIGNATURE(uz_softmax_workspace_bytes);

/* nn.Softmax(dim=1) of a (b, rows, cols) tensor = axis 0 (unet_transformer.py:123); nn.Softmax(dim=-1) = axis 1 */
int uz_softmax_fwd_ref(int dtype, void* s, int ldm, long long sb, int batch, int rows, int cols, int axis, float scale,
                       void* workspace, void* stream) {
  (void)workspace;
  (void)stream;
  const int outer = axis == 0 ? cols : rows, inner = axis == 0 ? rows : cols;
  for (int b = 0; b < batch; ++b)
    for (int o = 0; o < outer; ++o) {
#define UZ_AT(i) (b * sb + (axis == 0 ? (long long)(i) * ldm + o : (long long)o * ldm + (i)))
      double mx = -INFINITY, z = 0.0;
      for (int i = 0; i < inner; ++i) {
        const double v = ld(dtype, s, UZ_AT(i)) * scale;
        if (v > mx) mx = v;
      }
      for (int i = 0; i < inner; ++i) z += exp(ld(dtype, s, UZ_AT(i)) * scale - mx);
      for (int i = 0; i < inner; ++i) st(dtype, s, UZ_AT(i), exp(ld(dtype, s, UZ_AT(i)) * scale - mx) / z);
    }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_softmax_fwd);

int uz_softmax_bwd_ref(int dtype, const void* a, void* g, int ldm, long long sb, int batch, int rows, int cols, int axis, float scale,
                       float* dot, int dot_given, void* stream) {
  (void)stream;
  const int outer = axis == 0 ? cols : rows, inner = axis == 0 ? rows : cols;
  for (int b = 0; b < batch; ++b)
    for (int o = 0; o < outer; ++o) {
      double dsum = 0.0;
      if (axis == 0 && dot_given) {
        dsum = dot[(long long)b * cols + o];
      } else {
        for (int i = 0; i < inner; ++i) dsum += ld(dtype, a, UZ_AT(i)) * ld(dtype, g, UZ_AT(i));
        if (axis == 0 && dot) dot[(long long)b * cols + o] = (float)dsum;
      }
      for (int i = 0; i < inner; ++i) st(dtype, g, UZ_AT(i), ld(dtype, a, UZ_AT(i)) * (ld(dtype, g, UZ_AT(i)) - dsum) * scale);
#undef UZ_AT
    }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_softmax_bwd);

/* F.adaptive_avg_pool2d (unet_transformer.py:196-198): window of output i = [floor(i In / Out), ceil((i + 1) In / Out)) */
int uz_adaptive_avgpool_fwd_ref(int dtype, const void* x, int ldx, int N, int Hi, int Wi, int C, void* y, int ldy, int Ho, int Wo,
                                void* stream) {
  (void)stream;
  for (int n = 0; n < N; ++n)
    for (int oh = 0; oh < Ho; ++oh)
      for (int ow = 0; ow < Wo; ++ow) {
        const int h0 = (int)((long long)oh * Hi / Ho), h1 = (int)(((long long)(oh + 1) * Hi + Ho - 1) / Ho);
        const int w0 = (int)((long long)ow * Wi / Wo), w1 = (int)(((long long)(ow + 1) * Wi + Wo - 1) / Wo);
        for (int c = 0; c < C; ++c) {
          double acc = 0.0;
          for (int h = h0; h < h1; ++h)
            for (int w = w0; w < w1; ++w) acc += ld(dtype, x, (((long long)n * Hi + h) * Wi + w) * ldx + c);
          st(dtype, y, (((long long)n * Ho + oh) * Wo + ow) * ldy + c, acc / ((h1 - h0) * (w1 - w0)));
        }
      }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_adaptive_avgpool_fwd);

int uz_adaptive_avgpool_bwd_ref(int dtype, const void* g, int ldg, int N, int Hi, int Wi, int C, void* dx, int lddx, int Ho, int Wo,
                                int accumulate, void* stream) {
  (void)stream;
  double* acc = (double*)calloc((size_t)N * Hi * Wi * C, sizeof(double));
  for (int n = 0; n < N; ++n)
    for (int oh = 0; oh < Ho; ++oh)
      for (int ow = 0; ow < Wo; ++ow) {
        const int h0 = (int)((long long)oh * Hi / Ho), h1 = (int)(((long long)(oh + 1) * Hi + Ho - 1) / Ho);
        const int w0 = (int)((long long)ow * Wi / Wo), w1 = (int)(((long long)(ow + 1) * Wi + Wo - 1) / Wo);
        for (int c = 0; c < C; ++c) {
          const double v = ld(dtype, g, (((long long)n * Ho + oh) * Wo + ow) * ldg + c) / ((h1 - h0) * (w1 - w0));
          for (int h = h0; h < h1; ++h)
            for (int w = w0; w < w1; ++w) acc[(((long long)n * Hi + h) * Wi + w) * C + c] += v;
        }
      }
  for (long long p = 0; p < (long long)N * Hi * Wi; ++p)
    for (int c = 0; c < C; ++c) st(dtype, dx, p * lddx + c, acc[p * C + c] + (accumulate ? ld(dtype, dx, p * lddx + c) : 0.0));
  free(acc);
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_adaptive_avgpool_bwd);

int uz_add_map_ref(int dtype, const void* x, int ldx, const float* map, void* out, int ldo, long long P, int HW, int C, void* stream) {
  (void)stream;
  for (long long p = 0; p < P; ++p)
    for (int c = 0; c < C; ++c) st(dtype, out, p * ldo + c, (double)((float)ld(dtype, x, p * ldx + c) + map[(p % HW) * C + c]));
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_add_map);

int uz_rowdot_f32_ref(int dtype, const float* a, int lda, const void* b, int ldb, long long rows, int C, float* out, void* stream) {
  (void)stream;
  for (long long r = 0; r < rows; ++r) {
    double t = 0.0;
    for (int c = 0; c < C; ++c) t += (double)a[r * lda + c] * ld(dtype, b, r * ldb + c);
    out[r] = (float)t;
  }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_rowdot_f32);

int uz_cast_rows_ref(int dtype, const float* src, int lds, void* dst, int ldd, long long rows, int C, int accumulate, void* stream) {
  (void)stream;
  for (long long r = 0; r < rows; ++r)
    for (int c = 0; c < C; ++c)
      st(dtype, dst, r * ldd + c, (double)(accumulate ? (float)ld(dtype, dst, r * ldd + c) + src[r * lds + c] : src[r * lds + c]));
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_cast_rows);

/* ---- UCTransNet: softmax(InstanceNorm2d(scores / sqrt(KV))) per (image, head) plane (uctransnet.py:170-178) ---------- */
static void chan_plane(const float* pl, int C, int KV, float scale, float eps, double* sh, double* P) {
  const int n = C * KV;
  double mean = 0.0, var = 0.0;
  for (int i = 0; i < n; ++i) mean += (double)pl[i] * scale;
  mean /= n;
  for (int i = 0; i < n; ++i) {
    const double dlt = (double)pl[i] * scale - mean;
    var += dlt * dlt;
  }
  const double rstd = 1.0 / sqrt(var / n + (double)eps);
  for (int c = 0; c < C; ++c) {
    double mx = -INFINITY, z = 0.0;
    for (int k = 0; k < KV; ++k) {
      sh[c * KV + k] = ((double)pl[c * KV + k] * scale - mean) * rstd;
      if (sh[c * KV + k] > mx) mx = sh[c * KV + k];
    }
    for (int k = 0; k < KV; ++k) z += exp(sh[c * KV + k] - mx);
    for (int k = 0; k < KV; ++k) P[c * KV + k] = exp(sh[c * KV + k] - mx) / z;
  }
  sh[n] = rstd; /* one extra slot: the plane's 1 / sigma */
}

int uz_chanattn_probs_fwd_ref(int dtype, const float* scores, int B, int H, int C, int KV, float scale, float eps, void* pcat,
                              void* pcat_t, void* stream) {
  (void)stream;
  double* sh = (double*)malloc(((size_t)C * KV + 1) * sizeof(double));
  double* P = (double*)malloc((size_t)C * KV * sizeof(double));
  const long long HK = (long long)H * KV;
  for (int b = 0; b < B; ++b)
    for (int h = 0; h < H; ++h) {
      chan_plane(scores + ((long long)b * H + h) * C * KV, C, KV, scale, eps, sh, P);
      for (int c = 0; c < C; ++c)
        for (int k = 0; k < KV; ++k) {
          st(dtype, pcat, ((long long)b * C + c) * HK + (long long)h * KV + k, P[c * KV + k] / H);
          st(dtype, pcat_t, ((long long)b * HK + (long long)h * KV + k) * C + c, P[c * KV + k] / H);
        }
    }
  free(sh);
  free(P);
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_chanattn_probs_fwd);

int uz_chanattn_probs_bwd_ref(int dtype, const float* scores, const float* dpc, int B, int H, int C, int KV, float scale, float eps,
                              void* ds, void* ds_t, void* stream) {
  (void)stream;
  const int n = C * KV;
  double* sh = (double*)malloc(((size_t)n + 1) * sizeof(double));
  double* P = (double*)malloc((size_t)n * sizeof(double));
  double* dsh = (double*)malloc((size_t)n * sizeof(double));
  const long long HK = (long long)H * KV;
  for (int b = 0; b < B; ++b)
    for (int h = 0; h < H; ++h) {
      chan_plane(scores + ((long long)b * H + h) * n, C, KV, scale, eps, sh, P);
      const double rstd = sh[n];
      double m1 = 0.0, m2 = 0.0;
      for (int c = 0; c < C; ++c) {
        double dot = 0.0;
        for (int k = 0; k < KV; ++k) dot += P[c * KV + k] * (double)dpc[((long long)b * C + c) * HK + (long long)h * KV + k] / H;
        for (int k = 0; k < KV; ++k) {
          dsh[c * KV + k] = P[c * KV + k] * ((double)dpc[((long long)b * C + c) * HK + (long long)h * KV + k] / H - dot);
          m1 += dsh[c * KV + k];
          m2 += dsh[c * KV + k] * sh[c * KV + k];
        }
      }
      m1 /= n;
      m2 /= n;
      for (int c = 0; c < C; ++c)
        for (int k = 0; k < KV; ++k) {
          const double v = rstd * (dsh[c * KV + k] - m1 - sh[c * KV + k] * m2) * scale;
          st(dtype, ds, (((long long)b * H + h) * C + c) * KV + k, v);
          st(dtype, ds_t, (((long long)b * H + h) * KV + k) * C + c, v);
        }
    }
  free(sh);
  free(P);
  free(dsh);
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_chanattn_probs_bwd);


/* ---- element passes of the transformer / residual families --------------------------------------------------------- */
/* nn.GELU() (erf form; missformer.py:196,206) and its derivative */
int uz_gelu_fwd_ref(int dtype, const void* x, int ldx, void* y, int ldy, long long P, int C, void* stream) {
  (void)stream;
  for (long long p = 0; p < P; ++p)
    for (int c = 0; c < C; ++c) {
      const double v = ld(dtype, x, p * ldx + c);
      st(dtype, y, p * ldy + c, 0.5 * v * (1.0 + erf(v * 0.70710678118654752440)));
    }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_gelu_fwd);

int uz_gelu_bwd_ref(int dtype, const void* x, int ldx, const void* g, int ldg, void* dx, int lddx, long long P, int C, void* stream) {
  (void)stream;
  for (long long p = 0; p < P; ++p)
    for (int c = 0; c < C; ++c) {
      const double v = ld(dtype, x, p * ldx + c);
      const double d = 0.5 * (1.0 + erf(v * 0.70710678118654752440)) + v * exp(-0.5 * v * v) * 0.39894228040143267794;
      st(dtype, dx, p * lddx + c, ld(dtype, g, p * ldg + c) * d);
    }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_gelu_bwd);

/* relu(a + b) and its gradient (multiresunet.py:79-80, 127-129, 133-135) */
int uz_add_relu_ref(int dtype, const void* a, int lda, const void* b, int ldb, void* out, int ldo, long long P, int C, void* stream) {
  (void)stream;
  for (long long p = 0; p < P; ++p)
    for (int c = 0; c < C; ++c) {
      const double v = ld(dtype, a, p * lda + c) + (b ? ld(dtype, b, p * ldb + c) : 0.0);
      st(dtype, out, p * ldo + c, v > 0.0 ? v : 0.0);
    }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_add_relu);

int uz_relu_bwd_ref(int dtype, const void* out, int ldo, const void* g, int ldg, void* dx, int lddx, long long P, int C, void* stream) {
  (void)stream;
  for (long long p = 0; p < P; ++p)
    for (int c = 0; c < C; ++c) st(dtype, dx, p * lddx + c, ld(dtype, out, p * ldo + c) > 0.0 ? ld(dtype, g, p * ldg + c) : 0.0);
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_relu_bwd);

/* backward of nn.Upsample(scale_factor=2) (common_layers.py:70): dx[coarse] = sum of its 2 x 2 fine pixels */
int uz_sum2x2_ref(int dtype, const void* du, int ldu, int N, int H, int W, int C, void* dx, int lddx, void* stream) {
  (void)stream;
  for (int n = 0; n < N; ++n)
    for (int h = 0; h < H; ++h)
      for (int w = 0; w < W; ++w)
        for (int c = 0; c < C; ++c) {
          double acc = 0.0;
          for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b) acc += ld(dtype, du, (((long long)n * 2 * H + 2 * h + a) * 2 * W + 2 * w + b) * ldu + c);
          st(dtype, dx, (((long long)n * H + h) * W + w) * lddx + c, acc);
        }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_sum2x2);

/* F.interpolate(mode='bilinear', align_corners=...) (u2net.py:19-22; nested_unet.py:32): ATen's source index rule */
int uz_resize_bilinear_fwd_ref(int dtype, const void* x, int ldx, long long x_img_stride, int N, int Hi, int Wi, int C, void* y,
                               int ldy, long long y_img_stride, int Ho, int Wo, int align_corners, void* stream) {
  (void)stream;
  const double sh = align_corners ? (Ho > 1 ? (double)(Hi - 1) / (Ho - 1) : 0.0) : (double)Hi / Ho;
  const double sw = align_corners ? (Wo > 1 ? (double)(Wi - 1) / (Wo - 1) : 0.0) : (double)Wi / Wo;
  for (int n = 0; n < N; ++n)
    for (int oh = 0; oh < Ho; ++oh)
      for (int ow = 0; ow < Wo; ++ow) {
        double fh = align_corners ? oh * sh : (oh + 0.5) * sh - 0.5, fw = align_corners ? ow * sw : (ow + 0.5) * sw - 0.5;
        if (fh < 0.0) fh = 0.0;
        if (fw < 0.0) fw = 0.0;
        const int h0 = (int)fh < Hi - 1 ? (int)fh : Hi - 1, w0 = (int)fw < Wi - 1 ? (int)fw : Wi - 1;
        const int h1 = h0 < Hi - 1 ? h0 + 1 : h0, w1 = w0 < Wi - 1 ? w0 + 1 : w0;
        const double lh = fh - h0, lw = fw - w0;
        for (int c = 0; c < C; ++c) {
#define UZ_X(hh, ww) ld(dtype, x, n * x_img_stride + ((long long)(hh) * Wi + (ww)) * ldx + c)
          const double v = (1 - lh) * ((1 - lw) * UZ_X(h0, w0) + lw * UZ_X(h0, w1)) + lh * ((1 - lw) * UZ_X(h1, w0) + lw * UZ_X(h1, w1));
#undef UZ_X
          st(dtype, y, n * y_img_stride + ((long long)oh * Wo + ow) * ldy + c, v);
        }
      }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_resize_bilinear_fwd);

int uz_bilinear_fwd_ref(int dtype, const void* x, int ldx, long long x_img_stride, int N, int Hi, int Wi, int C, void* y, int ldy,
                        long long y_img_stride, int Ho, int Wo, void* stream) {
  return uz_resize_bilinear_fwd_ref(dtype, x, ldx, x_img_stride, N, Hi, Wi, C, y, ldy, y_img_stride, Ho, Wo, 0, stream);
}
UZ_SAME_SIGNATURE(uz_bilinear_fwd);

/* dst[n, ho, wo, (ty r + tx) C + c] = src[n, ho r + ty, wo r + tx, c] (missformer.py:17-18, 26-27) and its inverse */
int uz_space_to_depth_ref(int dtype, const void* src, int lds, void* dst, int ldd, int N, int Ho, int Wo, int C, int r, int inverse,
                          void* stream) {
  (void)stream;
  for (int n = 0; n < N; ++n)
    for (int ho = 0; ho < Ho; ++ho)
      for (int wo = 0; wo < Wo; ++wo)
        for (int ty = 0; ty < r; ++ty)
          for (int tx = 0; tx < r; ++tx)
            for (int c = 0; c < C; ++c) {
              const long long fine = (((long long)n * Ho * r + ho * r + ty) * Wo * r + wo * r + tx);
              const long long coarse = ((long long)n * Ho + ho) * Wo + wo;
              const long long col = (long long)(ty * r + tx) * C + c;
              if (!inverse) st(dtype, dst, coarse * ldd + col, ld(dtype, src, fine * lds + c));
              else st(dtype, dst, fine * ldd + c, ld(dtype, src, coarse * lds + col));
            }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_space_to_depth);

/* per-channel sums over the pixels: nn.Linear / PatchEmbed bias gradients (swin_unet_v2.py:120,123,546) */
int uz_colsum_ref(int dtype, const void* x, int ldm, int P, int C, float* out, void* stream) {
  (void)stream;
  for (int c = 0; c < C; ++c) {
    double t = 0.0;
    for (long long p = 0; p < P; ++p) t += ld(dtype, x, p * ldm + c);
    out[c] = (float)t;
  }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_colsum);

/* DWConv: Conv2d(C, C, 3, 1, 1, groups=C) on NHWC tokens (missformer.py:168-177); w_taps [9][C] */
int uz_dwconv3x3_ref(int dtype, const void* x, int ldx, const float* w_taps, const float* bias, void* y, int ldy, int N, int H, int W,
                     int C, int flags, void* stream) {
  (void)stream;
  for (int n = 0; n < N; ++n)
    for (int h = 0; h < H; ++h)
      for (int w = 0; w < W; ++w)
        for (int c = 0; c < C; ++c) {
          double acc = bias ? bias[c] : 0.0;
          for (int t = 0; t < 9; ++t) {
            const int hh = h + t / 3 - 1, ww = w + t % 3 - 1;
            if (hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
            acc += ld(dtype, x, (((long long)n * H + hh) * W + ww) * ldx + c) * w_taps[((flags & 2) ? 8 - t : t) * C + c];
          }
          if (flags & 1) acc += ld(dtype, x, (((long long)n * H + h) * W + w) * ldx + c);
          st(dtype, y, (((long long)n * H + h) * W + w) * ldy + c, acc);
        }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_dwconv3x3);

/* nn.LayerNorm over the channels of a token map (swin_unet_v2.py, missformer.py): plain addressing (mode 0) with the
 * optional residual / per-image scale / GELU of the engine's fused forms; stats[token] = (mean, rstd).
 * y = res + image_scale[n] * LN(x); act = 1: y = GELU(LN(x)).  The merge / expand addressings are not restated. */
int uz_layernorm_fwd_ref(const uz_ln_desc* d, const void* x, const float* gamma, const float* beta, const void* res,
                         const float* image_scale, void* y, float* stats, void* stream) {
  (void)stream;
  if (d->mode < 0 || d->mode > 2) return UZ_ENOTIMPL;
  const long long P = (long long)d->N * d->Ho * d->Wo;
  double* row = malloc(sizeof(double) * d->C);
  for (long long p = 0; p < P; ++p) {
    /* the input row of output token p under the addressing mode (PatchMerging :315-332, PatchExpand :352-362, :375-387) */
    const int img = (int)(p / ((long long)d->Ho * d->Wo)), rem = (int)(p % ((long long)d->Ho * d->Wo));
    const int i = rem / d->Wo, j = rem % d->Wo;
    for (int c = 0; c < d->C; ++c) {
      long long off;
      if (d->mode == UZ_LN_PLAIN) {
        off = p * d->ldx + c;
      } else if (d->mode == UZ_LN_MERGE) { /* input grid (2Ho, 2Wo), C/4 channels; segment s of (i, j) is token (2i + (s&1), 2j + (s>>1)) */
        const int cq = d->C / 4, sgm = c / cq;
        off = (((long long)img * 2 * d->Ho + 2 * i + (sgm & 1)) * (2 * d->Wo) + 2 * j + (sgm >> 1)) * d->ldx + (c - sgm * cq);
      } else { /* input grid (Ho/r, Wo/r), r*r*C channels; token (h r + p1, w r + p2) reads channels (p1 r + p2) C + c */
        const int r = d->r, h = i / r, p1 = i % r, w = j / r, p2 = j % r;
        off = (((long long)img * (d->Ho / r) + h) * (d->Wo / r) + w) * d->ldx + (long long)(p1 * r + p2) * d->C + c;
      }
      row[c] = ld(d->dtype, x, off);
    }
    double m = 0.0, v = 0.0;
    for (int c = 0; c < d->C; ++c) m += row[c];
    m /= d->C;
    for (int c = 0; c < d->C; ++c) {
      const double t = row[c] - m;
      v += t * t;
    }
    const double rstd = 1.0 / sqrt(v / d->C + (double)d->eps);
    if (stats) {
      stats[2 * p] = (float)m;
      stats[2 * p + 1] = (float)rstd;
    }
    const double sc = image_scale ? image_scale[p / ((long long)d->Ho * d->Wo)] : 1.0;
    for (int c = 0; c < d->C; ++c) {
      double o = ((row[c] - m) * rstd * gamma[c] + beta[c]) * sc;
      if (d->act == 1) o = 0.5 * o * (1.0 + erf(o * 0.70710678118654752440));
      if (res) o += ld(d->dtype, res, p * d->ldr + c);
      st(d->dtype, y, p * d->ldy + c, o);
    }
  }
  free(row);
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_layernorm_fwd);

/* ---- uz_conv3x3_first_*: the network's first convolution on the fp32 NCHW input (common_layers.py:28 from unet.py:15) -- */
int uz_conv3x3_first_supported_ref(int dtype, int C, int Cout) {
  return dtype == UZ_BF16 && C >= 1 && C <= 3 && (Cout == 32 || Cout == 64);
}
UZ_SAME_SIGNATURE(uz_conv3x3_first_supported);

int uz_conv3x3_first_rows_ref(int N, int H, int W) { /* the restatement keeps its sums in one row */
  (void)N, (void)H, (void)W;
  return 1;
}
UZ_SAME_SIGNATURE(uz_conv3x3_first_rows);

static inline double rb(double v) { return (double)bf16_to_f32(f32_to_bf16((float)v)); } /* what the im2col path stored */

int uz_conv3x3_first_fwd_ref(int dtype, const float* x, int N, int C, int H, int W, const float* w, const float* bias, int Cout,
                             void* y, int ldy, float* stats, void* stream) {
  (void)stream;
  if (!uz_conv3x3_first_supported_ref(dtype, C, Cout)) return UZ_ENOTIMPL;
  double* s1 = stats ? (double*)calloc(2 * (size_t)Cout, sizeof(double)) : NULL;
  for (int n = 0; n < N; ++n)
    for (int h = 0; h < H; ++h)
      for (int wq = 0; wq < W; ++wq)
        for (int co = 0; co < Cout; ++co) {
          double acc = 0.0;
          for (int c = 0; c < C; ++c)
            for (int ty = 0; ty < 3; ++ty)
              for (int tx = 0; tx < 3; ++tx) {
                const int hh = h + ty - 1, ww = wq + tx - 1;
                if (hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
                acc += rb(x[(((long long)n * C + c) * H + hh) * W + ww]) * rb(w[((co * C + c) * 3 + ty) * 3 + tx]);
              }
          if (bias) acc += bias[co];
          const double v = st(dtype, y, (((long long)n * H + h) * W + wq) * ldy + co, acc);
          if (s1) {
            s1[co] += v;
            s1[Cout + co] += v * v;
          }
        }
  if (s1) {
    for (int i = 0; i < 2 * Cout; ++i) stats[i] = (float)s1[i];
    free(s1);
  }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_conv3x3_first_fwd);

long long uz_conv3x3_first_wgrad_workspace_bytes_ref(int N, int H, int W, int Cout) {
  (void)N, (void)H, (void)W, (void)Cout;
  return 0;
}
UZ_SAME_SIGNATURE(uz_conv3x3_first_wgrad_workspace_bytes);

int uz_conv3x3_first_wgrad_ref(int dtype, const float* x, int N, int C, int H, int W, const void* dy, int lddy, int Cout, float* dw,
                               void* workspace, void* stream) {
  (void)workspace, (void)stream;
  if (!uz_conv3x3_first_supported_ref(dtype, C, Cout)) return UZ_ENOTIMPL;
  for (int co = 0; co < Cout; ++co)
    for (int c = 0; c < C; ++c)
      for (int ty = 0; ty < 3; ++ty)
        for (int tx = 0; tx < 3; ++tx) {
          double acc = 0.0;
          for (int n = 0; n < N; ++n)
            for (int h = 0; h < H; ++h) {
              const int hh = h + ty - 1;
              if (hh < 0 || hh >= H) continue;
              for (int wq = 0; wq < W; ++wq) {
                const int ww = wq + tx - 1;
                if (ww < 0 || ww >= W) continue;
                acc += ld(dtype, dy, (((long long)n * H + h) * W + wq) * lddy + co) *
                       rb(x[(((long long)n * C + c) * H + hh) * W + ww]);
              }
            }
          dw[((co * C + c) * 3 + ty) * 3 + tx] = (float)acc;
        }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_conv3x3_first_wgrad);

/* ---- uz_wgrad_multi: the same weight gradients, issued together (the host defers the nn.Linear ones, swin_unet_v2.py:205-240) -- */
long long uz_wgrad_multi_workspace_bytes_ref(const uz_wgrad_item* items, int n) {
  (void)items, (void)n;
  return 0;
}
UZ_SAME_SIGNATURE(uz_wgrad_multi_workspace_bytes);

int uz_wgrad_multi_ref(const uz_wgrad_item* items, int n, void* workspace, void* stream) {
  for (int i = 0; i < n; ++i) {
    const int rc = uz_wgrad_ref(&items[i].desc, items[i].L, items[i].R, items[i].out, workspace, stream);
    if (rc != UZ_OK) return rc;
  }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_wgrad_multi);

/* ---- uz_winattn_*: WindowAttention core with roll / window_partition / window_reverse as index arithmetic
 * (swin_unet_v2.py:127-159 with :30-56, :214-238, :246-262) ----------------------------------------------------------- */
typedef struct {
  long long tok; /* row of the token tensor */
  int cnt;       /* region id of the shifted-window mask (:214-236) */
} wa_tok;
static wa_tok wa_token(const uz_winattn_desc* d, int win, int i) {
  const int nwx = d->W / d->ws, nwy = d->H / d->ws, nW = nwx * nwy;
  const int b = win / nW, wi = win % nW, wy = wi / nwx, wx = wi % nwx;
  const int hs = wy * d->ws + i / d->ws, wsx = wx * d->ws + i % d->ws; /* coordinates in the rolled image */
  const int h = (hs + d->shift) % d->H, w = (wsx + d->shift) % d->W;
  wa_tok t;
  t.tok = ((long long)b * d->H + h) * d->W + w;
  const int hid = hs < d->H - d->ws ? 0 : (hs < d->H - d->shift ? 1 : 2);
  const int wid = wsx < d->W - d->ws ? 0 : (wsx < d->W - d->shift ? 1 : 2);
  t.cnt = d->shift > 0 ? hid * 3 + wid : 0;
  return t;
}
/* scores of one (window, head): c = cosine, S = c / clip(tau) + bias (+ mask); P = softmax(S); returns nothing, fills arrays */
static void wa_scores(const uz_winattn_desc* d, const void* qkv, const float* tau, const float* bias, int win, int h, const wa_tok* tk,
                      double* c, double* P, double* lse, int* clamped) {
  const int N = d->ws * d->ws;
  for (int i = 0; i < N; ++i) {
    double nq = 0.0, mx = -1e300;
    for (int e = 0; e < 32; ++e) {
      const double q = d->scale * ld(d->dtype, qkv, tk[i].tok * d->ldq + h * 32 + e);
      nq += q * q;
    }
    nq = sqrt(nq);
    for (int j = 0; j < N; ++j) {
      double u = 0.0, nk = 0.0;
      for (int e = 0; e < 32; ++e) {
        const double q = d->scale * ld(d->dtype, qkv, tk[i].tok * d->ldq + h * 32 + e);
        const double k = ld(d->dtype, qkv, tk[j].tok * d->ldq + d->C + h * 32 + e);
        u += q * k;
        nk += k * k;
      }
      nk = sqrt(nk);
      const double den = nq * nk > 1e-6 ? nq * nk : 1e-6;
      clamped[i * N + j] = !(nq * nk > 1e-6);
      c[i * N + j] = u / den;
      const double tv = tau[((long long)h * d->Nt + i) * d->Nt + j];
      double sv = c[i * N + j] / (tv > 0.01 ? tv : 0.01) + bias[((long long)h * N + i) * N + j];
      if (tk[i].cnt != tk[j].cnt) sv -= 100.0;
      P[i * N + j] = sv;
      if (sv > mx) mx = sv;
    }
    double sum = 0.0;
    for (int j = 0; j < N; ++j) sum += exp(P[i * N + j] - mx);
    lse[i] = mx + log(sum);
    for (int j = 0; j < N; ++j) P[i * N + j] = exp(P[i * N + j] - lse[i]);
  }
  (void)win;
}

int uz_winattn_fwd_ref(const uz_winattn_desc* d, const void* qkv, const float* tau, const float* bias, void* out, float* lse,
                       void* stream) {
  (void)stream;
  const int N = d->ws * d->ws, nWin = d->B * (d->H / d->ws) * (d->W / d->ws);
  if (d->C != 32 * d->heads || N > 64) return UZ_ENOTIMPL;
  double *c = malloc(sizeof(double) * N * N), *P = malloc(sizeof(double) * N * N), *l = malloc(sizeof(double) * N);
  int* cl = malloc(sizeof(int) * N * N);
  wa_tok* tk = malloc(sizeof(wa_tok) * N);
  for (int win = 0; win < nWin; ++win) {
    for (int i = 0; i < N; ++i) tk[i] = wa_token(d, win, i);
    for (int h = 0; h < d->heads; ++h) {
      wa_scores(d, qkv, tau, bias, win, h, tk, c, P, l, cl);
      for (int i = 0; i < N; ++i) {
        lse[((long long)win * d->heads + h) * N + i] = (float)l[i];
        for (int e = 0; e < 32; ++e) {
          double o = 0.0;
          for (int j = 0; j < N; ++j) o += P[i * N + j] * ld(d->dtype, qkv, tk[j].tok * d->ldq + 2 * d->C + h * 32 + e);
          st(d->dtype, out, tk[i].tok * d->ldo + h * 32 + e, o);
        }
      }
    }
  }
  free(c), free(P), free(l), free(cl), free(tk);
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_winattn_fwd);

int uz_winattn_bwd_rows_ref(const uz_winattn_desc* d) { /* the restatement keeps its sums in one row */
  (void)d;
  return 1;
}
UZ_SAME_SIGNATURE(uz_winattn_bwd_rows);

/* gradients of the above by the chain rule, scores recomputed (the forward's out / lse are not read); partial[0][2][heads][N][N] */
int uz_winattn_bwd_ref(const uz_winattn_desc* d, const void* qkv, const float* tau, const float* bias, const void* out,
                       const float* lse_in, const void* dout, int lddo, void* dqkv, int lddq, float* partial, void* stream) {
  (void)out, (void)lse_in, (void)stream;
  const int N = d->ws * d->ws, nWin = d->B * (d->H / d->ws) * (d->W / d->ws);
  if (d->C != 32 * d->heads || N > 64) return UZ_ENOTIMPL;
  double *c = malloc(sizeof(double) * N * N), *P = malloc(sizeof(double) * N * N), *l = malloc(sizeof(double) * N);
  double *dc = malloc(sizeof(double) * N * N), *acc = calloc((size_t)2 * d->heads * N * N, sizeof(double));
  int* cl = malloc(sizeof(int) * N * N);
  wa_tok* tk = malloc(sizeof(wa_tok) * N);
  for (int win = 0; win < nWin; ++win) {
    for (int i = 0; i < N; ++i) tk[i] = wa_token(d, win, i);
    for (int h = 0; h < d->heads; ++h) {
      wa_scores(d, qkv, tau, bias, win, h, tk, c, P, l, cl);
      double* ab = acc + (size_t)h * N * N;
      double* at = acc + ((size_t)d->heads + h) * N * N;
      for (int i = 0; i < N; ++i) {
        double D = 0.0;
        for (int j = 0; j < N; ++j) {
          double dp = 0.0;
          for (int e = 0; e < 32; ++e)
            dp += ld(d->dtype, dout, tk[i].tok * lddo + h * 32 + e) * ld(d->dtype, qkv, tk[j].tok * d->ldq + 2 * d->C + h * 32 + e);
          dc[i * N + j] = dp;
          D += P[i * N + j] * dp;
        }
        for (int j = 0; j < N; ++j) {
          const double ds = P[i * N + j] * (dc[i * N + j] - D);
          const double tv = tau[((long long)h * d->Nt + i) * d->Nt + j], t = tv > 0.01 ? tv : 0.01;
          ab[i * N + j] += ds;
          if (tv >= 0.01) at[i * N + j] += -ds * c[i * N + j] / (t * t);
          dc[i * N + j] = ds / t;
        }
      }
      for (int j = 0; j < N; ++j) /* dv_j = sum_i P_ij dO_i */
        for (int e = 0; e < 32; ++e) {
          double v = 0.0;
          for (int i = 0; i < N; ++i) v += P[i * N + j] * ld(d->dtype, dout, tk[i].tok * lddo + h * 32 + e);
          st(d->dtype, dqkv, tk[j].tok * lddq + 2 * d->C + h * 32 + e, v);
        }
      for (int i = 0; i < N; ++i) { /* dq_i = scale sum_j dc_ij d c_ij / d(scale q_i) */
        double nq2 = 0.0, g[32];
        for (int e = 0; e < 32; ++e) {
          const double q = d->scale * ld(d->dtype, qkv, tk[i].tok * d->ldq + h * 32 + e);
          nq2 += q * q;
          g[e] = 0.0;
        }
        for (int j = 0; j < N; ++j) {
          double nk2 = 0.0;
          for (int e = 0; e < 32; ++e) {
            const double k = ld(d->dtype, qkv, tk[j].tok * d->ldq + d->C + h * 32 + e);
            nk2 += k * k;
          }
          const double den = cl[i * N + j] ? 1e-6 : sqrt(nq2 * nk2);
          for (int e = 0; e < 32; ++e) {
            const double q = d->scale * ld(d->dtype, qkv, tk[i].tok * d->ldq + h * 32 + e);
            const double k = ld(d->dtype, qkv, tk[j].tok * d->ldq + d->C + h * 32 + e);
            g[e] += dc[i * N + j] * (k / den - (cl[i * N + j] ? 0.0 : c[i * N + j] * q / nq2));
          }
        }
        for (int e = 0; e < 32; ++e) st(d->dtype, dqkv, tk[i].tok * lddq + h * 32 + e, d->scale * g[e]);
      }
      for (int j = 0; j < N; ++j) { /* dk_j = sum_i dc_ij d c_ij / d k_j */
        double nk2 = 0.0, g[32];
        for (int e = 0; e < 32; ++e) {
          const double k = ld(d->dtype, qkv, tk[j].tok * d->ldq + d->C + h * 32 + e);
          nk2 += k * k;
          g[e] = 0.0;
        }
        for (int i = 0; i < N; ++i) {
          double nq2 = 0.0;
          for (int e = 0; e < 32; ++e) {
            const double q = d->scale * ld(d->dtype, qkv, tk[i].tok * d->ldq + h * 32 + e);
            nq2 += q * q;
          }
          const double den = cl[i * N + j] ? 1e-6 : sqrt(nq2 * nk2);
          for (int e = 0; e < 32; ++e) {
            const double q = d->scale * ld(d->dtype, qkv, tk[i].tok * d->ldq + h * 32 + e);
            const double k = ld(d->dtype, qkv, tk[j].tok * d->ldq + d->C + h * 32 + e);
            g[e] += dc[i * N + j] * (q / den - (cl[i * N + j] ? 0.0 : c[i * N + j] * k / nk2));
          }
        }
        for (int e = 0; e < 32; ++e) st(d->dtype, dqkv, tk[j].tok * lddq + d->C + h * 32 + e, g[e]);
      }
    }
  }
  for (size_t i = 0; i < (size_t)2 * d->heads * N * N; ++i) partial[i] = (float)acc[i];
  free(c), free(P), free(l), free(dc), free(acc), free(cl), free(tk);
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_winattn_bwd);

/* ---- additive attention gate, forward (AttentionBlock.forward, attention_unet.py:34-40) without its two 1x1 convolutions ---- */
int uz_attn_grid_ref(int dtype, int P, int channels) { /* the restatement keeps its sums in one row */
  (void)dtype, (void)P, (void)channels;
  return 1;
}
UZ_SAME_SIGNATURE(uz_attn_grid);

/* q[p] = b_psi + sum_c relu(bn_g(g1raw) + bn_x(x1raw))[p, c] w_psi[c]; partial[0] = (sum q, sum q^2): the statistics of bn_q */
int uz_attn_psi_fwd_ref(int dtype, const void* g1raw, int ldg, const void* x1raw, int ldx, const float* vec_g, const float* vec_x,
                        const float* wpsi, const float* bpsi, int P, int F, float* q, float* partial, void* stream) {
  (void)stream;
  double s1 = 0.0, s2 = 0.0;
  for (long long p = 0; p < P; ++p) {
    double acc = bpsi ? (double)bpsi[0] : 0.0;
    for (int c = 0; c < F; ++c) {
      const double a = ld(dtype, g1raw, p * ldg + c) * vec_g[c] + vec_g[F + c] + ld(dtype, x1raw, p * ldx + c) * vec_x[c] + vec_x[F + c];
      if (a > 0.0) acc += a * wpsi[c];
    }
    q[p] = (float)acc;
    s1 += (double)q[p];
    s2 += (double)q[p] * (double)q[p];
  }
  if (partial) {
    partial[0] = (float)s1;
    partial[1] = (float)s2;
  }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_attn_psi_fwd);

/* out = x * sigmoid(bn_q(q)) */
int uz_attn_gate_fwd_ref(int dtype, const void* x, int ldx, const float* q, const float* vec_q, int P, int C, void* out, int ldo,
                         void* stream) {
  (void)stream;
  for (long long p = 0; p < P; ++p) {
    const double z = (double)q[p] * vec_q[0] + vec_q[1], sg = 1.0 / (1.0 + exp(-z));
    for (int c = 0; c < C; ++c) st(dtype, out, p * ldo + c, ld(dtype, x, p * ldx + c) * sg);
  }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_attn_gate_fwd);


/* ======================================================================================================================
 * Round 5: the input-transform forms, the attention gate's backward, the resize backward, spatial-reduction attention
 * and the element / bookkeeping passes that had no restatement yet.
 * ==================================================================================================================== */

/* a[p][c] = relu(x * scale + shift) rounded to the tensor type: ONE fp32 fma and one rounding, as uz_bn_relu_apply stores it
 * (common_layers.py:29-30: BatchNorm2d folded into scale / shift, then ReLU) */
static void xf_apply(int dtype, const void* x, int ldx, long long P, int C, const float* scale, const float* shift, void* a) {
  for (long long p = 0; p < P; ++p)
    for (int c = 0; c < C; ++c) {
      const float z = fmaf((float)ld(dtype, x, p * ldx + c), scale[c], shift[c]);
      st(dtype, a, p * C + c, z > 0.f ? (double)z : 0.0);
    }
}

/* the second convolution of a DoubleConv reading the first one's RAW output through its BatchNorm + ReLU
 * (common_layers.py:28-33): exactly uz_bn_relu_apply followed by uz_conv_igemm, the middle tensor private to the call */
int uz_conv_igemm_xf_ref(const uz_conv_desc* d, const void* x, const float* in_scale, const float* in_shift, const void* w_packed,
                         const float* bias, void* y, float* stats_partial, void* stream) {
  const long long Pin = (long long)d->N * d->Hin * d->Win;
  const size_t es = d->dtype == UZ_BF16 ? 2 : 4;
  void* a = malloc((size_t)Pin * d->Cin * es);
  if (!a) return UZ_EINVAL;
  xf_apply(d->dtype, x, d->ldx, Pin, d->Cin, in_scale, in_shift, a);
  uz_conv_desc d2 = *d;
  d2.ldx = d->Cin;
  const int rc = uz_conv_igemm_ref(&d2, a, w_packed, bias, y, stats_partial, stream);
  free(a);
  return rc;
}
UZ_SAME_SIGNATURE(uz_conv_igemm_xf);

/* ... and that convolution's weight gradient (autograd, training_loop.py:119) with R = the raw output */
int uz_wgrad_xf_ref(const uz_wgrad_desc* d, const void* L, const void* R, const float* r_scale, const float* r_shift, float* out,
                    void* workspace, void* stream, int phase) {
  (void)phase;
  const long long Pr = (long long)d->N * d->Hr * d->Wr;
  const size_t es = d->dtype == UZ_BF16 ? 2 : 4;
  void* a = malloc((size_t)Pr * d->Cj * es);
  if (!a) return UZ_EINVAL;
  xf_apply(d->dtype, R, d->ldr, Pr, d->Cj, r_scale, r_shift, a);
  uz_wgrad_desc d2 = *d;
  d2.ldr = d->Cj;
  const int rc = uz_wgrad_ref(&d2, L, a, out, workspace, stream);
  free(a);
  return rc;
}
UZ_SAME_SIGNATURE(uz_wgrad_xf);

/* act = relu(scale y + shift) + res, pooled = MaxPool2d(2, 2[, ceil_mode]) of the STORED act (u2net.py:74, :30, :221-229);
 * bit 1 of pool_ceil: BatchNorm without the ReLU (resunet's skip branch) */
int uz_bn_relu_add_apply_ref(int dtype, const void* y, int ldy, const float* scale, const float* shift, int N, int H, int W, int C,
                             const void* res, int ldr, void* act, int lda, void* pooled, int ldp, int pool_ceil, void* stream) {
  (void)stream;
  const int relu = !(pool_ceil & 2), ceil_mode = pool_ceil & 1;
  for (long long p = 0; p < (long long)N * H * W; ++p)
    for (int c = 0; c < C; ++c) {
      float v = fmaf((float)ld(dtype, y, p * ldy + c), scale[c], shift[c]);
      if (relu && !(v > 0.f)) v = 0.f;
      if (res) v += (float)ld(dtype, res, p * ldr + c);
      st(dtype, act, p * lda + c, (double)v);
    }
  if (pooled) {
    const int Hp = ceil_mode ? (H + 1) / 2 : H / 2, Wp = ceil_mode ? (W + 1) / 2 : W / 2;
    for (int n = 0; n < N; ++n)
      for (int h = 0; h < Hp; ++h)
        for (int w = 0; w < Wp; ++w)
          for (int c = 0; c < C; ++c) {
            double m = -INFINITY;
            for (int a = 0; a < 2; ++a)
              for (int b = 0; b < 2; ++b) {
                if (2 * h + a >= H || 2 * w + b >= W) continue;
                const double v = ld(dtype, act, (((long long)n * H + 2 * h + a) * W + 2 * w + b) * lda + c);
                if (v > m) m = v;
              }
            st(dtype, pooled, (((long long)n * Hp + h) * Wp + w) * ldp + c, m);
          }
  }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_bn_relu_add_apply);

/* out = g0 + g1 + unpool(gp): the pooled gradient goes to the FIRST maximum of its 2x2 window of act in raster order, the
 * element ATen's max_pool2d records (common_layers.py:90 under autograd) */
int uz_pool_grad_combine_ref(int dtype, int N, int H, int W, int C, const void* act, int lda, const void* g0, int ldg0, const void* g1,
                             int ldg1, const void* gp, int ldgp, void* out, int ldo, int pool_ceil, void* stream) {
  (void)stream;
  const int Hp = (pool_ceil & 1) ? (H + 1) / 2 : H / 2, Wp = (pool_ceil & 1) ? (W + 1) / 2 : W / 2;
  for (int n = 0; n < N; ++n)
    for (int h = 0; h < H; ++h)
      for (int w = 0; w < W; ++w)
        for (int c = 0; c < C; ++c) {
          const long long p = ((long long)n * H + h) * W + w;
          double g = 0.0;
          if (g0) g += ld(dtype, g0, p * ldg0 + c);
          if (g1) g += ld(dtype, g1, p * ldg1 + c);
          const int ph = h / 2, pw = w / 2;
          if (gp && ph < Hp && pw < Wp) {
            double best = -INFINITY;
            int bh = -1, bw = -1;
            for (int a = 0; a < 2; ++a)
              for (int b = 0; b < 2; ++b) {
                const int hh = 2 * ph + a, ww = 2 * pw + b;
                if (hh >= H || ww >= W) continue;
                const double v = ld(dtype, act, (((long long)n * H + hh) * W + ww) * lda + c);
                if (v > best) best = v, bh = hh, bw = ww;
              }
            if (bh == h && bw == w) g += ld(dtype, gp, (((long long)n * Hp + ph) * Wp + pw) * ldgp + c);
          }
          st(dtype, out, p * ldo + c, g);
        }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_pool_grad_combine);

/* backward of F.interpolate(mode='bilinear') (u2net.py:19-22, nested_unet.py:32): every output pixel hands its gradient to its
 * four sources with the forward's weights; sums in double, one rounding */
int uz_resize_bilinear_bwd_ref(int dtype, const void* g, int ldg, long long g_img_stride, int N, int Hi, int Wi, int C, void* dx,
                               int lddx, long long dx_img_stride, int Ho, int Wo, int align_corners, void* stream) {
  (void)stream;
  const double sh = align_corners ? (Ho > 1 ? (double)(Hi - 1) / (Ho - 1) : 0.0) : (double)Hi / Ho;
  const double sw = align_corners ? (Wo > 1 ? (double)(Wi - 1) / (Wo - 1) : 0.0) : (double)Wi / Wo;
  double* acc = (double*)calloc((size_t)Hi * Wi * C, sizeof(double));
  if (!acc) return UZ_EINVAL;
  for (int n = 0; n < N; ++n) {
    memset(acc, 0, (size_t)Hi * Wi * C * sizeof(double));
    for (int oh = 0; oh < Ho; ++oh)
      for (int ow = 0; ow < Wo; ++ow) {
        double fh = align_corners ? oh * sh : (oh + 0.5) * sh - 0.5, fw = align_corners ? ow * sw : (ow + 0.5) * sw - 0.5;
        if (fh < 0.0) fh = 0.0;
        if (fw < 0.0) fw = 0.0;
        const int h0 = (int)fh < Hi - 1 ? (int)fh : Hi - 1, w0 = (int)fw < Wi - 1 ? (int)fw : Wi - 1;
        const int h1 = h0 < Hi - 1 ? h0 + 1 : h0, w1 = w0 < Wi - 1 ? w0 + 1 : w0;
        const double lh = fh - h0, lw = fw - w0;
        for (int c = 0; c < C; ++c) {
          const double gv = ld(dtype, g, n * g_img_stride + ((long long)oh * Wo + ow) * ldg + c);
          acc[((size_t)h0 * Wi + w0) * C + c] += (1 - lh) * (1 - lw) * gv;
          acc[((size_t)h0 * Wi + w1) * C + c] += (1 - lh) * lw * gv;
          acc[((size_t)h1 * Wi + w0) * C + c] += lh * (1 - lw) * gv;
          acc[((size_t)h1 * Wi + w1) * C + c] += lh * lw * gv;
        }
      }
    for (long long q = 0; q < (long long)Hi * Wi; ++q)
      for (int c = 0; c < C; ++c) st(dtype, dx, n * dx_img_stride + q * lddx + c, acc[(size_t)q * C + c]);
  }
  free(acc);
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_resize_bilinear_bwd);

int uz_bilinear_bwd_ref(int dtype, const void* g, int ldg, long long g_img_stride, int N, int Hi, int Wi, int C, void* dx, int lddx,
                        long long dx_img_stride, int Ho, int Wo, void* stream) {
  return uz_resize_bilinear_bwd_ref(dtype, g, ldg, g_img_stride, N, Hi, Wi, C, dx, lddx, dx_img_stride, Ho, Wo, 0, stream);
}
UZ_SAME_SIGNATURE(uz_bilinear_bwd);

/* nn.BCEWithLogitsLoss (mean) + its gradient + dice_coefficient of the thresholded prediction (scripts/train.py:135,
 * utils/metrics.py:7-24: sigmoid > 0.5 i.e. logit > 0, epsilon 1e-7, 1.0 for an empty union) */
long long uz_bce_dice_workspace_bytes_ref(long long n) {
  (void)n;
  return 64;
}
UZ_SAME_SIGNATURE(uz_bce_dice_workspace_bytes);

int uz_bce_dice_ref(const float* logits, const float* target, long long n, float* dlogits, float* out2, void* workspace, void* stream) {
  (void)workspace, (void)stream;
  double loss = 0.0, inter = 0.0, sp = 0.0, stt = 0.0;
  for (long long i = 0; i < n; ++i) {
    const double x = logits[i], t = target[i];
    loss += (x > 0.0 ? x : 0.0) - x * t + log1p(exp(-fabs(x)));
    const double sg = 1.0 / (1.0 + exp(-x));
    if (dlogits) dlogits[i] = (float)((sg - t) / (double)n);
    const double pr = x > 0.0 ? 1.0 : 0.0;
    inter += pr * t;
    sp += pr;
    stt += t;
  }
  out2[0] = (float)(loss / (double)n);
  out2[1] = (sp + stt == 0.0) ? 1.0f : (float)((2.0 * inter + 1e-7) / (sp + stt + 1e-7));
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_bce_dice);

/* nn.Dropout(p) in training mode on the caller's uniform draw (uctransnet.py:54; swin_unet_v2.py:158) */
int uz_dropout_ref(int dtype, const void* x, int ldx, const float* u, float p, void* out, int ldo, long long P, int C, void* stream) {
  (void)stream;
  const float inv = 1.0f / (1.0f - p);
  for (long long r = 0; r < P; ++r)
    for (int c = 0; c < C; ++c) st(dtype, out, r * ldo + c, u[r * C + c] >= p ? (double)((float)ld(dtype, x, r * ldx + c) * inv) : 0.0);
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_dropout);

/* the gate of UCTransNet's CCA (uctransnet.py:417-427) and its two gradient forms */
int uz_chanscale_relu_ref(int dtype, int mode, const void* g, int ldg, const void* x, int ldx, const float* s, const float* a, int N,
                          int HW, int C, void* out, int ldo, void* stream) {
  (void)stream;
  for (int n = 0; n < N; ++n)
    for (long long q = 0; q < HW; ++q)
      for (int c = 0; c < C; ++c) {
        const long long p = (long long)n * HW + q;
        const double xv = ld(dtype, x, p * ldx + c), sv = s[(long long)n * C + c];
        double v;
        if (mode == 2) v = xv * sv > 0.0 ? (double)((float)xv * (float)sv) : 0.0;
        else {
          const double gv = xv > 0.0 ? ld(dtype, g, p * ldg + c) : 0.0;
          v = mode == 0 ? gv * xv : gv * sv + (a ? (double)a[(long long)n * C + c] : 0.0);
        }
        st(dtype, out, p * ldo + c, v);
      }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_chanscale_relu);

/* PatchEmbed's Conv2d(kernel = stride = patch) input as GEMM rows (swin_unet_v2.py:548-556) */
int uz_patchify_ref(int dtype, const float* x_nchw, int N, int C, int H, int W, int patch, int Kpad, void* out, void* stream) {
  (void)stream;
  const int Hp = H / patch, Wp = W / patch;
  for (int n = 0; n < N; ++n)
    for (int i = 0; i < Hp; ++i)
      for (int j = 0; j < Wp; ++j) {
        const long long row = ((long long)n * Hp + i) * Wp + j;
        for (int k = 0; k < Kpad; ++k) {
          double v = 0.0;
          if (k < patch * patch * C) {
            const int c = k % C, kw = (k / C) % patch, kh = k / C / patch;
            v = x_nchw[(((long long)n * C + c) * H + i * patch + kh) * W + j * patch + kw];
          }
          st(dtype, out, row * Kpad + k, v);
        }
      }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_patchify);

/* im2col of the NCHW fp32 network input for its first 3x3 convolution (unet.py:31): dst[p][t C + c], zero padded */
int uz_im2col3x3_nchw_ref(int dtype, const float* x_nchw, int N, int C, int H, int W, int Kpad, void* dst, void* stream) {
  (void)stream;
  for (int n = 0; n < N; ++n)
    for (int h = 0; h < H; ++h)
      for (int w = 0; w < W; ++w) {
        const long long p = ((long long)n * H + h) * W + w;
        for (int k = 0; k < Kpad; ++k) {
          double v = 0.0;
          if (k < 9 * C) {
            const int t = k / C, c = k % C, hh = h + t / 3 - 1, ww = w + t % 3 - 1;
            if (hh >= 0 && hh < H && ww >= 0 && ww < W) v = x_nchw[(((long long)n * C + c) * H + hh) * W + ww];
          }
          st(dtype, dst, p * Kpad + k, v);
        }
      }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_im2col3x3_nchw);

/* fixed-order row sums written where they are wanted (weight | bias gradient halves of a partial row) */
int uz_sum_rows_f32_ld_ref(const float* partial, int ldp, int rows, int n, float* out0, int n0, float* out1, void* stream) {
  (void)stream;
  for (int e = 0; e < n; ++e) {
    double t = 0.0;
    for (int r = 0; r < rows; ++r) t += (double)partial[(long long)r * ldp + e];
    if (e < n0) out0[e] = (float)t;
    else out1[e - n0] = (float)t;
  }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_sum_rows_f32_ld);

int uz_sum_rows_f32_ref(const float* partial, int rows, int n, float* out0, int n0, float* out1, void* stream) {
  return uz_sum_rows_f32_ld_ref(partial, n, rows, n, out0, n0, out1, stream);
}
UZ_SAME_SIGNATURE(uz_sum_rows_f32);

/* pixel-grid moves: copy into a concat slot / every second pixel (what a stride-2 convolution reads, common_layers.py:188) /
 * that selection's gradient */
int uz_resample2_ref(int dtype, const void* src, int lds, int N, int Hs, int Ws, int C, void* dst, int ldd, int Hd, int Wd, int mode,
                     void* stream) {
  (void)stream;
  for (int n = 0; n < N; ++n)
    for (int h = 0; h < Hd; ++h)
      for (int w = 0; w < Wd; ++w)
        for (int c = 0; c < C; ++c) {
          double v = 0.0;
          if (mode == 0) v = ld(dtype, src, (((long long)n * Hs + h) * Ws + w) * lds + c);
          else if (mode == 1) v = ld(dtype, src, (((long long)n * Hs + 2 * h) * Ws + 2 * w) * lds + c);
          else if (!(h & 1) && !(w & 1)) v = ld(dtype, src, (((long long)n * Hs + h / 2) * Ws + w / 2) * lds + c);
          st(dtype, dst, (((long long)n * Hd + h) * Wd + w) * ldd + c, v);
        }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_resample2);

/* ---- attention gate backward (attention_unet.py:34-40 under autograd; uz_attn.hip) -------------------------------------- */
/* dx_direct = dOut psi;  dZ[p] = (sum_c dOut x) psi (1 - psi);  partial[0] = (sum dZ, sum dZ qhat) */
int uz_attn_bwd_psi_ref(int dtype, const void* dout, int ldd, const void* x, int ldx, const float* q, const float* vec_q, int P, int C,
                        void* dx_direct, int lddx, float* dz, float* partial, void* stream) {
  (void)stream;
  double a0 = 0.0, a1 = 0.0;
  for (long long p = 0; p < P; ++p) {
    const double z = (double)q[p] * vec_q[0] + vec_q[1], psi = 1.0 / (1.0 + exp(-z));
    double dot = 0.0;
    for (int c = 0; c < C; ++c) {
      const double d = ld(dtype, dout, p * ldd + c);
      dot += d * ld(dtype, x, p * ldx + c);
      st(dtype, dx_direct, p * lddx + c, d * psi);
    }
    dz[p] = (float)(dot * psi * (1.0 - psi));
    a0 += (double)dz[p];
    a1 += (double)dz[p] * ((double)q[p] - vec_q[2]) * vec_q[3];
  }
  partial[0] = (float)a0;
  partial[1] = (float)a1;
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_attn_bwd_psi);

/* dq = s_q (dZ - a0/P - qhat a1/P) (BatchNorm backward of the 1-channel psi map);  dPre = dq w_psi [G1 + X1 > 0];
 * partial[0] = [B0 | B1 | D1 | W | sum dq]: sums of dPre, dPre ghat, dPre xhat, dq relu(G1 + X1) per channel */
static double gate_dq(const float* q, const float* dz, const float* vec_q, const double* a01, int P, long long p) {
  const double k0 = a01[0] / P, k1 = a01[1] / P;
  return (double)vec_q[0] * ((double)dz[p] - k0 - ((double)q[p] - vec_q[2]) * vec_q[3] * k1);
}

int uz_attn_bwd_reduce_ref(int dtype, const void* g1raw, int ldg, const void* x1raw, int ldx, const float* q, const float* dz,
                           const float* wpsi, const float* vec_g, const float* vec_x, const float* vec_q, const double* a01, int P,
                           int F, float* partial, void* stream) {
  (void)stream;
  double* acc = (double*)calloc((size_t)4 * F + 1, sizeof(double));
  if (!acc) return UZ_EINVAL;
  for (long long p = 0; p < P; ++p) {
    const double dq = gate_dq(q, dz, vec_q, a01, P, p);
    for (int c = 0; c < F; ++c) {
      const double gv = ld(dtype, g1raw, p * ldg + c), xv = ld(dtype, x1raw, p * ldx + c);
      const double s = gv * vec_g[c] + vec_g[F + c] + xv * vec_x[c] + vec_x[F + c];
      const double dpre = s > 0.0 ? dq * wpsi[c] : 0.0;
      acc[c] += dpre;
      acc[F + c] += dpre * (gv - vec_g[2 * F + c]) * vec_g[3 * F + c];
      acc[2 * F + c] += dpre * (xv - vec_x[2 * F + c]) * vec_x[3 * F + c];
      acc[3 * F + c] += dq * (s > 0.0 ? s : 0.0);
    }
    acc[4 * F] += dq;
  }
  for (int e = 0; e < 4 * F + 1; ++e) partial[e] = (float)acc[e];
  free(acc);
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_attn_bwd_reduce);

/* dg1raw = s_g (dPre - B0/P - ghat B1/P),  dx1raw = s_x (dPre - B0/P - xhat D1/P): the two BatchNorm backwards */
int uz_attn_bwd_apply_ref(int dtype, const void* g1raw, int ldg, const void* x1raw, int ldx, const float* q, const float* dz,
                          const float* wpsi, const float* vec_g, const float* vec_x, const float* vec_q, const double* a01,
                          const double* totals, int P, int F, void* dg1raw, int lddg, void* dx1raw, int lddx, void* stream) {
  (void)stream;
  for (long long p = 0; p < P; ++p) {
    const double dq = gate_dq(q, dz, vec_q, a01, P, p);
    for (int c = 0; c < F; ++c) {
      const double gv = ld(dtype, g1raw, p * ldg + c), xv = ld(dtype, x1raw, p * ldx + c);
      const double s = gv * vec_g[c] + vec_g[F + c] + xv * vec_x[c] + vec_x[F + c];
      const double dpre = s > 0.0 ? dq * wpsi[c] : 0.0;
      const double gh = (gv - vec_g[2 * F + c]) * vec_g[3 * F + c], xh = (xv - vec_x[2 * F + c]) * vec_x[3 * F + c];
      st(dtype, dg1raw, p * lddg + c, vec_g[c] * (dpre - totals[c] / P - gh * totals[F + c] / P));
      st(dtype, dx1raw, p * lddx + c, vec_x[c] * (dpre - totals[c] / P - xh * totals[2 * F + c] / P));
    }
  }
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_attn_bwd_apply);

/* ---- spatial-reduction attention (EfficientSelfAtten, missformer.py:21-39, :113-128): softmax(q k^T scale) v per (image, head),
 * key j of image b at row ((j / kps) B + b) kps + j % kps ----------------------------------------------------------------- */
static long long sra_krow(const uz_sra_desc* d, int b, int j) { return ((long long)(j / d->kps) * d->B + b) * d->kps + j % d->kps; }

int uz_sra_fwd_ref(const uz_sra_desc* d, const void* q, const void* k, const void* v, void* out, float* lse, void* stream) {
  (void)stream;
  const int D = d->head_dim;
  double* s = (double*)malloc((size_t)d->NK * sizeof(double));
  if (!s) return UZ_EINVAL;
  for (int b = 0; b < d->B; ++b)
    for (int h = 0; h < d->heads; ++h)
      for (int i = 0; i < d->N; ++i) {
        const long long qr = (long long)b * d->N + i;
        double m = -INFINITY;
        for (int j = 0; j < d->NK; ++j) {
          double acc = 0.0;
          for (int e = 0; e < D; ++e) acc += ld(d->dtype, q, qr * d->ldq + h * D + e) * ld(d->dtype, k, sra_krow(d, b, j) * d->ldk + h * D + e);
          s[j] = acc * d->scale;
          if (s[j] > m) m = s[j];
        }
        double z = 0.0;
        for (int j = 0; j < d->NK; ++j) z += exp(s[j] - m);
        if (lse) lse[((long long)b * d->heads + h) * d->N + i] = (float)((m + log(z)) * 1.4426950408889634);   /* log2 units, as the kernels keep it */
        for (int e = 0; e < D; ++e) {
          double acc = 0.0;
          for (int j = 0; j < d->NK; ++j) acc += exp(s[j] - m) / z * ld(d->dtype, v, sra_krow(d, b, j) * d->ldv + h * D + e);
          st(d->dtype, out, qr * d->ldo + h * D + e, acc);
        }
      }
  free(s);
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_sra_fwd);

long long uz_sra_bwd_workspace_bytes_ref(const uz_sra_desc* d) {
  (void)d;
  return 64;
}
UZ_SAME_SIGNATURE(uz_sra_bwd_workspace_bytes);

/* dq = scale (P o (dP - delta)) K, dK = scale (P o (dP - delta))^T Q, dV = P^T dO, delta_i = dO_i . O_i (softmax backward with P
 * recomputed from the kept log-sum-exp); dkv rows laid out like the kv tensor: [dK | dV] */
int uz_sra_bwd_ref(const uz_sra_desc* d, const void* q, const void* k, const void* v, const void* o, const float* lse, const void* go,
                   int ldgo, void* dq, int lddq, void* dkv, int lddkv, void* workspace, void* stream) {
  (void)workspace, (void)stream;
  const int D = d->head_dim, HD = d->heads * D;
  const long long nkv = (long long)d->B * d->NK;
  double* dK = (double*)calloc((size_t)nkv * HD * 2, sizeof(double));
  if (!dK) return UZ_EINVAL;
  double* dV = dK + (size_t)nkv * HD;
  double* dqa = (double*)malloc((size_t)D * sizeof(double));
  for (int b = 0; b < d->B; ++b)
    for (int h = 0; h < d->heads; ++h)
      for (int i = 0; i < d->N; ++i) {
        const long long qr = (long long)b * d->N + i;
        const double l = lse[((long long)b * d->heads + h) * d->N + i] * 0.6931471805599453;
        double delta = 0.0;
        for (int e = 0; e < D; ++e) delta += ld(d->dtype, go, qr * ldgo + h * D + e) * ld(d->dtype, o, qr * d->ldo + h * D + e);
        for (int e = 0; e < D; ++e) dqa[e] = 0.0;
        for (int j = 0; j < d->NK; ++j) {
          const long long kr = sra_krow(d, b, j);
          double sc = 0.0, dp = 0.0;
          for (int e = 0; e < D; ++e) {
            sc += ld(d->dtype, q, qr * d->ldq + h * D + e) * ld(d->dtype, k, kr * d->ldk + h * D + e);
            dp += ld(d->dtype, go, qr * ldgo + h * D + e) * ld(d->dtype, v, kr * d->ldv + h * D + e);
          }
          const double pr = exp(sc * d->scale - l), ds = pr * (dp - delta) * d->scale;
          for (int e = 0; e < D; ++e) {
            dqa[e] += ds * ld(d->dtype, k, kr * d->ldk + h * D + e);
            dK[(size_t)kr * HD + h * D + e] += ds * ld(d->dtype, q, qr * d->ldq + h * D + e);
            dV[(size_t)kr * HD + h * D + e] += pr * ld(d->dtype, go, qr * ldgo + h * D + e);
          }
        }
        for (int e = 0; e < D; ++e) st(d->dtype, dq, qr * lddq + h * D + e, dqa[e]);
      }
  for (long long r = 0; r < nkv; ++r)
    for (int c = 0; c < HD; ++c) {
      st(d->dtype, dkv, r * lddkv + c, dK[(size_t)r * HD + c]);
      st(d->dtype, dkv, r * lddkv + HD + c, dV[(size_t)r * HD + c]);
    }
  free(dqa);
  free(dK);
  return UZ_OK;
}
UZ_SAME_SIGNATURE(uz_sra_bwd);

int uz_ref_abi_version(void) { return 1; }
