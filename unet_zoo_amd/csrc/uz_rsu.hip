// U^2-Net specific bandwidth-bound kernels for gfx950 (MI355X):
//   * bilinear resize (align_corners=False) forward / backward      (reference: u2net.py:19-22)
//   * gradient merge of a residual + max-pooled tensor               (u2net.py:74 `hx1d + hxin`, :221-229)
//   * 3x3 side heads C -> 1 logit map, forward / backward            (u2net.py:238-243, 277-287)
//   * 1x1 fuse convolution over the six NCHW fp32 side maps          (u2net.py:244, 288)
// Activations are NHWC (16-byte accesses per lane); logits are NCHW fp32 like the reference's outputs.
// Every reduction is a deterministic two-stage sum (per-workgroup partial rows + a finalize kernel).
#include "uz_common.h"

namespace {

template <typename T> __device__ __forceinline__ void load_f(const T* p, float* f) {
  const Vec16<T> v = ld16(p);
#pragma unroll
  for (int i = 0; i < ElemTraits<T>::VEC; ++i) f[i] = (float)v.v[i];
}
template <typename T> __device__ __forceinline__ void store_f(T* p, const float* f) {
  Vec16<T> v;
#pragma unroll
  for (int i = 0; i < ElemTraits<T>::VEC; ++i) v.v[i] = (T)f[i];
  st16(p, v);
}

inline int grid_cap(long long units, int per_block) {
  long long g = (units + per_block - 1) / per_block;
  const long long cap = (long long)UZ_NUM_CU * 16;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// ---------------------------------------------------------------------------------------------
// Bilinear source index, exactly as ATen's upsample_bilinear2d with align_corners=False and an
// explicit output size: scale = in/out (fp32), src = scale*(dst+0.5)-0.5 clamped at 0,
// i0 = min(int(src), in-1), i1 = i0 + (i0 < in-1), l1 = clamp(src - i0, 0, 1), l0 = 1 - l1.
// ---------------------------------------------------------------------------------------------
struct Tap2 {
  int i0, i1;
  float l0, l1;
};
// align_corners=True (nn.Upsample(..., align_corners=True) of nested_unet.py:32): scale = (in-1)/(out-1), src = scale*dst.
__device__ __forceinline__ Tap2 bilinear_tap(int o, float scale, int in, int ac) {
  float s = ac ? scale * (float)o : scale * ((float)o + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  Tap2 t;
  t.i0 = min((int)s, in - 1);
  t.i1 = t.i0 + (t.i0 < in - 1 ? 1 : 0);
  t.l1 = fminf(fmaxf(s - (float)t.i0, 0.f), 1.f);
  t.l0 = 1.f - t.l1;
  return t;
}

struct ResizeArgs {
  const void* src;
  void* dst;
  long long src_img, dst_img;  // image strides in elements
  int lds_, ldd;               // pixel strides in elements
  int N, Hi, Wi, Ho, Wo, C;
  float sh, sw;                // Hi/Ho, Wi/Wo; align_corners: (Hi-1)/(Ho-1), (Wi-1)/(Wo-1)
  int ac;                      // align_corners
};

// forward: one thread per output pixel x VEC channels (VECP) or x 1 channel
template <typename T, bool VECP>
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const ResizeArgs a) {
  constexpr int VEC = VECP ? ElemTraits<T>::VEC : 1;
  const int CC = a.C / VEC;
  const long long total = (long long)a.N * a.Ho * a.Wo * CC;
  const T* __restrict__ x = static_cast<const T*>(a.src);
  T* __restrict__ y = static_cast<T*>(a.dst);
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(idx % CC);
    long long u = idx / CC;
    const int ow = (int)(u % a.Wo);
    u /= a.Wo;
    const int oh = (int)(u % a.Ho);
    const int n = (int)(u / a.Ho);
    const Tap2 th = bilinear_tap(oh, a.sh, a.Hi, a.ac), tw = bilinear_tap(ow, a.sw, a.Wi, a.ac);
    const T* xb = x + (size_t)n * a.src_img + cc * VEC;
    const size_t p00 = ((size_t)th.i0 * a.Wi + tw.i0) * a.lds_, p01 = ((size_t)th.i0 * a.Wi + tw.i1) * a.lds_;
    const size_t p10 = ((size_t)th.i1 * a.Wi + tw.i0) * a.lds_, p11 = ((size_t)th.i1 * a.Wi + tw.i1) * a.lds_;
    T* yo = y + (size_t)n * a.dst_img + ((size_t)oh * a.Wo + ow) * a.ldd + cc * VEC;
    if constexpr (VECP) {
      float v00[VEC], v01[VEC], v10[VEC], v11[VEC], r[VEC];
      load_f(xb + p00, v00);
      load_f(xb + p01, v01);
      load_f(xb + p10, v10);
      load_f(xb + p11, v11);
#pragma unroll
      for (int i = 0; i < VEC; ++i)
        r[i] = th.l0 * (tw.l0 * v00[i] + tw.l1 * v01[i]) + th.l1 * (tw.l0 * v10[i] + tw.l1 * v11[i]);
      store_f(yo, r);
    } else {
      const float v00 = (float)xb[p00], v01 = (float)xb[p01], v10 = (float)xb[p10], v11 = (float)xb[p11];
      *yo = (T)(th.l0 * (tw.l0 * v00 + tw.l1 * v01) + th.l1 * (tw.l0 * v10 + tw.l1 * v11));
    }
  }
}

// candidate output range [lo, hi] whose taps can touch input index i
__device__ __forceinline__ void cand_range(int i, float scale, int out, int ac, int& lo, int& hi) {
  if (ac) {
    if (scale <= 0.f) {   // a single input row / column feeds every output
      lo = 0;
      hi = out - 1;
      return;
    }
    const float inv = 1.f / scale;
    lo = (int)floorf(((float)i - 1.f) * inv) - 1;
    hi = (int)ceilf(((float)i + 1.f) * inv) + 1;
  } else {
    const float inv = 1.f / scale;
    lo = (int)floorf(((float)i - 0.5f) * inv - 0.5f) - 1;
    hi = (int)ceilf(((float)i + 1.5f) * inv - 0.5f) + 1;
  }
  lo = max(lo, 0);
  hi = min(hi, out - 1);
}
__device__ __forceinline__ float tap_weight(int o, float scale, int in, int i, int ac) {
  const Tap2 t = bilinear_tap(o, scale, in, ac);
  return (t.i0 == i ? t.l0 : 0.f) + (t.i1 == i ? t.l1 : 0.f);
}

// backward, gather form (no atomics): src = gradient at OUTPUT resolution (Ho, Wo), dst = gradient at
// INPUT resolution (Hi, Wi).  VECP: one thread per input pixel x VEC channels.
template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_vec_kernel(const ResizeArgs a) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int CC = a.C / VEC;
  const long long total = (long long)a.N * a.Hi * a.Wi * CC;
  const T* __restrict__ g = static_cast<const T*>(a.src);
  T* __restrict__ dx = static_cast<T*>(a.dst);
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(idx % CC);
    long long u = idx / CC;
    const int iw = (int)(u % a.Wi);
    u /= a.Wi;
    const int ih = (int)(u % a.Hi);
    const int n = (int)(u / a.Hi);
    int hlo, hhi, wlo, whi;
    cand_range(ih, a.sh, a.Ho, a.ac, hlo, hhi);
    cand_range(iw, a.sw, a.Wo, a.ac, wlo, whi);
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    const T* gb = g + (size_t)n * a.src_img + cc * VEC;
    for (int oh = hlo; oh <= hhi; ++oh) {
      const float wy = tap_weight(oh, a.sh, a.Hi, ih, a.ac);
      if (wy == 0.f) continue;
      for (int ow = wlo; ow <= whi; ++ow) {
        const float wx = tap_weight(ow, a.sw, a.Wi, iw, a.ac);
        if (wx == 0.f) continue;
        float v[VEC];
        load_f(gb + ((size_t)oh * a.Wo + ow) * a.lds_, v);
        const float wgt = wy * wx;
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = fmaf(wgt, v[i], acc[i]);
      }
    }
    store_f(dx + (size_t)n * a.dst_img + ((size_t)ih * a.Wi + iw) * a.ldd + cc * VEC, acc);
  }
}

// scalar channels (the 1-channel logit maps, scale factors up to 32): one WAVE per input element,
// the lanes share the candidate window and reduce with shuffles.
template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_wave_kernel(const ResizeArgs a) {
  const long long total = (long long)a.N * a.Hi * a.Wi * a.C;
  const int lane = threadIdx.x & 63;
  const T* __restrict__ g = static_cast<const T*>(a.src);
  T* __restrict__ dx = static_cast<T*>(a.dst);
  for (long long e = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); e < total; e += (long long)gridDim.x * 4) {
    const int c = (int)(e % a.C);
    long long u = e / a.C;
    const int iw = (int)(u % a.Wi);
    u /= a.Wi;
    const int ih = (int)(u % a.Hi);
    const int n = (int)(u / a.Hi);
    int hlo, hhi, wlo, whi;
    cand_range(ih, a.sh, a.Ho, a.ac, hlo, hhi);
    cand_range(iw, a.sw, a.Wo, a.ac, wlo, whi);
    const int nw = whi - wlo + 1, cnt = (hhi - hlo + 1) * nw;
    const T* gb = g + (size_t)n * a.src_img + c;
    float acc = 0.f;
    for (int k = lane; k < cnt; k += 64) {
      const int oh = hlo + k / nw, ow = wlo + k % nw;
      const float wgt = tap_weight(oh, a.sh, a.Hi, ih, a.ac) * tap_weight(ow, a.sw, a.Wi, iw, a.ac);
      if (wgt != 0.f) acc = fmaf(wgt, (float)gb[((size_t)oh * a.Wo + ow) * a.lds_], acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) dx[(size_t)n * a.dst_img + ((size_t)ih * a.Wi + iw) * a.ldd + c] = (T)acc;
  }
}

// ---------------------------------------------------------------------------------------------
// Pixel-grid moves of NHWC tensors, 16 bytes per thread:
//   mode 0  copy        dst[n, h, w] = src[n, h, w]                      (a tensor into its slot of a concat buffer)
//   mode 1  subsample   dst[n, h, w] = src[n, 2h, 2w]                    (what a stride-2 convolution keeps / reads)
//   mode 2  zero-insert dst[n, h, w] = (h, w even) ? src[n, h/2, w/2] : 0 (its gradient, spread back between zeros)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void resample2_kernel(const T* __restrict__ src, int lds_, int Hs, int Ws,
                                                        T* __restrict__ dst, int ldd, int N, int Hd, int Wd, int C, int mode) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int CC = C / VEC;
  const long long total = (long long)N * Hd * Wd * CC;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(idx % CC);
    long long u = idx / CC;
    const int w = (int)(u % Wd);
    u /= Wd;
    const int h = (int)(u % Hd);
    const int n = (int)(u / Hd);
    Vec16<T> v;
    bool take = true;
    int hs = h, ws = w;
    if (mode == 1) {
      hs = 2 * h;
      ws = 2 * w;
    } else if (mode == 2) {
      take = !((h | w) & 1);
      hs = h >> 1;
      ws = w >> 1;
    }
    if (take) {
      v = ld16(src + (((size_t)n * Hs + hs) * Ws + ws) * lds_ + cc * VEC);
    } else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) v.v[i] = (T)0.f;
    }
    st16(dst + (((size_t)n * Hd + h) * Wd + w) * ldd + cc * VEC, v);
  }
}

// ---------------------------------------------------------------------------------------------
// out = g0 + g1 + unpool(gp): the total gradient of a tensor `act` that was consumed directly
// (g0, g1; either may be null) and through MaxPool2d(2,2) (gp; routed to the FIRST maximum of each
// window in (0,0),(0,1),(1,0),(1,1) order, the element ATen records).
// ---------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ void rsu_pin16(Vec16<T>& v) {   // see pin16 in uz_gemm_dma.hip
  unsigned* r = reinterpret_cast<unsigned*>(&v);
  asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));
}

template <typename T, bool POOL>
__global__ __launch_bounds__(256) void grad_combine_kernel(int N, int H, int W, int C, const T* __restrict__ act,
                                                           int lda, const T* __restrict__ g0, int ld0,
                                                           const T* __restrict__ g1, int ld1,
                                                           const T* __restrict__ gp, int ldp,
                                                           T* __restrict__ out, int ldo, int pool_ceil) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int CC = C / VEC;
  const int Ho = (H + 1) >> 1, Wo = (W + 1) >> 1;   // window grid (ceil): every pixel in exactly one window
  const int Hp = pool_ceil ? Ho : H >> 1, Wp = pool_ceil ? Wo : W >> 1;
  const long long total = POOL ? (long long)N * Ho * Wo * CC : (long long)N * H * W * CC;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c0 = (int)(idx % CC) * VEC;
    const long long u = idx / CC;
    if constexpr (!POOL) {
      float s[VEC], v[VEC];
      load_f(g0 + (size_t)u * ld0 + c0, s);
      if (g1 != nullptr) {
        load_f(g1 + (size_t)u * ld1 + c0, v);
#pragma unroll
        for (int i = 0; i < VEC; ++i) s[i] += v[i];
      }
      store_f(out + (size_t)u * ldo + c0, s);
    } else {
      const int wo = (int)(u % Wo);
      const long long t = u / Wo;
      const int ho = (int)(t % Ho);
      const int img = (int)(t / Ho);
      const size_t p00 = ((size_t)img * H + 2 * ho) * W + 2 * wo;
      float av[4][VEC], gv[VEC];
      int best[VEC];
      bool in[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) in[k] = 2 * ho + (k >> 1) < H && 2 * wo + (k & 1) < W;
      const bool has_pool = ho < Hp && wo < Wp;
      // every load of the window UNCONDITIONAL and in flight together (pixel 0 of `act` for a clipped tap / a missing operand,
      // dropped by a select; the registers pass through an empty asm so that the selects cannot pull the loads back under
      // branches): as loads under `if (in[k])` / `if (g != nullptr)` the thirteen of them came one round trip after the other
      Vec16<T> gpr, ar[4], g0r[4], g1r[4];
      gpr = ld16(has_pool ? gp + (((size_t)img * Hp + ho) * Wp + wo) * ldp + c0 : act + c0);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const size_t p = p00 + (k >> 1) * W + (k & 1);
        ar[k] = ld16(in[k] ? act + p * lda + c0 : act + c0);
        g0r[k] = ld16(in[k] && g0 != nullptr ? g0 + p * ld0 + c0 : act + c0);
        g1r[k] = ld16(in[k] && g1 != nullptr ? g1 + p * ld1 + c0 : act + c0);
      }
      rsu_pin16(gpr);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        rsu_pin16(ar[k]);
        rsu_pin16(g0r[k]);
        rsu_pin16(g1r[k]);
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) gv[i] = has_pool ? (float)gpr.v[i] : 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < VEC; ++i) av[k][i] = (float)ar[k].v[i];
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        best[i] = 0;
        float m = av[0][i];
#pragma unroll
        for (int k = 1; k < 4; ++k)
          if (in[k] && av[k][i] > m) {
            m = av[k][i];
            best[i] = k;
          }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (!in[k]) continue;
        const size_t p = p00 + (k >> 1) * W + (k & 1);
        float s[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) s[i] = best[i] == k ? gv[i] : 0.f;
        if (g0 != nullptr) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) s[i] += (float)g0r[k].v[i];
        }
        if (g1 != nullptr) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) s[i] += (float)g1r[k].v[i];
        }
        store_f(out + p * ldo + c0, s);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Side head: Conv2d(C, 1, 3, padding=1).  Forward in two bandwidth-bound steps that read x once:
//   t[n][tap][p] = sum_c x[p][c] * w[c][tap]           (side_taps_kernel: 9 dot products per pixel)
//   y[n][p]      = bias + sum_tap t[n][tap][p + off(tap)]   (side_tapsum_kernel, zero outside)
// Backward reads dy at the nine shifted positions directly: dt[q][tap] = dy[q - off(tap)].
// LPP lanes share a pixel (16 bytes of channels each, CH chunks per lane for C/VEC > 64).
// ---------------------------------------------------------------------------------------------
template <typename T, int CH>
__global__ __launch_bounds__(256) void side_taps_kernel(const T* __restrict__ x, int ldx, int N, int HW, int C,
                                                        const float* __restrict__ w, float* __restrict__ t) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int LPP = C / (VEC * CH);
  const int ppb = blockDim.x / LPP;
  const int sub = threadIdx.x % LPP, pl = threadIdx.x / LPP;
  const int P = N * HW;
  float wr[9][CH][VEC];
#pragma unroll
  for (int j = 0; j < CH; ++j)
#pragma unroll
    for (int i = 0; i < VEC; ++i)
#pragma unroll
      for (int k = 0; k < 9; ++k) wr[k][j][i] = w[(size_t)((sub + j * LPP) * VEC + i) * 9 + k];
  for (int p0 = blockIdx.x * ppb; p0 < P; p0 += gridDim.x * ppb) {
    const int p = p0 + pl;
    float v[CH][VEC];
#pragma unroll
    for (int j = 0; j < CH; ++j) {   // unconditional loads (pixel 0 past the end: its sums are not stored), DESIGN 3h
      load_f(x + (size_t)(p < P ? p : 0) * ldx + (sub + j * LPP) * VEC, v[j]);
    }
    const int img = p / HW, hw = p - img * HW;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < CH; ++j)
#pragma unroll
        for (int i = 0; i < VEC; ++i) s = fmaf(v[j][i], wr[k][j][i], s);
      for (int o = LPP >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o);
      if (sub == 0 && p < P) t[((size_t)img * 9 + k) * HW + hw] = s;
    }
  }
}

__global__ __launch_bounds__(256) void side_tapsum_kernel(const float* __restrict__ t, int N, int H, int W,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          long long out_img) {
  const int HW = H * W;
  const long long total = (long long)N * HW;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(idx / HW), hw = (int)(idx - (long long)n * HW);
    const int h = hw / W, w = hw - h * W;
    float s = bias != nullptr ? bias[0] : 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int hh = h + k / 3 - 1, ww = w + k % 3 - 1;
      if ((unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W) s += t[((size_t)n * 9 + k) * HW + hh * W + ww];
    }
    out[(size_t)n * out_img + hw] = s;
  }
}

template <typename T, int CH>
__global__ __launch_bounds__(256) void side_bwd_kernel(const T* __restrict__ x, int ldx, int N, int H, int W, int C,
                                                       const float* __restrict__ w, const float* __restrict__ g,
                                                       long long g_img, T* __restrict__ dx, int lddx,
                                                       float* __restrict__ partial) {
  // partial[blockIdx.x][9][C + 1]: sums of dt*x (C values) and dt (1 value) of this workgroup
  constexpr int VEC = ElemTraits<T>::VEC;
  constexpr int NV = CH * VEC;
  __shared__ float red[256 * (NV + 1)];
  const int LPP = C / (VEC * CH);
  const int ppb = blockDim.x / LPP;
  const int sub = threadIdx.x % LPP, pl = threadIdx.x / LPP;
  const int HW = H * W, P = N * HW;
  float wr[9][CH][VEC], aw[9][CH][VEC], ab[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    ab[k] = 0.f;
#pragma unroll
    for (int j = 0; j < CH; ++j)
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        wr[k][j][i] = w[(size_t)((sub + j * LPP) * VEC + i) * 9 + k];
        aw[k][j][i] = 0.f;
      }
  }
  for (int p0 = blockIdx.x * ppb; p0 < P; p0 += gridDim.x * ppb) {
    const int p = p0 + pl;
    const bool ok = p < P;
    float v[CH][VEC], gk[9], d[CH][VEC];
    const int img = ok ? p / HW : 0, hw = ok ? p - img * HW : 0;
    const int h = hw / W, ww0 = hw - h * W;
#pragma unroll
    // unconditional loads, all in flight (pixel 0 / tap position 0 out of range, dropped by the selects below; the tap values
    // pass through an empty asm so that the selects cannot pull the loads back under branches): DESIGN 3h
    for (int j = 0; j < CH; ++j) {
      load_f(x + (size_t)(ok ? p : 0) * ldx + (sub + j * LPP) * VEC, v[j]);
#pragma unroll
      for (int i = 0; i < VEC; ++i) d[j][i] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int hh = h - (k / 3 - 1), ww = ww0 - (k % 3 - 1);  // dt[q][tap] = dy[q - off(tap)]
      const bool in = ok && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
      gk[k] = g[in ? (size_t)img * g_img + hh * W + ww : 0];
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) asm volatile("" : "+v"(gk[k]));
    if (!ok) {
#pragma unroll
      for (int j = 0; j < CH; ++j)
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[j][i] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int hh = h - (k / 3 - 1), ww = ww0 - (k % 3 - 1);
      gk[k] = (ok && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W) ? gk[k] : 0.f;
      ab[k] += gk[k];
#pragma unroll
      for (int j = 0; j < CH; ++j)
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          d[j][i] = fmaf(gk[k], wr[k][j][i], d[j][i]);
          aw[k][j][i] = fmaf(gk[k], v[j][i], aw[k][j][i]);
        }
    }
    if (dx != nullptr && ok) {
#pragma unroll
      for (int j = 0; j < CH; ++j) store_f(dx + (size_t)p * lddx + (sub + j * LPP) * VEC, d[j]);
    }
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < CH; ++j)
#pragma unroll
      for (int i = 0; i < VEC; ++i) red[threadIdx.x * (NV + 1) + j * VEC + i] = aw[k][j][i];
    red[threadIdx.x * (NV + 1) + NV] = ab[k];
    __syncthreads();
    if (pl == 0) {
      float* row = partial + ((size_t)blockIdx.x * 9 + k) * (C + 1);
#pragma unroll
      for (int e = 0; e <= NV; ++e) {
        float s = 0.f;
        for (int r = 0; r < ppb; ++r) s += red[(r * LPP + sub) * (NV + 1) + e];
        if (e < NV) row[(sub + (e / VEC) * LPP) * VEC + (e % VEC)] = s;
        else if (sub == 0) row[C] = s;
      }
    }
  }
}

// dw[c][tap] (reference layout (1, C, 3, 3)) and db from the partial rows; 1024 threads / 32 elements
__global__ __launch_bounds__(1024) void side_bwd_finalize_kernel(const float* __restrict__ partial, int rows,
                                                                 int C, float* __restrict__ dw,
                                                                 float* __restrict__ db) {
  __shared__ double sh[32][33];
  const int el = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int ne = 9 * (C + 1);
  const int e = blockIdx.x * 32 + el;
  double s = 0.0;
  if (e < ne)
    for (int r = g; r < rows; r += 32) s += (double)partial[(size_t)r * ne + e];
  sh[g][el] = s;
  __syncthreads();
  if (g == 0 && e < ne) {
    double t = 0.0;
    for (int r = 0; r < 32; ++r) t += sh[r][el];
    const int k = e / (C + 1), c = e - k * (C + 1);
    if (c < C) dw[c * 9 + k] = (float)t;
    else if (k == 4 && db != nullptr) db[0] = (float)t;  // centre tap: dt == dy
  }
}

// ---------------------------------------------------------------------------------------------
// Fuse conv: Conv2d(Cc = 6*K, K, 1) on the NCHW fp32 concat of the side maps.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fuse_fwd_kernel(const float* __restrict__ d, int N, int HW, int Cc, int K,
                                                       const float* __restrict__ w, const float* __restrict__ b,
                                                       float* __restrict__ out) {
  const long long total = (long long)N * HW;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(idx / HW), p = (int)(idx - (long long)n * HW);
    for (int o = 0; o < K; ++o) {
      float s = b != nullptr ? b[o] : 0.f;
      for (int c = 0; c < Cc; ++c) s = fmaf(w[o * Cc + c], d[((size_t)n * Cc + c) * HW + p], s);
      out[((size_t)n * K + o) * HW + p] = s;
    }
  }
}

struct FuseExtra {
  const float* p[8];
};

// dcat[n][c][p] = sum_o w[o][c] g[n][o][p]  +  extra[c / K][n][c % K][p]
__global__ __launch_bounds__(256) void fuse_bwd_data_kernel(int N, int HW, int Cc, int K, const float* __restrict__ w,
                                                            const float* __restrict__ g, const FuseExtra ex,
                                                            float* __restrict__ dcat) {
  const long long total = (long long)N * HW;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(idx / HW), p = (int)(idx - (long long)n * HW);
    for (int c = 0; c < Cc; ++c) {
      float s = 0.f;
      if (g != nullptr)
        for (int o = 0; o < K; ++o) s = fmaf(w[o * Cc + c], g[((size_t)n * K + o) * HW + p], s);
      const float* e = ex.p[c / K];
      if (e != nullptr) s += e[((size_t)n * K + (c % K)) * HW + p];
      dcat[((size_t)n * Cc + c) * HW + p] = s;
    }
  }
}

// partial[e][chunk], e = o*(Cc+1) + c: sum over this chunk's pixels of g[n][o][p] * (c < Cc ? d[n][c][p] : 1)
__global__ __launch_bounds__(256) void fuse_bwd_w_kernel(const float* __restrict__ d, int N, int HW, int Cc, int K,
                                                         const float* __restrict__ g, float* __restrict__ partial) {
  __shared__ float red[256];
  const int e = blockIdx.y, o = e / (Cc + 1), c = e - o * (Cc + 1);
  const long long total = (long long)N * HW;
  float s = 0.f;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(idx / HW), p = (int)(idx - (long long)n * HW);
    const float gv = g[((size_t)n * K + o) * HW + p];
    s += c < Cc ? gv * d[((size_t)n * Cc + c) * HW + p] : gv;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[(size_t)e * gridDim.x + blockIdx.x] = red[0];
}

__global__ __launch_bounds__(64) void fuse_bwd_w_finalize_kernel(const float* __restrict__ partial, int chunks,
                                                                 int Cc, int K, float* __restrict__ dw,
                                                                 float* __restrict__ db) {
  const int e = blockIdx.x, o = e / (Cc + 1), c = e - o * (Cc + 1);
  double s = 0.0;
  for (int r = threadIdx.x; r < chunks; r += 64) s += (double)partial[(size_t)e * chunks + r];
#pragma unroll
  for (int k = 32; k > 0; k >>= 1) s += __shfl_xor(s, k);
  if (threadIdx.x == 0) {
    if (c < Cc) dw[o * Cc + c] = (float)s;
    else if (db != nullptr) db[o] = (float)s;
  }
}

constexpr int FUSE_CHUNKS = 128;

inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// lanes per pixel / chunks per lane of the side-head kernels; 0 when C is unsupported
inline int side_ch(int dtype, int C) {
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  if (C <= 0 || C % vec != 0) return 0;
  const int chunks = C / vec;
  const int ch = chunks > 64 ? 2 : 1;
  if (chunks % ch != 0 || !pow2(chunks / ch) || chunks / ch > 64) return 0;
  return ch;
}

inline int side_grid(int dtype, int C, long long P) {
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  const int ch = side_ch(dtype, C);
  const int ppb = 256 / (C / (vec * ch));
  long long g = (P + ppb - 1) / ppb;
  if (g > UZ_NUM_CU * 4) g = UZ_NUM_CU * 4;
  if (g < 1) g = 1;
  return (int)g;
}

int resize_check(const char* fn, int dtype, const void* a, const void* b, int lda, int ldb, int N, int Hi, int Wi,
                 int C, int Ho, int Wo) {
  UZ_REQUIRE(dtype == UZ_F32 || dtype == UZ_BF16, "%s: bad dtype", fn);
  UZ_REQUIRE(a && b, "%s: null pointer", fn);
  UZ_REQUIRE(N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0, "%s: bad shape", fn);
  UZ_REQUIRE(lda >= C && ldb >= C, "%s: bad pixel stride", fn);
  UZ_REQUIRE((long long)N * Ho * Wo * C < (1LL << 40) && (long long)N * Hi * Wi * C < (1LL << 40), "%s: too large", fn);
  return UZ_OK;
}

inline bool vec_ok(int dtype, const void* a, const void* b, int lda, int ldb, long long sa, long long sb, int C) {
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  return C % vec == 0 && lda % vec == 0 && ldb % vec == 0 && sa % vec == 0 && sb % vec == 0 &&
         ((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0;
}

}  // namespace

static float resize_scale(int in, int out, int ac) {
  if (!ac) return (float)in / (float)out;
  return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;   // ATen: area_pixel_compute_scale
}

extern "C" int uz_bilinear_fwd(int dtype, const void* x, int ldx, long long x_img_stride, int N, int Hi, int Wi,
                               int C, void* y, int ldy, long long y_img_stride, int Ho, int Wo, void* stream) {
  return uz_resize_bilinear_fwd(dtype, x, ldx, x_img_stride, N, Hi, Wi, C, y, ldy, y_img_stride, Ho, Wo, 0, stream);
}

extern "C" int uz_resize_bilinear_fwd(int dtype, const void* x, int ldx, long long x_img_stride, int N, int Hi, int Wi,
                                      int C, void* y, int ldy, long long y_img_stride, int Ho, int Wo,
                                      int align_corners, void* stream) {
  const int rc = resize_check("uz_bilinear_fwd", dtype, x, y, ldx, ldy, N, Hi, Wi, C, Ho, Wo);
  if (rc != UZ_OK) return rc;
  const int ac = align_corners ? 1 : 0;
  ResizeArgs a{x, y, x_img_stride, y_img_stride, ldx, ldy, N, Hi, Wi, Ho, Wo, C, resize_scale(Hi, Ho, ac),
               resize_scale(Wi, Wo, ac), ac};
  hipStream_t s = (hipStream_t)stream;
  const bool v = vec_ok(dtype, x, y, ldx, ldy, x_img_stride, y_img_stride, C);
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  const long long total = (long long)N * Ho * Wo * (v ? C / vec : C);
  const dim3 grid(grid_cap(total, 256)), block(256);
  if (dtype == UZ_BF16) {
    if (v) hipLaunchKernelGGL((bilinear_fwd_kernel<bf16_t, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((bilinear_fwd_kernel<bf16_t, false>), grid, block, 0, s, a);
  } else {
    if (v) hipLaunchKernelGGL((bilinear_fwd_kernel<float, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((bilinear_fwd_kernel<float, false>), grid, block, 0, s, a);
  }
  UZ_LAUNCH_CHECK("uz_bilinear_fwd");
  return UZ_OK;
}

extern "C" int uz_bilinear_bwd(int dtype, const void* g, int ldg, long long g_img_stride, int N, int Hi, int Wi,
                               int C, void* dx, int lddx, long long dx_img_stride, int Ho, int Wo, void* stream) {
  return uz_resize_bilinear_bwd(dtype, g, ldg, g_img_stride, N, Hi, Wi, C, dx, lddx, dx_img_stride, Ho, Wo, 0, stream);
}

extern "C" int uz_resize_bilinear_bwd(int dtype, const void* g, int ldg, long long g_img_stride, int N, int Hi, int Wi,
                                      int C, void* dx, int lddx, long long dx_img_stride, int Ho, int Wo,
                                      int align_corners, void* stream) {
  const int rc = resize_check("uz_bilinear_bwd", dtype, g, dx, ldg, lddx, N, Hi, Wi, C, Ho, Wo);
  if (rc != UZ_OK) return rc;
  const int ac = align_corners ? 1 : 0;
  ResizeArgs a{g, dx, g_img_stride, dx_img_stride, ldg, lddx, N, Hi, Wi, Ho, Wo, C, resize_scale(Hi, Ho, ac),
               resize_scale(Wi, Wo, ac), ac};
  hipStream_t s = (hipStream_t)stream;
  const bool v = vec_ok(dtype, g, dx, ldg, lddx, g_img_stride, dx_img_stride, C);
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  if (v) {
    const dim3 grid(grid_cap((long long)N * Hi * Wi * (C / vec), 256)), block(256);
    if (dtype == UZ_BF16) hipLaunchKernelGGL((bilinear_bwd_vec_kernel<bf16_t>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((bilinear_bwd_vec_kernel<float>), grid, block, 0, s, a);
  } else {
    const dim3 grid(grid_cap((long long)N * Hi * Wi * C, 4)), block(256);
    if (dtype == UZ_BF16) hipLaunchKernelGGL((bilinear_bwd_wave_kernel<bf16_t>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((bilinear_bwd_wave_kernel<float>), grid, block, 0, s, a);
  }
  UZ_LAUNCH_CHECK("uz_bilinear_bwd");
  return UZ_OK;
}

extern "C" int uz_resample2(int dtype, const void* src, int lds_, int N, int Hs, int Ws, int C, void* dst, int ldd,
                            int Hd, int Wd, int mode, void* stream) {
  UZ_REQUIRE(dtype == UZ_F32 || dtype == UZ_BF16, "uz_resample2: bad dtype");
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(src && dst && N > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && C > 0 && C % vec == 0, "uz_resample2: bad shape");
  UZ_REQUIRE(lds_ % vec == 0 && lds_ >= C && ldd % vec == 0 && ldd >= C, "uz_resample2: bad leading dimension");
  UZ_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "uz_resample2: pointers must be 16-byte aligned");
  if (mode == 0) UZ_REQUIRE(Hs == Hd && Ws == Wd, "uz_resample2: copy needs equal grids");
  else if (mode == 1) UZ_REQUIRE(Hd == (Hs + 1) / 2 && Wd == (Ws + 1) / 2, "uz_resample2: subsample needs Hd = ceil(Hs/2)");
  else if (mode == 2) UZ_REQUIRE(Hs == (Hd + 1) / 2 && Ws == (Wd + 1) / 2, "uz_resample2: zero-insert needs Hs = ceil(Hd/2)");
  else UZ_REQUIRE(false, "uz_resample2: bad mode %d", mode);
  const long long total = (long long)N * Hd * Wd * (C / vec);
  const dim3 grid(grid_cap(total, 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UZ_BF16)
    hipLaunchKernelGGL((resample2_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)src, lds_, Hs, Ws, (bf16_t*)dst, ldd, N, Hd, Wd, C, mode);
  else
    hipLaunchKernelGGL((resample2_kernel<float>), grid, block, 0, s, (const float*)src, lds_, Hs, Ws, (float*)dst, ldd, N, Hd, Wd, C, mode);
  UZ_LAUNCH_CHECK("uz_resample2");
  return UZ_OK;
}

extern "C" int uz_pool_grad_combine(int dtype, int N, int H, int W, int C, const void* act, int lda, const void* g0,
                                    int ldg0, const void* g1, int ldg1, const void* gp, int ldgp, void* out,
                                    int ldo, int pool_ceil, void* stream) {
  UZ_REQUIRE(dtype == UZ_F32 || dtype == UZ_BF16, "uz_pool_grad_combine: bad dtype");
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % vec == 0, "uz_pool_grad_combine: bad shape");
  UZ_REQUIRE(out && ldo % vec == 0 && ldo >= C, "uz_pool_grad_combine: bad out");
  UZ_REQUIRE(g0 || gp, "uz_pool_grad_combine: needs g0 or gp (pass a lone g1 as g0)");
  if (g0) UZ_REQUIRE(ldg0 % vec == 0 && ldg0 >= C, "uz_pool_grad_combine: bad ldg0");
  if (g1) UZ_REQUIRE(ldg1 % vec == 0 && ldg1 >= C, "uz_pool_grad_combine: bad ldg1");
  if (gp) {
    UZ_REQUIRE(act && lda % vec == 0 && lda >= C && ldgp % vec == 0 && ldgp >= C, "uz_pool_grad_combine: bad act/gp");
  }
  hipStream_t s = (hipStream_t)stream;
  const long long total = gp ? (long long)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / vec) : (long long)N * H * W * (C / vec);
  const dim3 grid(grid_cap(total, 256)), block(256);
  if (dtype == UZ_BF16) {
    if (gp) hipLaunchKernelGGL((grad_combine_kernel<bf16_t, true>), grid, block, 0, s, N, H, W, C, (const bf16_t*)act, lda, (const bf16_t*)g0, ldg0, (const bf16_t*)g1, ldg1, (const bf16_t*)gp, ldgp, (bf16_t*)out, ldo, pool_ceil);
    else hipLaunchKernelGGL((grad_combine_kernel<bf16_t, false>), grid, block, 0, s, N, H, W, C, (const bf16_t*)act, lda, (const bf16_t*)g0, ldg0, (const bf16_t*)g1, ldg1, (const bf16_t*)gp, ldgp, (bf16_t*)out, ldo, pool_ceil);
  } else {
    if (gp) hipLaunchKernelGGL((grad_combine_kernel<float, true>), grid, block, 0, s, N, H, W, C, (const float*)act, lda, (const float*)g0, ldg0, (const float*)g1, ldg1, (const float*)gp, ldgp, (float*)out, ldo, pool_ceil);
    else hipLaunchKernelGGL((grad_combine_kernel<float, false>), grid, block, 0, s, N, H, W, C, (const float*)act, lda, (const float*)g0, ldg0, (const float*)g1, ldg1, (const float*)gp, ldgp, (float*)out, ldo, pool_ceil);
  }
  UZ_LAUNCH_CHECK("uz_pool_grad_combine");
  return UZ_OK;
}

static int side_check(const char* fn, int dtype, const void* x, int ldx, int N, int H, int W, int C) {
  UZ_REQUIRE(dtype == UZ_F32 || dtype == UZ_BF16, "%s: bad dtype", fn);
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(side_ch(dtype, C) > 0, "%s: C=%d unsupported (C/%d must be a power of two <= 128)", fn, C, vec);
  UZ_REQUIRE(x && ldx % vec == 0 && ldx >= C && ((uintptr_t)x & 15) == 0, "%s: bad x", fn);
  UZ_REQUIRE(N > 0 && H > 0 && W > 0 && (long long)N * H * W < (1LL << 31), "%s: bad shape", fn);
  return UZ_OK;
}

extern "C" int uz_sideconv3x3_fwd(int dtype, const void* x, int ldx, int N, int H, int W, int C, const float* w,
                                  const float* bias, float* taps_ws, float* out, long long out_img_stride,
                                  void* stream) {
  const int rc = side_check("uz_sideconv3x3_fwd", dtype, x, ldx, N, H, W, C);
  if (rc != UZ_OK) return rc;
  UZ_REQUIRE(w && taps_ws && out && out_img_stride >= (long long)H * W, "uz_sideconv3x3_fwd: bad pointers");
  hipStream_t s = (hipStream_t)stream;
  const int HW = H * W;
  const dim3 grid(side_grid(dtype, C, (long long)N * HW)), block(256);
  const int ch = side_ch(dtype, C);
  if (dtype == UZ_BF16) {
    if (ch == 1) hipLaunchKernelGGL((side_taps_kernel<bf16_t, 1>), grid, block, 0, s, (const bf16_t*)x, ldx, N, HW, C, w, taps_ws);
    else hipLaunchKernelGGL((side_taps_kernel<bf16_t, 2>), grid, block, 0, s, (const bf16_t*)x, ldx, N, HW, C, w, taps_ws);
  } else {
    if (ch == 1) hipLaunchKernelGGL((side_taps_kernel<float, 1>), grid, block, 0, s, (const float*)x, ldx, N, HW, C, w, taps_ws);
    else hipLaunchKernelGGL((side_taps_kernel<float, 2>), grid, block, 0, s, (const float*)x, ldx, N, HW, C, w, taps_ws);
  }
  UZ_LAUNCH_CHECK("uz_sideconv3x3_fwd(taps)");
  hipLaunchKernelGGL(side_tapsum_kernel, dim3(grid_cap((long long)N * HW, 256)), dim3(256), 0, s, taps_ws, N, H, W,
                     bias, out, out_img_stride);
  UZ_LAUNCH_CHECK("uz_sideconv3x3_fwd(sum)");
  return UZ_OK;
}

extern "C" long long uz_sideconv3x3_bwd_workspace_bytes(int dtype, int N, int H, int W, int C) {
  UZ_REQUIRE(dtype == UZ_F32 || dtype == UZ_BF16, "uz_sideconv3x3_bwd_workspace_bytes: bad dtype");
  UZ_REQUIRE(side_ch(dtype, C) > 0 && N > 0 && H > 0 && W > 0 && (long long)N * H * W < (1LL << 31),
             "uz_sideconv3x3_bwd_workspace_bytes: bad shape");
  return (long long)side_grid(dtype, C, (long long)N * H * W) * 9 * (C + 1) * (long long)sizeof(float);
}

extern "C" int uz_sideconv3x3_bwd(int dtype, const void* x, int ldx, int N, int H, int W, int C, const float* w,
                                  const float* g, long long g_img_stride, void* dx, int lddx, float* dw, float* db,
                                  void* workspace, void* stream) {
  const int rc = side_check("uz_sideconv3x3_bwd", dtype, x, ldx, N, H, W, C);
  if (rc != UZ_OK) return rc;
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(w && g && dw && workspace && g_img_stride >= (long long)H * W, "uz_sideconv3x3_bwd: bad pointers");
  if (dx) UZ_REQUIRE(lddx % vec == 0 && lddx >= C && ((uintptr_t)dx & 15) == 0, "uz_sideconv3x3_bwd: bad dx");
  hipStream_t s = (hipStream_t)stream;
  const int gsz = side_grid(dtype, C, (long long)N * H * W);
  const dim3 grid(gsz), block(256);
  const int ch = side_ch(dtype, C);
  float* part = static_cast<float*>(workspace);
  if (dtype == UZ_BF16) {
    if (ch == 1) hipLaunchKernelGGL((side_bwd_kernel<bf16_t, 1>), grid, block, 0, s, (const bf16_t*)x, ldx, N, H, W, C, w, g, g_img_stride, (bf16_t*)dx, lddx, part);
    else hipLaunchKernelGGL((side_bwd_kernel<bf16_t, 2>), grid, block, 0, s, (const bf16_t*)x, ldx, N, H, W, C, w, g, g_img_stride, (bf16_t*)dx, lddx, part);
  } else {
    if (ch == 1) hipLaunchKernelGGL((side_bwd_kernel<float, 1>), grid, block, 0, s, (const float*)x, ldx, N, H, W, C, w, g, g_img_stride, (float*)dx, lddx, part);
    else hipLaunchKernelGGL((side_bwd_kernel<float, 2>), grid, block, 0, s, (const float*)x, ldx, N, H, W, C, w, g, g_img_stride, (float*)dx, lddx, part);
  }
  UZ_LAUNCH_CHECK("uz_sideconv3x3_bwd");
  hipLaunchKernelGGL(side_bwd_finalize_kernel, dim3(uz_cdiv(9 * (C + 1), 32)), dim3(1024), 0, s, part, gsz, C, dw, db);
  UZ_LAUNCH_CHECK("uz_sideconv3x3_bwd(finalize)");
  return UZ_OK;
}

static int fuse_check(const char* fn, int N, int HW, int Cc, int K) {
  UZ_REQUIRE(N > 0 && HW > 0 && K > 0 && Cc > 0 && Cc % K == 0 && Cc / K <= 8, "%s: bad shape (Cc=%d, K=%d)", fn, Cc, K);
  UZ_REQUIRE((long long)N * HW < (1LL << 31) && (long long)K * (Cc + 1) <= 65535, "%s: too large", fn);
  return UZ_OK;
}

extern "C" int uz_fuse1x1_fwd(const float* d, int N, int HW, int Cc, int K, const float* w, const float* b,
                              float* out, void* stream) {
  const int rc = fuse_check("uz_fuse1x1_fwd", N, HW, Cc, K);
  if (rc != UZ_OK) return rc;
  UZ_REQUIRE(d && w && out, "uz_fuse1x1_fwd: null pointer");
  hipLaunchKernelGGL(fuse_fwd_kernel, dim3(grid_cap((long long)N * HW, 256)), dim3(256), 0, (hipStream_t)stream, d, N,
                     HW, Cc, K, w, b, out);
  UZ_LAUNCH_CHECK("uz_fuse1x1_fwd");
  return UZ_OK;
}

__global__ void fuse_zero_kernel(float* __restrict__ a, int na, float* __restrict__ b, int nb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < na) a[i] = 0.0f;
  if (i < nb) b[i] = 0.0f;
}

extern "C" long long uz_fuse1x1_bwd_workspace_bytes(int N, int HW, int Cc, int K) {
  const int rc = fuse_check("uz_fuse1x1_bwd_workspace_bytes", N, HW, Cc, K);
  if (rc != UZ_OK) return rc;
  return (long long)K * (Cc + 1) * FUSE_CHUNKS * (long long)sizeof(float);
}

extern "C" int uz_fuse1x1_bwd(const float* d, int N, int HW, int Cc, int K, const float* w, const float* g,
                              const float* const* g_extra, int n_extra, float* dcat, float* dw, float* db,
                              void* workspace, void* stream) {
  const int rc = fuse_check("uz_fuse1x1_bwd", N, HW, Cc, K);
  if (rc != UZ_OK) return rc;
  UZ_REQUIRE(d && w && dcat && dw && workspace, "uz_fuse1x1_bwd: null pointer");
  UZ_REQUIRE(n_extra == 0 || (g_extra != nullptr && n_extra == Cc / K), "uz_fuse1x1_bwd: n_extra must be 0 or Cc/K");
  hipStream_t s = (hipStream_t)stream;
  FuseExtra ex;
  for (int i = 0; i < 8; ++i) ex.p[i] = (i < n_extra) ? g_extra[i] : nullptr;
  hipLaunchKernelGGL(fuse_bwd_data_kernel, dim3(grid_cap((long long)N * HW, 256)), dim3(256), 0, s, N, HW, Cc, K, w, g,
                     ex, dcat);
  UZ_LAUNCH_CHECK("uz_fuse1x1_bwd(data)");
  float* part = static_cast<float*>(workspace);
  const int ne = K * (Cc + 1);
  if (g != nullptr) {
    hipLaunchKernelGGL(fuse_bwd_w_kernel, dim3(FUSE_CHUNKS, ne), dim3(256), 0, s, d, N, HW, Cc, K, g, part);
    UZ_LAUNCH_CHECK("uz_fuse1x1_bwd(weights)");
    hipLaunchKernelGGL(fuse_bwd_w_finalize_kernel, dim3(ne), dim3(64), 0, s, part, FUSE_CHUNKS, Cc, K, dw, db);
    UZ_LAUNCH_CHECK("uz_fuse1x1_bwd(finalize)");
  } else {
    // a kernel, not hipMemsetAsync: on this stack a memset NODE of a replayed hipGraph writes its value only in the
    // first replay (tools/graph_canary.py), and every entry point must be capturable
    hipLaunchKernelGGL(fuse_zero_kernel, dim3(uz_cdiv(K * Cc, 256)), dim3(256), 0, s, dw, K * Cc, db, db ? K : 0);
    UZ_LAUNCH_CHECK("uz_fuse1x1_bwd(zero)");
  }
  return UZ_OK;
}
