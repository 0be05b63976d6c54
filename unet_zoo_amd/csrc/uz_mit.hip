// MISSFormer / MiT blocks for gfx950 (reference unet_zoo/models/missformer.py):
//   * spatial-reduction attention  softmax(q k^T * scale) v  with head_dim 64, N queries against NK << N reduced
//     keys per image (EfficientSelfAtten :21-39, M_EfficientSelfAtten :113-128): forward, dQ and dK/dV kernels on
//     v_mfma_f32_32x32x16_bf16 (bf16) and a scalar fp32 path for the fp32 run mode;
//   * depthwise 3x3 convolution of MixFFN_skip (DWConv :168-177, "dwconv(fc1) + fc1" :205) forward / input
//     gradient / weight gradient;
//   * exact GELU (nn.GELU(), :196) forward / backward;
//   * space-to-depth (the Conv2d(dim, dim, r, r) of the spatial reduction becomes a GEMM over r*r*C columns);
//   * im2col of the NCHW fp32 network input for OverlapPatchEmbeddings' 7x7 stride-4 convolution (:238-250).
#include <math.h>
#include <stdint.h>

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

inline int grid_cap(long long units, int per_block, int waves = 16) {
  long long g = (units + per_block - 1) / per_block;
  const long long cap = (long long)UZ_NUM_CU * waves;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// ---------------------------------------------------------------------------------------------
// GELU (erf form): y = x * Phi(x);  backward dx = g * (Phi(x) + x * phi(x))
// ---------------------------------------------------------------------------------------------
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void gelu_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ g, int ldg,
                                                   T* __restrict__ y, int ldy, long long P, int C) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int CC = C / VEC;
  const long long total = P * CC;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c0 = (int)(idx % CC) * VEC;
    const long long p = idx / CC;
    float v[VEC], r[VEC];
    load_f(x + (size_t)p * ldx + c0, v);
    if constexpr (BWD) {
      float gv[VEC];
      load_f(g + (size_t)p * ldg + c0, gv);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float cdf = 0.5f * (1.f + erff(v[i] * 0.70710678118654752f));
        const float pdf = 0.3989422804014327f * __expf(-0.5f * v[i] * v[i]);
        r[i] = gv[i] * (cdf + v[i] * pdf);
      }
    } else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) r[i] = 0.5f * v[i] * (1.f + erff(v[i] * 0.70710678118654752f));
    }
    store_f(y + (size_t)p * ldy + c0, r);
  }
}

// ---------------------------------------------------------------------------------------------
// Depthwise 3x3, padding 1, NHWC.  wt is tap-major [9][C] fp32 (the host transposes the (C,1,3,3) parameter).
//   flags bit 0: add the centre input (the "+ fc1_out" of MixFFN_skip, or "+ g" in its gradient)
//   flags bit 1: flipped taps (gradient with respect to the input)
// One thread: 8 consecutive output pixels of a row x one 16-byte channel chunk; the 3 x 10 input vectors it
// needs are each loaded once.
// ---------------------------------------------------------------------------------------------
#ifndef DW_SW
#define DW_SW 6    // pixels per thread; measured on the missformer step (SW, waves/SIMD): (8,1) 2.24 ms, (8,2) 2.03, (6,2) 1.86, (4,3) 4.2 (spills)
#define DW_OCC 2
#endif
template <typename T>
__global__ __launch_bounds__(256, DW_OCC) void dwconv3x3_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ wt,
                                                        const float* __restrict__ bias, T* __restrict__ y, int ldy,
                                                        int N, int H, int W, int C, int flags) {
  constexpr int VEC = ElemTraits<T>::VEC;
  constexpr int SW = DW_SW;
  const int CC = C / VEC, WS = (W + SW - 1) / SW;
  const long long total = (long long)N * H * WS * CC;
  const bool skip = flags & 1, flip = flags & 2;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c0 = (int)(idx % CC) * VEC;
    long long u = idx / CC;
    const int ws = (int)(u % WS);
    u /= WS;
    const int h = (int)(u % H), n = (int)(u / H);
    float wr[9][VEC];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float* wp = wt + (size_t)(flip ? 8 - t : t) * C + c0;
#pragma unroll
      for (int i = 0; i < VEC; ++i) wr[t][i] = wp[i];
    }
    float bv[VEC], acc[SW][VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) bv[i] = 0.f;
    if (bias != nullptr) {
#pragma unroll
      for (int i = 0; i < VEC; i += 4) *reinterpret_cast<f32x4*>(bv + i) = *reinterpret_cast<const f32x4*>(bias + c0 + i);
    }
#pragma unroll
    for (int o = 0; o < SW; ++o)
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[o][i] = bv[i];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      __builtin_amdgcn_sched_barrier(0);   // one row's loads in flight at a time (three rows at once cost 320 VGPRs)
      const int hh = h + dy - 1;
      const bool rok = (unsigned)hh < (unsigned)H;
      const T* row = x + ((size_t)n * H + (rok ? hh : h)) * W * ldx + c0;
      // all ten loads of the row are issued before the first use: addresses are clamped into the row and the
      // out-of-range vectors zeroed afterwards (a branch per load would serialise their latencies)
      Vec16<T> raw[SW + 2];
#pragma unroll
      for (int j = 0; j < SW + 2; ++j) {
        const int ww = ws * SW - 1 + j;
        raw[j] = ld16(row + (size_t)min(max(ww, 0), W - 1) * ldx);
      }
#pragma unroll
      for (int j = 0; j < SW + 2; ++j) {
        const int ww = ws * SW - 1 + j;
        const bool ok = rok && (unsigned)ww < (unsigned)W;
        __builtin_amdgcn_sched_barrier(0);   // convert one packed vector at a time (all ten at once: +80 VGPRs)
        float v[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] = ok ? (float)raw[j].v[i] : 0.f;
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
          const int o = j - tx;   // output pixel ws*SW + o reads input column (ws*SW + o) + tx - 1
          if (o < 0 || o >= SW) continue;
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[o][i] = fmaf(wr[dy * 3 + tx][i], v[i], acc[o][i]);
          if (skip && dy == 1 && tx == 1) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[o][i] += v[i];
          }
        }
      }
    }
#pragma unroll
    for (int o = 0; o < SW; ++o) {
      const int w = ws * SW + o;
      if (w < W) store_f(y + (((size_t)n * H + h) * W + w) * ldy + c0, acc[o]);
    }
  }
}

// Weight / bias gradient of the depthwise convolution: part[row][10][C], taps 0..8 then the bias; a
// workgroup is (channel chunks) x (pixel segments of 32 along a row); segments are summed in LDS in a
// fixed order, rows by uz_sum_rows_f32.
#ifndef DWG_SEG_PX
#define DWG_SEG_PX 32
#endif
constexpr int DWG_SEG = DWG_SEG_PX;
template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ g,
                                                              int ldg, float* __restrict__ part, int N, int H, int W,
                                                              int C, int ccb, int pr) {
  constexpr int VEC = ElemTraits<T>::VEC;
  extern __shared__ float red[];   // [pr][ccb][10][VEC]
  const int CC = C / VEC, WS = (W + DWG_SEG - 1) / DWG_SEG;
  const int lc = threadIdx.x % ccb, ls = threadIdx.x / ccb;
  const int cc = blockIdx.y * ccb + lc;
  const long long seg = (long long)blockIdx.x * pr + ls, nseg = (long long)N * H * WS;
  float acc[10][VEC];
#pragma unroll
  for (int t = 0; t < 10; ++t)
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[t][i] = 0.f;
  if (cc < CC && seg < nseg && ls < pr) {
    const int c0 = cc * VEC;
    const int ws = (int)(seg % WS);
    const long long u = seg / WS;
    const int h = (int)(u % H), n = (int)(u / H);
    const int w0 = ws * DWG_SEG, w1 = min(W, w0 + DWG_SEG);
    // sliding 3x3 window: columns a = w-1, b = w, c = w+1 of rows h-1, h, h+1; every step loads one new
    // column (3 vectors) and one gradient vector instead of nine inputs
    const T* xr[3];
    bool rv[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int hh = h + r - 1;
      rv[r] = (unsigned)hh < (unsigned)H;
      xr[r] = x + ((size_t)n * H + (rv[r] ? hh : h)) * W * ldx + c0;
    }
    auto column = [&](float (&col)[3][VEC], int ww) {   // clamped address + select: the three loads issue together
      const bool cok = (unsigned)ww < (unsigned)W;
      const size_t off = (size_t)min(max(ww, 0), W - 1) * ldx;
      Vec16<T> raw[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) raw[r] = ld16(xr[r] + off);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int i = 0; i < VEC; ++i) col[r][i] = (rv[r] && cok) ? (float)raw[r].v[i] : 0.f;
    };
    const T* gr = g + ((size_t)n * H + h) * W * ldg + c0;
    // (loading column w+2 and the next gradient one pixel ahead of their use was measured slower: 1.56 vs 1.45 ms
    // per missformer step, 217 vs 182 VGPRs)
    auto pixel = [&](const float (&a)[3][VEC], const float (&b)[3][VEC], float (&c)[3][VEC], int w) {
      column(c, w + 1);
      float gv[VEC];
      load_f(gr + (size_t)w * ldg, gv);
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[9][i] += gv[i];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          acc[3 * r + 0][i] = fmaf(gv[i], a[r][i], acc[3 * r + 0][i]);
          acc[3 * r + 1][i] = fmaf(gv[i], b[r][i], acc[3 * r + 1][i]);
          acc[3 * r + 2][i] = fmaf(gv[i], c[r][i], acc[3 * r + 2][i]);
        }
    };
    float A[3][VEC], B[3][VEC], Cc[3][VEC];
    column(A, w0 - 1);
    column(B, w0);
    for (int w = w0; w < w1; w += 3) {
      pixel(A, B, Cc, w);
      if (w + 1 < w1) pixel(B, Cc, A, w + 1);
      if (w + 2 < w1) pixel(Cc, A, B, w + 2);
    }
  }
  if (ls < pr) {
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
      for (int i = 0; i < VEC; ++i) red[((ls * ccb + lc) * 10 + t) * VEC + i] = acc[t][i];
  }
  __syncthreads();
  if (ls == 0 && cc < CC) {
    for (int s = 1; s < pr; ++s)
#pragma unroll
      for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[t][i] += red[((s * ccb + lc) * 10 + t) * VEC + i];
    float* o = part + (size_t)blockIdx.x * 10 * C + cc * VEC;
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
      for (int i = 0; i < VEC; ++i) o[(size_t)t * C + i] = acc[t][i];
  }
}

// ---------------------------------------------------------------------------------------------
// space-to-depth: d[n, ho, wo, (ty*r + tx)*C + c] = s[n, ho*r + ty, wo*r + tx, c]   (inverse: the same
// index map with source and destination exchanged)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void space_to_depth_kernel(const T* __restrict__ src, int lds_, T* __restrict__ dst,
                                                             int ldd, int N, int Ho, int Wo, int C, int r, int inverse) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int CC = C / VEC;
  const long long total = (long long)N * Ho * r * Wo * r * CC;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c0 = (int)(idx % CC) * VEC;
    long long u = idx / CC;
    const int w = (int)(u % (Wo * r));
    u /= Wo * r;
    const int h = (int)(u % (Ho * r)), n = (int)(u / (Ho * r));
    const size_t fine = (((size_t)n * Ho * r + h) * Wo * r + w);
    const size_t coarse = (((size_t)n * Ho + h / r) * Wo + w / r);
    const int tap = (h % r) * r + (w % r);
    if (inverse) st16(dst + fine * ldd + c0, ld16(src + coarse * lds_ + (size_t)tap * C + c0));
    else st16(dst + coarse * ldd + (size_t)tap * C + c0, ld16(src + fine * lds_ + c0));
  }
}

// im2col of the NCHW fp32 input for a k x k convolution with stride s and zero padding p:
// out[(n, ho, wo)][(kh*k + kw)*C + c], columns >= k*k*C are zero.
template <typename T>
__global__ __launch_bounds__(256) void im2col_nchw_kernel(const float* __restrict__ x, int N, int C, int H, int W, int k,
                                                          int s, int p, int Ho, int Wo, int Kpad, T* __restrict__ out) {
  const int K = k * k * C;
  const long long total = (long long)N * Ho * Wo * Kpad;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int kk = (int)(idx % Kpad);
    const long long pix = idx / Kpad;
    float v = 0.f;
    if (kk < K) {
      const int c = kk % C, tap = kk / C, kh = tap / k, kw = tap - kh * k;
      const int j = (int)(pix % Wo);
      const long long t = pix / Wo;
      const int i = (int)(t % Ho), b = (int)(t / Ho);
      const int hh = i * s + kh - p, ww = j * s + kw - p;
      if ((unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W) v = x[(((size_t)b * C + c) * H + hh) * W + ww];
    }
    out[idx] = (T)v;
  }
}

// out[i] = T(sum_r part[r][i])  (fixed order)
template <typename T>
__global__ __launch_bounds__(256) void sum_parts_kernel(const float* __restrict__ part, int rows, long long n,
                                                        T* __restrict__ out) {
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n;
       i += (long long)gridDim.x * blockDim.x * 4) {
    f32x4 s = *reinterpret_cast<const f32x4*>(part + i);
    for (int r = 1; r < rows; ++r) s += *reinterpret_cast<const f32x4*>(part + (size_t)r * n + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) out[i + e] = (T)s[e];
  }
}

// ---------------------------------------------------------------------------------------------
// Spatial-reduction attention.  q [B][N][ldq], head h = columns h*64 .. h*64+63; keys and values are rows of
// the kv tensor: key j of image b is row ((j / kps) * B + b) * kps + j % kps  (kps = NK for one [B][NK] block;
// the bridge's keys are four blocks of kps rows, one per scale, each stored [B][kps]).  lse is kept in
// log2 units of the scaled scores: p = exp2(s * scale * log2(e) - lse).
// ---------------------------------------------------------------------------------------------
struct SraArgs {
  const void *q, *k, *v, *o, *go;   // go: gradient of o (backward)
  void *out, *dq;                   // forward output / dQ
  float *lse, *delta, *ws;          // [B][heads][N] each; ws: dK/dV partials [chunk][kv rows][ldws]
  int B, N, NK, heads, kps;
  int ldq, ldk, ldv, ldo, ldgo, lddq, ldws;
  int qc;                            // queries per chunk of the dK/dV kernel
  long long ws_chunk;                // floats per chunk of ws
  float scale;
};

constexpr int SD = 64;   // head dim
constexpr int TS = 72;   // LDS row stride in bf16 elements (144 B: 16-byte row reads of 16 lanes hit 16 distinct bank quads)

__device__ __forceinline__ size_t kv_row(const SraArgs& a, int b, int j) {
  return ((size_t)(j / a.kps) * a.B + b) * a.kps + j % a.kps;
}

typedef short s16x4 __attribute__((ext_vector_type(4)));

// A operand (32 rows d = col0 + l31, 16 contraction indices) of v_mfma_f32_32x32x16_bf16 read TRANSPOSED from a
// row-major [row = contraction index][TS] LDS tile: contraction slots 8*lh + {0..3} are tile rows row0 + {0..3},
// slots 8*lh + {4..7} rows row0 + 8 + {0..3} (row0 already holds the lane half's offset 4*lh) - the order in which
// the 32x32 accumulator of the previous product holds its rows, so that accumulator feeds the B operand as is.
// ds_read_b64_tr_b16: a group of 16 lanes reads a 4-row x 16-column block; lane 4q+p supplies the address of row q,
// columns 4p..4p+3 and lane i receives column i of the four rows.
__device__ __forceinline__ bf16x8 tr_operand(const bf16_t* tile, int row0, int col0, int lane) {
  const int i = lane & 15, cb = col0 + 16 * ((lane >> 4) & 1);
  const bf16_t* p = tile + (row0 + (i >> 2)) * TS + cb + 4 * (i & 3);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 8 * TS));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, both);
}

__device__ __forceinline__ bf16x8 row_operand(const bf16_t* tile, int row, int s, int lh) {
  return *reinterpret_cast<const bf16x8*>(tile + row * TS + 16 * s + 8 * lh);
}

// B operand rows of this lane's query / key straight from global memory (zero beyond `valid`)
__device__ __forceinline__ void load_b_operand(const bf16_t* rowp, bool valid, int lh, bf16x8* f) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (valid) f[s] = *reinterpret_cast<const bf16x8*>(rowp + 16 * s + 8 * lh);
    else
#pragma unroll
      for (int e = 0; e < 8; ++e) f[s][e] = (bf16_t)0.f;
  }
}

__device__ __forceinline__ f32x16 zero_acc() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}

// store a transposed accumulator pair (rows d of two 32-row tiles, column = this lane's token) as token-major bf16
__device__ __forceinline__ void store_t_tiles(bf16_t* rowp, int lh, const f32x16& t0, const f32x16& t1, float mul) {
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      bf16x4 o4;
#pragma unroll
      for (int e = 0; e < 4; ++e) o4[e] = (bf16_t)((dt ? t1 : t0)[4 * q4 + e] * mul);
      *reinterpret_cast<bf16x4*>(rowp + 32 * dt + 8 * q4 + 4 * lh) = o4;
    }
}

__global__ __launch_bounds__(256) void sra_fwd_mfma_kernel(const SraArgs a) {
  __shared__ __attribute__((aligned(16))) bf16_t sK[32 * TS];
  __shared__ __attribute__((aligned(16))) bf16_t sV[32 * TS];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int qi = blockIdx.x * 128 + 32 * w + l31;
  const bf16_t* Q = (const bf16_t*)a.q;
  const bf16_t* K = (const bf16_t*)a.k;
  const bf16_t* V = (const bf16_t*)a.v;
  bf16x8 qf[4];
  load_b_operand(Q + ((size_t)b * a.N + min(qi, a.N - 1)) * a.ldq + h * SD, qi < a.N, lh, qf);
  const int srow = tid >> 3, sch = tid & 7;
  uint4 rk, rv;
  auto fetch = [&](int kb) {
    const int key = kb * 32 + srow;
    rk = make_uint4(0, 0, 0, 0);
    rv = rk;
    if (key < a.NK) {
      const size_t row = kv_row(a, b, key);
      rk = *reinterpret_cast<const uint4*>(K + row * a.ldk + h * SD + 8 * sch);
      rv = *reinterpret_cast<const uint4*>(V + row * a.ldv + h * SD + 8 * sch);
    }
  };
  f32x16 o0 = zero_acc(), o1 = zero_acc();
  float m = -INFINITY, l = 0.f;
  const float c = a.scale * 1.4426950408889634f;
  const int nkb = (a.NK + 31) / 32;
  fetch(0);
  for (int kb = 0; kb < nkb; ++kb) {
    __syncthreads();
    *reinterpret_cast<uint4*>(sK + srow * TS + 8 * sch) = rk;
    *reinterpret_cast<uint4*>(sV + srow * TS + 8 * sch) = rv;
    __syncthreads();
    if (kb + 1 < nkb) fetch(kb + 1);
    f32x16 st = zero_acc();
#pragma unroll
    for (int s = 0; s < 4; ++s)
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_operand(sK, l31, s, lh), qf[s], st, 0, 0, 0);
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float sv = key < a.NK ? st[r] * c : -INFINITY;
      st[r] = sv;
      mx = fmaxf(mx, sv);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mn = fmaxf(m, mx);   // finite: every block holds at least one real key
    const float alpha = __builtin_amdgcn_exp2f(m - mn);
    m = mn;
    l *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      o0[r] *= alpha;
      o1[r] *= alpha;
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8 pf;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bf16_t pb = (bf16_t)__builtin_amdgcn_exp2f(st[8 * s2 + e] - m);
        l += (float)pb;   // normalise by what is actually multiplied
        pf[e] = pb;
      }
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand(sV, 16 * s2 + 4 * lh, 0, lane), pf, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand(sV, 16 * s2 + 4 * lh, 32, lane), pf, o1, 0, 0, 0);
    }
  }
  l += __shfl_xor(l, 32);
  if (qi < a.N) {
    store_t_tiles((bf16_t*)a.out + ((size_t)b * a.N + qi) * a.ldo + h * SD, lh, o0, o1, 1.f / l);
    if (lh == 0) a.lse[((size_t)b * a.heads + h) * a.N + qi] = m + __log2f(l);
  }
}

// dQ (and delta = rowsum(dO * O), kept for the dK/dV kernel): the forward's loop with dP^T = V dO^T,
// dS^T = P^T (dP^T - delta) * scale and dQ^T += K^T dS^T.
__global__ __launch_bounds__(256) void sra_bwd_dq_mfma_kernel(const SraArgs a) {
  __shared__ __attribute__((aligned(16))) bf16_t sK[32 * TS];
  __shared__ __attribute__((aligned(16))) bf16_t sV[32 * TS];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int qi = blockIdx.x * 128 + 32 * w + l31;
  const bool qv = qi < a.N;
  const size_t qrow = (size_t)b * a.N + min(qi, a.N - 1);
  const bf16_t* K = (const bf16_t*)a.k;
  const bf16_t* V = (const bf16_t*)a.v;
  bf16x8 qf[4], gf[4], of[4];
  load_b_operand((const bf16_t*)a.q + qrow * a.ldq + h * SD, qv, lh, qf);
  load_b_operand((const bf16_t*)a.go + qrow * a.ldgo + h * SD, qv, lh, gf);
  load_b_operand((const bf16_t*)a.o + qrow * a.ldo + h * SD, qv, lh, of);
  float delta = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) delta = fmaf((float)gf[s][e], (float)of[s][e], delta);
  delta += __shfl_xor(delta, 32);
  const size_t li = ((size_t)b * a.heads + h) * a.N + min(qi, a.N - 1);
  const float lse = qv ? a.lse[li] : INFINITY;
  if (qv && lh == 0) a.delta[li] = delta;
  const int srow = tid >> 3, sch = tid & 7;
  uint4 rk, rv;
  auto fetch = [&](int kb) {
    const int key = kb * 32 + srow;
    rk = make_uint4(0, 0, 0, 0);
    rv = rk;
    if (key < a.NK) {
      const size_t row = kv_row(a, b, key);
      rk = *reinterpret_cast<const uint4*>(K + row * a.ldk + h * SD + 8 * sch);
      rv = *reinterpret_cast<const uint4*>(V + row * a.ldv + h * SD + 8 * sch);
    }
  };
  f32x16 d0 = zero_acc(), d1 = zero_acc();
  const float c = a.scale * 1.4426950408889634f;
  const int nkb = (a.NK + 31) / 32;
  fetch(0);
  for (int kb = 0; kb < nkb; ++kb) {
    __syncthreads();
    *reinterpret_cast<uint4*>(sK + srow * TS + 8 * sch) = rk;
    *reinterpret_cast<uint4*>(sV + srow * TS + 8 * sch) = rv;
    __syncthreads();
    if (kb + 1 < nkb) fetch(kb + 1);
    f32x16 st = zero_acc(), dp = zero_acc();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_operand(sK, l31, s, lh), qf[s], st, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_operand(sV, l31, s, lh), gf[s], dp, 0, 0, 0);
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8 df;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int r = 8 * s2 + e;
        const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float p = key < a.NK ? __builtin_amdgcn_exp2f(st[r] * c - lse) : 0.f;
        df[e] = (bf16_t)(p * (dp[r] - delta) * a.scale);
      }
      d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand(sK, 16 * s2 + 4 * lh, 0, lane), df, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand(sK, 16 * s2 + 4 * lh, 32, lane), df, d1, 0, 0, 0);
    }
  }
  if (qv) store_t_tiles((bf16_t*)a.dq + qrow * a.lddq + h * SD, lh, d0, d1, 1.f);
}

// dK / dV: a wave owns 32 keys (its K and V rows are the B operands for the whole loop) and walks the
// queries of its chunk in blocks of 32 staged in LDS: S = Q K^T and dP = dO V^T with the query as the
// accumulator row, dV^T += dO^T P, dK^T += Q^T dS.  Partials per chunk go to ws, summed by sum_parts_kernel.
__global__ __launch_bounds__(256) void sra_bwd_dkv_mfma_kernel(const SraArgs a) {
  __shared__ __attribute__((aligned(16))) bf16_t sQ[32 * TS];
  __shared__ __attribute__((aligned(16))) bf16_t sG[32 * TS];
  __shared__ float sL[32], sD[32];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int kgroups = (a.NK + 127) / 128;
  const int b = blockIdx.z, h = blockIdx.y / kgroups, kg = blockIdx.y % kgroups;
  const int kj = kg * 128 + 32 * w + l31;
  const bool kvalid = kj < a.NK;
  const size_t krow = kv_row(a, b, min(kj, a.NK - 1));
  bf16x8 kf[4], vf[4];
  load_b_operand((const bf16_t*)a.k + krow * a.ldk + h * SD, kvalid, lh, kf);
  load_b_operand((const bf16_t*)a.v + krow * a.ldv + h * SD, kvalid, lh, vf);
  const bf16_t* Q = (const bf16_t*)a.q;
  const bf16_t* G = (const bf16_t*)a.go;
  const int q_begin = blockIdx.x * a.qc, q_end = min(a.N, q_begin + a.qc);
  const int srow = tid >> 3, sch = tid & 7;
  uint4 rq, rg;
  float rl = 0.f, rd = 0.f;
  auto fetch = [&](int q0) {
    const int qi = q0 + srow;
    rq = make_uint4(0, 0, 0, 0);
    rg = rq;
    if (qi < q_end) {
      const size_t row = (size_t)b * a.N + qi;
      rq = *reinterpret_cast<const uint4*>(Q + row * a.ldq + h * SD + 8 * sch);
      rg = *reinterpret_cast<const uint4*>(G + row * a.ldgo + h * SD + 8 * sch);
    }
    if (tid < 32) {
      const int qj = q0 + tid;
      const size_t li = ((size_t)b * a.heads + h) * a.N + min(qj, a.N - 1);
      rl = qj < q_end ? a.lse[li] : INFINITY;   // padded query: p = exp2(-inf) = 0
      rd = qj < q_end ? a.delta[li] : 0.f;
    }
  };
  f32x16 dk0 = zero_acc(), dk1 = zero_acc(), dv0 = zero_acc(), dv1 = zero_acc();
  const float c = a.scale * 1.4426950408889634f;
  if (q_begin < q_end) fetch(q_begin);
  for (int q0 = q_begin; q0 < q_end; q0 += 32) {
    __syncthreads();
    *reinterpret_cast<uint4*>(sQ + srow * TS + 8 * sch) = rq;
    *reinterpret_cast<uint4*>(sG + srow * TS + 8 * sch) = rg;
    if (tid < 32) {
      sL[tid] = rl;
      sD[tid] = rd;
    }
    __syncthreads();
    if (q0 + 32 < q_end) fetch(q0 + 32);
    f32x16 st = zero_acc(), dp = zero_acc();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_operand(sQ, l31, s, lh), kf[s], st, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_operand(sG, l31, s, lh), vf[s], dp, 0, 0, 0);
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8 pf, df;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int r = 8 * s2 + e;
        const int qq = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float p = kvalid ? __builtin_amdgcn_exp2f(st[r] * c - sL[qq]) : 0.f;
        pf[e] = (bf16_t)p;
        df[e] = (bf16_t)(p * (dp[r] - sD[qq]) * a.scale);
      }
      dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand(sG, 16 * s2 + 4 * lh, 0, lane), pf, dv0, 0, 0, 0);
      dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand(sG, 16 * s2 + 4 * lh, 32, lane), pf, dv1, 0, 0, 0);
      dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand(sQ, 16 * s2 + 4 * lh, 0, lane), df, dk0, 0, 0, 0);
      dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand(sQ, 16 * s2 + 4 * lh, 32, lane), df, dk1, 0, 0, 0);
    }
  }
  if (kvalid) {
    // ws row = the kv row, columns [0, C) dK and [C, 2C) dV with C = heads * 64
    float* o = a.ws + (size_t)blockIdx.x * a.ws_chunk + krow * a.ldws + h * SD + 4 * lh;
    const int C = a.heads * SD;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        f32x4 k4, v4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          k4[e] = (dt ? dk1 : dk0)[4 * q4 + e];
          v4[e] = (dt ? dv1 : dv0)[4 * q4 + e];
        }
        *reinterpret_cast<f32x4*>(o + 32 * dt + 8 * q4) = k4;
        *reinterpret_cast<f32x4*>(o + C + 32 * dt + 8 * q4) = v4;
      }
  }
}

// ---- fp32 run mode: scalar kernels, one thread per query (forward, dQ) or per key (dK/dV) -----------------
constexpr int SF_KB = 32;   // keys (or queries) staged per block

__global__ __launch_bounds__(128) void sra_fwd_f32_kernel(const SraArgs a) {
  __shared__ float sK[SF_KB][SD], sV[SF_KB][SD];
  const int b = blockIdx.z, h = blockIdx.y, qi = blockIdx.x * 128 + threadIdx.x;
  const bool qv = qi < a.N;
  const float* qp = (const float*)a.q + ((size_t)b * a.N + min(qi, a.N - 1)) * a.ldq + h * SD;
  float q[SD], o[SD];
#pragma unroll
  for (int d = 0; d < SD; ++d) {
    q[d] = qp[d] * a.scale;
    o[d] = 0.f;
  }
  float m = -INFINITY, l = 0.f;
  for (int k0 = 0; k0 < a.NK; k0 += SF_KB) {
    __syncthreads();
    for (int i = threadIdx.x; i < SF_KB * SD; i += 128) {
      const int j = i / SD, d = i % SD;
      float kv = 0.f, vv = 0.f;
      if (k0 + j < a.NK) {
        const size_t row = kv_row(a, b, k0 + j);
        kv = ((const float*)a.k)[row * a.ldk + h * SD + d];
        vv = ((const float*)a.v)[row * a.ldv + h * SD + d];
      }
      sK[j][d] = kv;
      sV[j][d] = vv;
    }
    __syncthreads();
    const int nk = min(SF_KB, a.NK - k0);
    for (int j = 0; j < nk; ++j) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < SD; ++d) s = fmaf(q[d], sK[j][d], s);
      const float mn = fmaxf(m, s);
      const float alpha = __expf(m - mn), p = __expf(s - mn);
      m = mn;
      l = l * alpha + p;
#pragma unroll
      for (int d = 0; d < SD; ++d) o[d] = fmaf(o[d], alpha, p * sV[j][d]);
    }
  }
  if (qv) {
    float* op = (float*)a.out + ((size_t)b * a.N + qi) * a.ldo + h * SD;
    const float inv = 1.f / l;
#pragma unroll
    for (int d = 0; d < SD; ++d) op[d] = o[d] * inv;
    a.lse[((size_t)b * a.heads + h) * a.N + qi] = (m + __logf(l)) * 1.4426950408889634f;
  }
}

__global__ __launch_bounds__(128) void sra_bwd_dq_f32_kernel(const SraArgs a) {
  __shared__ float sK[SF_KB][SD], sV[SF_KB][SD];
  const int b = blockIdx.z, h = blockIdx.y, qi = blockIdx.x * 128 + threadIdx.x;
  const bool qv = qi < a.N;
  const size_t qrow = (size_t)b * a.N + min(qi, a.N - 1);
  const float* qp = (const float*)a.q + qrow * a.ldq + h * SD;
  const float* gp = (const float*)a.go + qrow * a.ldgo + h * SD;
  const float* op = (const float*)a.o + qrow * a.ldo + h * SD;
  float q[SD], g[SD], dq[SD];
  float delta = 0.f;
#pragma unroll
  for (int d = 0; d < SD; ++d) {
    q[d] = qp[d];
    g[d] = gp[d];
    dq[d] = 0.f;
    delta = fmaf(g[d], op[d], delta);
  }
  const size_t li = ((size_t)b * a.heads + h) * a.N + min(qi, a.N - 1);
  const float lse = a.lse[li] * 0.6931471805599453f;
  if (qv) a.delta[li] = delta;
  for (int k0 = 0; k0 < a.NK; k0 += SF_KB) {
    __syncthreads();
    for (int i = threadIdx.x; i < SF_KB * SD; i += 128) {
      const int j = i / SD, d = i % SD;
      float kv = 0.f, vv = 0.f;
      if (k0 + j < a.NK) {
        const size_t row = kv_row(a, b, k0 + j);
        kv = ((const float*)a.k)[row * a.ldk + h * SD + d];
        vv = ((const float*)a.v)[row * a.ldv + h * SD + d];
      }
      sK[j][d] = kv;
      sV[j][d] = vv;
    }
    __syncthreads();
    const int nk = min(SF_KB, a.NK - k0);
    for (int j = 0; j < nk; ++j) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < SD; ++d) {
        s = fmaf(q[d], sK[j][d], s);
        dp = fmaf(g[d], sV[j][d], dp);
      }
      const float ds = __expf(s * a.scale - lse) * (dp - delta) * a.scale;
#pragma unroll
      for (int d = 0; d < SD; ++d) dq[d] = fmaf(ds, sK[j][d], dq[d]);
    }
  }
  if (qv) {
    float* dp_ = (float*)a.dq + qrow * a.lddq + h * SD;
#pragma unroll
    for (int d = 0; d < SD; ++d) dp_[d] = dq[d];
  }
}

// one thread per key; blockIdx.x = query chunk; PASS 0: dV, PASS 1: dK (each keeps k and one accumulator row)
template <int PASS>
__global__ __launch_bounds__(64) void sra_bwd_dkv_f32_kernel(const SraArgs a) {
  __shared__ float sQ[SF_KB][SD], sG[SF_KB][SD], sL[SF_KB], sD[SF_KB];
  const int kgroups = (a.NK + 63) / 64;
  const int b = blockIdx.z, h = blockIdx.y / kgroups, kj = (blockIdx.y % kgroups) * 64 + threadIdx.x;
  const bool kvalid = kj < a.NK;
  const size_t krow = kv_row(a, b, min(kj, a.NK - 1));
  const float* kp = (const float*)a.k + krow * a.ldk + h * SD;
  const float* vp = (const float*)a.v + krow * a.ldv + h * SD;
  float k[SD], v[SD], acc[SD];
#pragma unroll
  for (int d = 0; d < SD; ++d) {
    k[d] = kp[d];
    v[d] = PASS ? vp[d] : 0.f;
    acc[d] = 0.f;
  }
  const int q_begin = blockIdx.x * a.qc, q_end = min(a.N, q_begin + a.qc);
  for (int q0 = q_begin; q0 < q_end; q0 += SF_KB) {
    __syncthreads();
    const int nq = min(SF_KB, q_end - q0);
    for (int i = threadIdx.x; i < SF_KB * SD; i += 64) {
      const int j = i / SD, d = i % SD;
      float qv = 0.f, gv = 0.f;
      if (j < nq) {
        const size_t row = (size_t)b * a.N + q0 + j;
        qv = ((const float*)a.q)[row * a.ldq + h * SD + d];
        gv = ((const float*)a.go)[row * a.ldgo + h * SD + d];
      }
      sQ[j][d] = qv;
      sG[j][d] = gv;
    }
    if (threadIdx.x < SF_KB && threadIdx.x < nq) {
      const size_t li = ((size_t)b * a.heads + h) * a.N + q0 + threadIdx.x;
      sL[threadIdx.x] = a.lse[li] * 0.6931471805599453f;
      sD[threadIdx.x] = a.delta[li];
    }
    __syncthreads();
    for (int j = 0; j < nq; ++j) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < SD; ++d) {
        s = fmaf(sQ[j][d], k[d], s);
        if (PASS) dp = fmaf(sG[j][d], v[d], dp);
      }
      const float p = __expf(s * a.scale - sL[j]);
      if (PASS) {
        const float ds = p * (dp - sD[j]) * a.scale;
#pragma unroll
        for (int d = 0; d < SD; ++d) acc[d] = fmaf(ds, sQ[j][d], acc[d]);
      } else {
#pragma unroll
        for (int d = 0; d < SD; ++d) acc[d] = fmaf(p, sG[j][d], acc[d]);
      }
    }
  }
  if (kvalid) {
    float* o = a.ws + (size_t)blockIdx.x * a.ws_chunk + krow * a.ldws + h * SD + (PASS ? 0 : a.heads * SD);
#pragma unroll
    for (int d = 0; d < SD; ++d) o[d] = acc[d];
  }
}

int sra_check(const char* fn, const uz_sra_desc* d) {
  UZ_REQUIRE(d != nullptr, "%s: null descriptor", fn);
  UZ_REQUIRE(d->dtype == UZ_F32 || d->dtype == UZ_BF16, "%s: bad dtype", fn);
  const int vec = d->dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(d->B > 0 && d->N > 0 && d->NK > 0 && d->heads > 0 && d->B <= 65535 && d->heads * ((d->NK + 63) / 64) <= 65535,
             "%s: bad shape B=%d N=%d NK=%d heads=%d", fn, d->B, d->N, d->NK, d->heads);
  UZ_REQUIRE(d->head_dim == SD, "%s: head_dim %d unsupported (the kernels are built for 64)", fn, d->head_dim);
  UZ_REQUIRE(d->kps > 0 && d->NK % d->kps == 0, "%s: NK=%d is not a multiple of the segment length %d", fn, d->NK, d->kps);
  const int C = d->heads * SD;
  UZ_REQUIRE(d->ldq >= C && d->ldk >= C && d->ldv >= C && d->ldo >= C && d->ldq % vec == 0 && d->ldk % vec == 0 &&
                 d->ldv % vec == 0 && d->ldo % vec == 0, "%s: bad leading dimensions", fn);
  UZ_REQUIRE((long long)d->B * d->N < (1LL << 31) && (long long)d->B * d->NK < (1LL << 31), "%s: too many rows", fn);
  return UZ_OK;
}

void sra_fill(const uz_sra_desc* d, SraArgs* a) {
  a->B = d->B; a->N = d->N; a->NK = d->NK; a->heads = d->heads; a->kps = d->kps;
  a->ldq = d->ldq; a->ldk = d->ldk; a->ldv = d->ldv; a->ldo = d->ldo;
  a->scale = d->scale;
}

// query chunks of the dK/dV kernel: enough workgroups to fill the chip, chunks of at least 64 queries
void sra_chunks(const uz_sra_desc* d, int* qc, int* nchunks) {
  const int per = d->dtype == UZ_BF16 ? 128 : 64;
  const long long base = (long long)d->B * d->heads * ((d->NK + per - 1) / per);
  const int nqb = (d->N + 31) / 32;
  long long want = (4LL * UZ_NUM_CU + base - 1) / base;
  if (want < 1) want = 1;
  if (want > (nqb + 1) / 2) want = (nqb + 1) / 2;
  if (want < 1) want = 1;
  const int blocks = (int)((nqb + want - 1) / want);
  *qc = blocks * 32;
  *nchunks = (d->N + *qc - 1) / *qc;
}

// floats of the delta vector, rounded so the partial tiles that follow stay 64-byte aligned
long long sra_delta_floats(const uz_sra_desc* d) { return (((long long)d->B * d->heads * d->N + 15) / 16) * 16; }

}  // namespace

extern "C" int uz_gelu_fwd(int dtype, const void* x, int ldx, void* y, int ldy, long long P, int C, void* stream) {
  UZ_REQUIRE(dtype == UZ_F32 || dtype == UZ_BF16, "uz_gelu_fwd: bad dtype");
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(x && y && P > 0 && C > 0 && C % vec == 0 && ldx % vec == 0 && ldy % vec == 0 && ldx >= C && ldy >= C, "uz_gelu_fwd: bad shape");
  const dim3 grid(grid_cap(P * (C / vec), 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UZ_BF16) hipLaunchKernelGGL((gelu_kernel<bf16_t, false>), grid, block, 0, s, (const bf16_t*)x, ldx, (const bf16_t*)nullptr, 0, (bf16_t*)y, ldy, P, C);
  else hipLaunchKernelGGL((gelu_kernel<float, false>), grid, block, 0, s, (const float*)x, ldx, (const float*)nullptr, 0, (float*)y, ldy, P, C);
  UZ_LAUNCH_CHECK("uz_gelu_fwd");
  return UZ_OK;
}

extern "C" int uz_gelu_bwd(int dtype, const void* x, int ldx, const void* g, int ldg, void* dx, int lddx, long long P,
                           int C, void* stream) {
  UZ_REQUIRE(dtype == UZ_F32 || dtype == UZ_BF16, "uz_gelu_bwd: bad dtype");
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(x && g && dx && P > 0 && C > 0 && C % vec == 0 && ldx % vec == 0 && ldg % vec == 0 && lddx % vec == 0 &&
                 ldx >= C && ldg >= C && lddx >= C, "uz_gelu_bwd: bad shape");
  const dim3 grid(grid_cap(P * (C / vec), 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UZ_BF16) hipLaunchKernelGGL((gelu_kernel<bf16_t, true>), grid, block, 0, s, (const bf16_t*)x, ldx, (const bf16_t*)g, ldg, (bf16_t*)dx, lddx, P, C);
  else hipLaunchKernelGGL((gelu_kernel<float, true>), grid, block, 0, s, (const float*)x, ldx, (const float*)g, ldg, (float*)dx, lddx, P, C);
  UZ_LAUNCH_CHECK("uz_gelu_bwd");
  return UZ_OK;
}

extern "C" int uz_dwconv3x3(int dtype, const void* x, int ldx, const float* w_taps, const float* bias, void* y, int ldy,
                            int N, int H, int W, int C, int flags, void* stream) {
  UZ_REQUIRE(dtype == UZ_F32 || dtype == UZ_BF16, "uz_dwconv3x3: bad dtype");
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(x && w_taps && y && N > 0 && H > 0 && W > 0 && C > 0 && C % vec == 0, "uz_dwconv3x3: bad shape");
  UZ_REQUIRE(ldx % vec == 0 && ldy % vec == 0 && ldx >= C && ldy >= C && (flags & ~3) == 0, "uz_dwconv3x3: bad strides / flags");
  const long long total = (long long)N * H * ((W + DW_SW - 1) / DW_SW) * (C / vec);
  const dim3 grid(grid_cap(total, 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UZ_BF16) hipLaunchKernelGGL((dwconv3x3_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)x, ldx, w_taps, bias, (bf16_t*)y, ldy, N, H, W, C, flags);
  else hipLaunchKernelGGL((dwconv3x3_kernel<float>), grid, block, 0, s, (const float*)x, ldx, w_taps, bias, (float*)y, ldy, N, H, W, C, flags);
  UZ_LAUNCH_CHECK("uz_dwconv3x3");
  return UZ_OK;
}

static void dwg_geometry(int dtype, int N, int H, int W, int C, int* ccb, int* pr, int* gx, int* gy) {
  const int CC = C / (dtype == UZ_BF16 ? 8 : 4);
  const int threads = dtype == UZ_BF16 ? 128 : 256;   // 40 KB of staging either way
  *ccb = CC < threads ? CC : threads;
  *pr = threads / *ccb;
  if (*pr > 8) *pr = 8;
  const long long nseg = (long long)N * H * ((W + DWG_SEG - 1) / DWG_SEG);
  *gx = (int)((nseg + *pr - 1) / *pr);
  *gy = (CC + *ccb - 1) / *ccb;
}

extern "C" int uz_dwconv3x3_wgrad_rows(int dtype, int N, int H, int W, int C) {
  int ccb, pr, gx, gy;
  if ((dtype != UZ_F32 && dtype != UZ_BF16) || N <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
  dwg_geometry(dtype, N, H, W, C, &ccb, &pr, &gx, &gy);
  return gx;
}

extern "C" int uz_dwconv3x3_wgrad(int dtype, const void* x, int ldx, const void* g, int ldg, float* part, int N, int H,
                                  int W, int C, void* stream) {
  UZ_REQUIRE(dtype == UZ_F32 || dtype == UZ_BF16, "uz_dwconv3x3_wgrad: bad dtype");
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(x && g && part && N > 0 && H > 0 && W > 0 && C > 0 && C % vec == 0, "uz_dwconv3x3_wgrad: bad shape");
  UZ_REQUIRE(ldx % vec == 0 && ldg % vec == 0 && ldx >= C && ldg >= C, "uz_dwconv3x3_wgrad: bad strides");
  int ccb, pr, gx, gy;
  dwg_geometry(dtype, N, H, W, C, &ccb, &pr, &gx, &gy);
  const size_t shm = (size_t)pr * ccb * 10 * vec * sizeof(float);
  UZ_REQUIRE(shm <= 64 * 1024, "uz_dwconv3x3_wgrad: staging exceeds 64 KiB");
  const dim3 grid(gx, gy), block(ccb * pr);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UZ_BF16) hipLaunchKernelGGL((dwconv3x3_wgrad_kernel<bf16_t>), grid, block, shm, s, (const bf16_t*)x, ldx, (const bf16_t*)g, ldg, part, N, H, W, C, ccb, pr);
  else hipLaunchKernelGGL((dwconv3x3_wgrad_kernel<float>), grid, block, shm, s, (const float*)x, ldx, (const float*)g, ldg, part, N, H, W, C, ccb, pr);
  UZ_LAUNCH_CHECK("uz_dwconv3x3_wgrad");
  return UZ_OK;
}

extern "C" int uz_space_to_depth(int dtype, const void* src, int lds_, void* dst, int ldd, int N, int Ho, int Wo, int C,
                                 int r, int inverse, void* stream) {
  UZ_REQUIRE(dtype == UZ_F32 || dtype == UZ_BF16, "uz_space_to_depth: bad dtype");
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(src && dst && N > 0 && Ho > 0 && Wo > 0 && C > 0 && r > 0 && C % vec == 0, "uz_space_to_depth: bad shape");
  const int lfine = inverse ? ldd : lds_, lcoarse = inverse ? lds_ : ldd;
  UZ_REQUIRE(lfine % vec == 0 && lcoarse % vec == 0 && lfine >= C && lcoarse >= r * r * C, "uz_space_to_depth: bad strides");
  const long long total = (long long)N * Ho * r * Wo * r * (C / vec);
  const dim3 grid(grid_cap(total, 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UZ_BF16) hipLaunchKernelGGL((space_to_depth_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)src, lds_, (bf16_t*)dst, ldd, N, Ho, Wo, C, r, inverse);
  else hipLaunchKernelGGL((space_to_depth_kernel<float>), grid, block, 0, s, (const float*)src, lds_, (float*)dst, ldd, N, Ho, Wo, C, r, inverse);
  UZ_LAUNCH_CHECK("uz_space_to_depth");
  return UZ_OK;
}

extern "C" int uz_im2col_nchw(int dtype, const float* x_nchw, int N, int C, int H, int W, int k, int stride, int pad,
                              int Kpad, void* out, void* stream) {
  UZ_REQUIRE(dtype == UZ_F32 || dtype == UZ_BF16, "uz_im2col_nchw: bad dtype");
  UZ_REQUIRE(x_nchw && out && N > 0 && C > 0 && H > 0 && W > 0 && k > 0 && stride > 0 && pad >= 0, "uz_im2col_nchw: bad shape");
  UZ_REQUIRE(H + 2 * pad >= k && W + 2 * pad >= k && Kpad >= k * k * C, "uz_im2col_nchw: bad kernel / Kpad");
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  const long long total = (long long)N * Ho * Wo * Kpad;
  const dim3 grid(grid_cap(total, 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UZ_BF16) hipLaunchKernelGGL((im2col_nchw_kernel<bf16_t>), grid, block, 0, s, x_nchw, N, C, H, W, k, stride, pad, Ho, Wo, Kpad, (bf16_t*)out);
  else hipLaunchKernelGGL((im2col_nchw_kernel<float>), grid, block, 0, s, x_nchw, N, C, H, W, k, stride, pad, Ho, Wo, Kpad, (float*)out);
  UZ_LAUNCH_CHECK("uz_im2col_nchw");
  return UZ_OK;
}

extern "C" int uz_sra_fwd(const uz_sra_desc* d, const void* q, const void* k, const void* v, void* out, float* lse,
                          void* stream) {
  const int rc = sra_check("uz_sra_fwd", d);
  if (rc != UZ_OK) return rc;
  UZ_REQUIRE(q && k && v && out && lse, "uz_sra_fwd: null pointer");
  SraArgs a{};
  sra_fill(d, &a);
  a.q = q; a.k = k; a.v = v; a.out = out; a.lse = lse;
  const dim3 grid((d->N + 127) / 128, d->heads, d->B);
  hipStream_t s = (hipStream_t)stream;
  if (d->dtype == UZ_BF16) hipLaunchKernelGGL(sra_fwd_mfma_kernel, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(sra_fwd_f32_kernel, grid, dim3(128), 0, s, a);
  UZ_LAUNCH_CHECK("uz_sra_fwd");
  return UZ_OK;
}

extern "C" long long uz_sra_bwd_workspace_bytes(const uz_sra_desc* d) {
  if (sra_check("uz_sra_bwd_workspace_bytes", d) != UZ_OK) return -1;
  int qc, nchunks;
  sra_chunks(d, &qc, &nchunks);
  // delta [B][heads][N] + partial dK/dV [chunks][kv rows][2C]
  return (sra_delta_floats(d) + (long long)nchunks * d->B * d->NK * 2 * d->heads * SD) * 4;
}

extern "C" int uz_sra_bwd(const uz_sra_desc* d, const void* q, const void* k, const void* v, const void* o,
                          const float* lse, const void* go, int ldgo, void* dq, int lddq, void* dkv, int lddkv,
                          void* workspace, void* stream) {
  const int rc = sra_check("uz_sra_bwd", d);
  if (rc != UZ_OK) return rc;
  const int vec = d->dtype == UZ_BF16 ? 8 : 4, C = d->heads * SD;
  UZ_REQUIRE(q && k && v && o && lse && go && dq && dkv && workspace, "uz_sra_bwd: null pointer");
  UZ_REQUIRE(ldgo >= C && lddq >= C && ldgo % vec == 0 && lddq % vec == 0, "uz_sra_bwd: bad gradient strides");
  UZ_REQUIRE(lddkv == 2 * C, "uz_sra_bwd: dkv must be a dense [B * NK][2 * heads * 64] tensor (lddkv = %d)", lddkv);
  SraArgs a{};
  sra_fill(d, &a);
  a.q = q; a.k = k; a.v = v; a.o = o; a.go = go; a.dq = dq; a.lse = const_cast<float*>(lse);
  a.ldgo = ldgo; a.lddq = lddq; a.ldws = 2 * C;
  int nchunks;
  sra_chunks(d, &a.qc, &nchunks);
  a.delta = (float*)workspace;
  a.ws = a.delta + sra_delta_floats(d);
  a.ws_chunk = (long long)d->B * d->NK * 2 * C;
  hipStream_t s = (hipStream_t)stream;
  const dim3 gq((d->N + 127) / 128, d->heads, d->B);
  if (d->dtype == UZ_BF16) {
    hipLaunchKernelGGL(sra_bwd_dq_mfma_kernel, gq, dim3(256), 0, s, a);
    UZ_LAUNCH_CHECK("uz_sra_bwd(dq)");
    hipLaunchKernelGGL(sra_bwd_dkv_mfma_kernel, dim3(nchunks, d->heads * ((d->NK + 127) / 128), d->B), dim3(256), 0, s, a);
    UZ_LAUNCH_CHECK("uz_sra_bwd(dkv)");
    hipLaunchKernelGGL((sum_parts_kernel<bf16_t>), dim3(grid_cap(a.ws_chunk / 4, 256)), dim3(256), 0, s, a.ws, nchunks, a.ws_chunk, (bf16_t*)dkv);
  } else {
    hipLaunchKernelGGL(sra_bwd_dq_f32_kernel, gq, dim3(128), 0, s, a);
    UZ_LAUNCH_CHECK("uz_sra_bwd(dq)");
    const dim3 gk(nchunks, d->heads * ((d->NK + 63) / 64), d->B);
    hipLaunchKernelGGL((sra_bwd_dkv_f32_kernel<0>), gk, dim3(64), 0, s, a);
    hipLaunchKernelGGL((sra_bwd_dkv_f32_kernel<1>), gk, dim3(64), 0, s, a);
    UZ_LAUNCH_CHECK("uz_sra_bwd(dkv)");
    hipLaunchKernelGGL((sum_parts_kernel<float>), dim3(grid_cap(a.ws_chunk / 4, 256)), dim3(256), 0, s, a.ws, nchunks, a.ws_chunk, (float*)dkv);
  }
  UZ_LAUNCH_CHECK("uz_sra_bwd(sum)");
  return UZ_OK;
}
