// Pieces of the dense token attention of U-Transformer / TransAttUNet around the batched matrix products
// (uz_gemm_nt for the NT products, uz_wgrad's one-tap kernel for the TN products):
//   * softmax over either axis of a batch of (rows, cols) score matrices, in place, and its gradient
//     (unet_zoo/models/unet_transformer.py:123,133-134: nn.Softmax(dim=1) of a (b, queries, keys) tensor normalises
//     every key's COLUMN over the queries; transatt_unet.py:35,47 and :101 normalise rows),
//   * F.adaptive_avg_pool2d on NHWC maps and its gradient (unet_transformer.py:196-198),
//   * row dot products and fp32 -> run-dtype conversion of the fp32 TN results.
// All of it is HBM-bound element work: 16-byte accesses, one pass where the mathematics allows it.
#include "uz_common.h"

namespace {

template <typename T> __device__ __forceinline__ float tof(T v) { return (float)v; }
// exponentials: the fp32 (parity) instantiations use the correctly rounded library function -- v_exp_f32 on a product
// with log2(e) carries a relative error of |x| * 6e-8, which the score matrices of a 4096-token softmax amplify --,
// the bf16 ones the hardware instruction (their results are rounded to 8 bits anyway)
template <typename T> __device__ __forceinline__ float uz_exp(float x) {
  if constexpr (sizeof(T) == 4) return expf(x);
  else return __expf(x);
}

// ---- softmax over the ROW axis of (rows, cols): every column is normalised ------------------------------------------
// Three launches, all of them streaming at full occupancy (a workgroup that walks a whole 4096-row strip twice leaves the
// CU with 8 waves and runs at 3.5 TB/s; this form reaches what the gradient kernel below reaches):
//   stats   : workgroup = (strip of 16 * VEC columns, matrix, slice of the rows); 16 threads across the strip, 16 row
//             groups, running maximum and running sum of exponentials per column, merged through LDS -> one (m, z) pair
//             per (slice, column)
//   combine : the slices of a column -> m and 1 / Z
//   apply   : s <- exp(scale * s - m[col]) / Z[col], elementwise
constexpr int SM_SLICE = 512;   // rows per stats workgroup

// keeps a 16-byte load unconditional (see pin16 in uz_gemm_dma.hip): the loaded registers pass through an empty asm, so a value
// that is only used on one path is not loaded under that path's branch -- where hipcc 7.2 ends the block with s_waitcnt vmcnt(0)
// and the loads of an unrolled group come one memory round trip after the other
template <typename T> __device__ __forceinline__ void ta_pin16(Vec16<T>& v) {
  unsigned* r = reinterpret_cast<unsigned*>(&v);
  asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_cols_stats_kernel(const T* s, int ld, long long sb, int rows, int cols, float scale,
                                                                 float* part) {   // part[b][slice][2][cols]
  constexpr int VEC = ElemTraits<T>::VEC;
  constexpr int SW = 16 * VEC;
  __shared__ float sm[16][SW], sz[16][SW];
  const int tid = threadIdx.x, cx = tid & 15, rg = tid >> 4;
  const int c0 = blockIdx.x * SW + cx * VEC;
  const T* base = s + (long long)blockIdx.y * sb;
  const int r_beg = blockIdx.z * SM_SLICE, r_end = min(rows, r_beg + SM_SLICE);
  float m[VEC], z[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { m[e] = -INFINITY; z[e] = 0.f; }
  if (c0 < cols) {
    for (int r = r_beg + rg; r < r_end; r += 64) {
      Vec16<T> v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)   // (unconditional: the slice's first row past its end; such a row enters as -inf below)
        v[u] = ld16(base + (long long)(r + 16 * u < r_end ? r + 16 * u : r_beg) * ld + c0);
#pragma unroll
      for (int u = 0; u < 4; ++u) ta_pin16(v[u]);
      // one rescale of the running sum per group of four rows: 5 exponentials per 4 elements instead of 8
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float x[4], mn = m[e];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          x[u] = (r + 16 * u < r_end) ? tof(v[u].v[e]) * scale : -INFINITY;
          mn = fmaxf(mn, x[u]);
        }
        float acc = z[e] * uz_exp<T>(m[e] - mn);      // (first group: 0 * exp(-inf) = 0)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += uz_exp<T>(x[u] - mn);   // exp(-inf) = 0 for the rows beyond the slice
        z[e] = acc;
        m[e] = mn;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) { sm[rg][cx * VEC + e] = m[e]; sz[rg][cx * VEC + e] = z[e]; }
  __syncthreads();
  if (tid < SW && blockIdx.x * SW + tid < cols) {
    float mm = -INFINITY;
#pragma unroll
    for (int g = 0; g < 16; ++g) mm = fmaxf(mm, sm[g][tid]);
    float zz = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const float mg = sm[g][tid];
      zz += mg == -INFINITY ? 0.f : sz[g][tid] * uz_exp<T>(mg - mm);
    }
    float* pp = part + ((long long)blockIdx.y * gridDim.z + blockIdx.z) * 2 * cols + blockIdx.x * SW + tid;
    pp[0] = mm;
    pp[cols] = zz;
  }
}

__global__ __launch_bounds__(256) void softmax_cols_combine_kernel(const float* part, int nslices, int cols, long long total,
                                                                   float* mz) {   // mz[b][2][cols] = (m, 1 / Z)
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long long b = i / cols;
  const int c = (int)(i % cols);
  const float* pp = part + b * nslices * 2 * cols + c;
  float mm = -INFINITY;
  for (int k = 0; k < nslices; ++k) mm = fmaxf(mm, pp[(long long)k * 2 * cols]);
  float zz = 0.f;
  for (int k = 0; k < nslices; ++k) {
    const float mk = pp[(long long)k * 2 * cols];
    zz += mk == -INFINITY ? 0.f : pp[(long long)k * 2 * cols + cols] * expf(mk - mm);
  }
  mz[b * 2 * cols + c] = mm;
  mz[b * 2 * cols + cols + c] = 1.f / zz;
}

// thread = one 16-byte column chunk, SM_ROWS consecutive rows: the column vectors stay in registers and no index needs a
// division (a grid-stride loop over flat chunk indices spends more VALU time in 64-bit div / mod than in the exponentials)
constexpr int SM_ROWS = 16;

template <typename T>
__global__ __launch_bounds__(256) void softmax_cols_apply_kernel(T* s, int ld, long long sb, int rows, int cols, float scale,
                                                                 const float* mz) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int c0 = (blockIdx.x * 256 + threadIdx.x) * VEC;
  if (c0 >= cols) return;
  const int b = blockIdx.z, r0 = blockIdx.y * SM_ROWS;
  const float* mp = mz + (long long)b * 2 * cols + c0;
  float mm[VEC], rz[VEC];
#pragma unroll
  for (int e = 0; e < VEC; e += 4) {
    *reinterpret_cast<f32x4*>(&mm[e]) = *reinterpret_cast<const f32x4*>(mp + e);
    *reinterpret_cast<f32x4*>(&rz[e]) = *reinterpret_cast<const f32x4*>(mp + cols + e);
  }
  T* p = s + (long long)b * sb + (long long)r0 * ld + c0;
  const int nr = min(SM_ROWS, rows - r0);
  for (int r = 0; r < nr; r += 4) {
    Vec16<T> v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = ld16(p + (long long)(r + u < nr ? r + u : 0) * ld);   // (unconditional: row 0 past the end, not stored)
#pragma unroll
    for (int u = 0; u < 4; ++u) ta_pin16(v[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (r + u < nr) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[u].v[e] = (T)(uz_exp<T>(tof(v[u].v[e]) * scale - mm[e]) * rz[e]);
        st16(p + (long long)(r + u) * ld, v[u]);
      }
  }
}

// column dot products dot[b][c] = sum_r a[r][c] * g[r][c] (only when the caller has no cheaper way to them)
template <typename T>
__global__ __launch_bounds__(256) void softmax_cols_dot_kernel(const T* a, const T* g, int ld, long long sb, int rows, int cols,
                                                               float* dot) {
  constexpr int VEC = ElemTraits<T>::VEC;
  constexpr int SW = 16 * VEC;
  __shared__ float sd[16][SW];
  const int tid = threadIdx.x, cx = tid & 15, rg = tid >> 4;
  const int c0 = blockIdx.x * SW + cx * VEC;
  const T* ab = a + (long long)blockIdx.y * sb;
  const T* gb = g + (long long)blockIdx.y * sb;
  float d[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) d[e] = 0.f;
  if (c0 < cols)
    for (int r = rg; r < rows; r += 16) {
      const Vec16<T> va = ld16(ab + (long long)r * ld + c0), vg = ld16(gb + (long long)r * ld + c0);
#pragma unroll
      for (int e = 0; e < VEC; ++e) d[e] = fmaf(tof(va.v[e]), tof(vg.v[e]), d[e]);
    }
#pragma unroll
  for (int e = 0; e < VEC; ++e) sd[rg][cx * VEC + e] = d[e];
  __syncthreads();
  if (tid < SW && blockIdx.x * SW + tid < cols) {
    float t = 0.f;
#pragma unroll
    for (int g2 = 0; g2 < 16; ++g2) t += sd[g2][tid];
    dot[(long long)blockIdx.y * cols + blockIdx.x * SW + tid] = t;
  }
}

// dS = A * (dA - dot[column]) * scale, written over dA.  (The apply kernel's thread layout -- one column chunk, 16 rows --
// measured 287 us against this flat grid-stride loop's 240 us on the 16 x 4096 x 4096 matrices: two input streams.)
template <typename T>
__global__ __launch_bounds__(256) void softmax_cols_bwd_kernel(const T* a, T* g, int ld, long long sb, int rows, int cols,
                                                               float scale, const float* dot, long long chunks) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int cpr = cols / VEC;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < chunks; i += (long long)gridDim.x * 256) {
    const int cc = (int)(i % cpr);
    const long long rr = i / cpr;
    const int r = (int)(rr % rows), b = (int)(rr / rows);
    const long long off = (long long)b * sb + (long long)r * ld + cc * VEC;
    const Vec16<T> va = ld16(a + off);
    Vec16<T> vg = ld16(g + off);
    const float* dp = dot + (long long)b * cols + cc * VEC;
#pragma unroll
    for (int e = 0; e < VEC; ++e) vg.v[e] = (T)(tof(va.v[e]) * (tof(vg.v[e]) - dp[e]) * scale);
    st16(g + off, vg);
  }
}

// ---- softmax over the COLUMN axis: one wave per row, the row stays in registers (cols <= 64 * VEC * RPT) ------------
template <typename T, int RPT, bool BWD>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const T* a, T* s, int ld, long long sb, int rows, int cols, float scale,
                                                           long long total_rows) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int lane = threadIdx.x & 63;
  for (long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); row < total_rows; row += (long long)gridDim.x * 4) {
    const int b = (int)(row / rows), r = (int)(row % rows);
    const long long off = (long long)b * sb + (long long)r * ld;
    Vec16<T> v[RPT], w[RPT];
    float red = BWD ? 0.f : -INFINITY;
#pragma unroll
    for (int u = 0; u < RPT; ++u) {   // (unconditional loads, all in flight: chunk 0 of the row for a lane beyond the columns)
      const int c = (u * 64 + lane) * VEC;
      v[u] = ld16(s + off + (c < cols ? c : 0));
      if constexpr (BWD) w[u] = ld16(a + off + (c < cols ? c : 0));
    }
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
      ta_pin16(v[u]);
      if constexpr (BWD) ta_pin16(w[u]);
    }
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
      const int c = (u * 64 + lane) * VEC;
      if (c < cols) {
        if constexpr (BWD) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) red = fmaf(tof(v[u].v[e]), tof(w[u].v[e]), red);
        } else {
#pragma unroll
          for (int e = 0; e < VEC; ++e) red = fmaxf(red, tof(v[u].v[e]) * scale);
        }
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const float t = __shfl_xor(red, o);
      red = BWD ? red + t : fmaxf(red, t);
    }
    if constexpr (BWD) {   // dS = A * (dA - sum(A dA)) * scale
#pragma unroll
      for (int u = 0; u < RPT; ++u) {
        const int c = (u * 64 + lane) * VEC;
        if (c < cols) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) v[u].v[e] = (T)(tof(w[u].v[e]) * (tof(v[u].v[e]) - red) * scale);
          st16(s + off + c, v[u]);
        }
      }
    } else {
      float ex[RPT][VEC], z = 0.f;
#pragma unroll
      for (int u = 0; u < RPT; ++u) {
        const int c = (u * 64 + lane) * VEC;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          ex[u][e] = c < cols ? uz_exp<T>(tof(v[u].v[e]) * scale - red) : 0.f;
          z += ex[u][e];
        }
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) z += __shfl_xor(z, o);
      const float rz = 1.f / z;
#pragma unroll
      for (int u = 0; u < RPT; ++u) {
        const int c = (u * 64 + lane) * VEC;
        if (c < cols) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) v[u].v[e] = (T)(ex[u][e] * rz);
          st16(s + off + c, v[u]);
        }
      }
    }
  }
}

// ---- F.adaptive_avg_pool2d on NHWC: window of output i = [floor(i * In / Out), ceil((i + 1) * In / Out)) --------------
__device__ __forceinline__ int ap_start(int i, int in, int out) { return (int)(((long long)i * in) / out); }
__device__ __forceinline__ int ap_end(int i, int in, int out) { return (int)(((long long)(i + 1) * in + out - 1) / out); }

template <typename T>
__global__ __launch_bounds__(256) void adaptive_pool_fwd_kernel(const T* x, int ldx, int N, int Hi, int Wi, int C, T* y, int ldy,
                                                                int Ho, int Wo, long long chunks) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int cpr = C / VEC;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < chunks; i += (long long)gridDim.x * 256) {
    const int cc = (int)(i % cpr);
    long long p = i / cpr;
    const int ow = (int)(p % Wo);
    p /= Wo;
    const int oh = (int)(p % Ho), n = (int)(p / Ho);
    const int h0 = ap_start(oh, Hi, Ho), h1 = ap_end(oh, Hi, Ho), w0 = ap_start(ow, Wi, Wo), w1 = ap_end(ow, Wi, Wo);
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    for (int h = h0; h < h1; ++h)
      for (int w = w0; w < w1; ++w) {
        const Vec16<T> v = ld16(x + (((long long)n * Hi + h) * Wi + w) * ldx + cc * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += tof(v.v[e]);
      }
    const float inv = 1.f / (float)((h1 - h0) * (w1 - w0));
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o.v[e] = (T)(acc[e] * inv);
    st16(y + (((long long)n * Ho + oh) * Wo + ow) * ldy + cc * VEC, o);
  }
}

// gradient, gathered per INPUT pixel: every output window that contains (h, w) contributes g / area
template <typename T>
__global__ __launch_bounds__(256) void adaptive_pool_bwd_kernel(const T* g, int ldg, int N, int Hi, int Wi, int C, T* dx, int lddx,
                                                                int Ho, int Wo, int accumulate, long long chunks) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int cpr = C / VEC;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < chunks; i += (long long)gridDim.x * 256) {
    const int cc = (int)(i % cpr);
    long long p = i / cpr;
    const int w = (int)(p % Wi);
    p /= Wi;
    const int h = (int)(p % Hi), n = (int)(p / Hi);
    int oh_lo = (int)(((long long)h * Ho) / Hi) - 1, oh_hi = (int)(((long long)(h + 1) * Ho + Hi - 1) / Hi);
    int ow_lo = (int)(((long long)w * Wo) / Wi) - 1, ow_hi = (int)(((long long)(w + 1) * Wo + Wi - 1) / Wi);
    oh_lo = oh_lo < 0 ? 0 : oh_lo;
    ow_lo = ow_lo < 0 ? 0 : ow_lo;
    oh_hi = oh_hi >= Ho ? Ho - 1 : oh_hi;
    ow_hi = ow_hi >= Wo ? Wo - 1 : ow_hi;
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    for (int oh = oh_lo; oh <= oh_hi; ++oh) {
      const int h0 = ap_start(oh, Hi, Ho), h1 = ap_end(oh, Hi, Ho);
      if (h < h0 || h >= h1) continue;
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        const int w0 = ap_start(ow, Wi, Wo), w1 = ap_end(ow, Wi, Wo);
        if (w < w0 || w >= w1) continue;
        const float inv = 1.f / (float)((h1 - h0) * (w1 - w0));
        const Vec16<T> v = ld16(g + (((long long)n * Ho + oh) * Wo + ow) * ldg + cc * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = fmaf(tof(v.v[e]), inv, acc[e]);
      }
    }
    T* dp = dx + (((long long)n * Hi + h) * Wi + w) * lddx + cc * VEC;
    Vec16<T> o;
    if (accumulate) {
      o = ld16(dp);
#pragma unroll
      for (int e = 0; e < VEC; ++e) o.v[e] = (T)(tof(o.v[e]) + acc[e]);
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) o.v[e] = (T)acc[e];
    }
    st16(dp, o);
  }
}

// ---- out[r] = sum_c a[r][c] * b[r][c] (a fp32, b run dtype): one wave per row ----------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rowdot_kernel(const float* a, int lda, const T* b, int ldb, long long rows, int C, float* out) {
  const int lane = threadIdx.x & 63;
  for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long long)gridDim.x * 4) {
    float t = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
      const f32x4 va = *reinterpret_cast<const f32x4*>(a + r * lda + c);
      const T* bp = b + r * ldb + c;
#pragma unroll
      for (int e = 0; e < 4; ++e) t = fmaf(va[e], tof(bp[e]), t);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) t += __shfl_xor(t, o);
    if (lane == 0) out[r] = t;
  }
}

// ---- dst[r][c] = (T) src[r][c] -----------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void cast_rows_kernel(const float* src, int lds, T* dst, int ldd, long long rows, int C,
                                                        int accumulate, long long chunks) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int cpr = C / VEC;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < chunks; i += (long long)gridDim.x * 256) {
    const int cc = (int)(i % cpr);
    const long long r = i / cpr;
    float sp[VEC];
#pragma unroll
    for (int e = 0; e < VEC; e += 4) *reinterpret_cast<f32x4*>(&sp[e]) = *reinterpret_cast<const f32x4*>(src + r * lds + cc * VEC + e);
    T* dp = dst + r * ldd + cc * VEC;
    Vec16<T> o;
    if (accumulate) {
      o = ld16(dp);
#pragma unroll
      for (int e = 0; e < VEC; ++e) o.v[e] = (T)(tof(o.v[e]) + sp[e]);
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) o.v[e] = (T)sp[e];
    }
    st16(dp, o);
  }
}

// ---- out[p][c] = x[p][c] + map[p % HW][c] (map fp32): position encodings / embeddings broadcast over the batch ---------
template <typename T>
__global__ __launch_bounds__(256) void add_map_kernel(const T* x, int ldx, const float* map, T* out, int ldo, long long P, int HW, int C,
                                                      long long chunks) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int cpr = C / VEC;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < chunks; i += (long long)gridDim.x * 256) {
    const int cc = (int)(i % cpr);
    const long long p = i / cpr;
    const int q = (int)(p % HW);
    Vec16<T> v = ld16(x + p * ldx + cc * VEC);
    const float* mp = map + (long long)q * C + cc * VEC;
#pragma unroll
    for (int e = 0; e < VEC; e += 4) {
      const f32x4 m4 = *reinterpret_cast<const f32x4*>(mp + e);
#pragma unroll
      for (int k = 0; k < 4; ++k) v.v[e + k] = (T)(tof(v.v[e + k]) + m4[k]);
    }
    st16(out + p * ldo + cc * VEC, v);
  }
}

inline unsigned grid_for_chunks(long long chunks) {
  long long g = (chunks + 255) / 256;
  const long long cap = 16LL * UZ_NUM_CU_HW;
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

bool softmax_args_ok(int dtype, const void* s, int ld, long long sb, int batch, int rows, int cols, int axis) {
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  return (dtype == UZ_F32 || dtype == UZ_BF16) && s && batch >= 1 && batch <= 65535 && rows >= 1 && cols >= 1 &&
         cols % vec == 0 && ld % vec == 0 && ld >= cols && sb % vec == 0 && (axis == 0 || axis == 1) &&
         ((uintptr_t)s & 15) == 0;
}

}  // namespace

extern "C" long long uz_softmax_workspace_bytes(int batch, int rows, int cols, int axis) {
  UZ_REQUIRE(batch >= 1 && rows >= 1 && cols >= 1 && (axis == 0 || axis == 1), "uz_softmax_workspace_bytes: bad arguments");
  if (axis == 1) return 0;
  return ((long long)batch * uz_cdiv(rows, SM_SLICE) + batch) * 2 * cols * (long long)sizeof(float);
}

extern "C" int uz_softmax_fwd(int dtype, void* s, int ld, long long sb, int batch, int rows, int cols, int axis, float scale,
                              void* workspace, void* stream) {
  UZ_REQUIRE(softmax_args_ok(dtype, s, ld, sb, batch, rows, cols, axis), "uz_softmax_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  if (axis == 0) {
    UZ_REQUIRE(workspace != nullptr, "uz_softmax_fwd: axis 0 needs uz_softmax_workspace_bytes() of workspace");
    const int nsl = uz_cdiv(rows, SM_SLICE);
    float* part = static_cast<float*>(workspace);
    float* mz = part + (long long)batch * nsl * 2 * cols;
    const dim3 grid(uz_cdiv(cols, 16 * vec), batch, nsl), block(256);
    if (dtype == UZ_BF16) hipLaunchKernelGGL(softmax_cols_stats_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)s, ld, sb, rows, cols, scale, part);
    else hipLaunchKernelGGL(softmax_cols_stats_kernel<float>, grid, block, 0, st, (const float*)s, ld, sb, rows, cols, scale, part);
    UZ_LAUNCH_CHECK("uz_softmax_fwd(stats)");
    const long long total = (long long)batch * cols;
    hipLaunchKernelGGL(softmax_cols_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const float*)part, nsl, cols, total, mz);
    UZ_LAUNCH_CHECK("uz_softmax_fwd(combine)");
    UZ_REQUIRE(uz_cdiv(rows, SM_ROWS) <= 65535, "uz_softmax_fwd: too many rows");
    const dim3 grid2(uz_cdiv(cols / vec, 256), uz_cdiv(rows, SM_ROWS), batch);
    if (dtype == UZ_BF16) hipLaunchKernelGGL(softmax_cols_apply_kernel<bf16_t>, grid2, block, 0, st, (bf16_t*)s, ld, sb, rows, cols, scale, (const float*)mz);
    else hipLaunchKernelGGL(softmax_cols_apply_kernel<float>, grid2, block, 0, st, (float*)s, ld, sb, rows, cols, scale, (const float*)mz);
  } else {
    UZ_REQUIRE(cols <= 64 * vec * 4, "uz_softmax_fwd: rows longer than 64 * 4 sixteen-byte chunks");
    const long long total = (long long)batch * rows;
    const dim3 grid(grid_for_chunks(total * 64)), block(256);
    const int rpt = uz_cdiv(cols, 64 * vec);
#define UZ_ROWS_FWD(T, R) hipLaunchKernelGGL((softmax_rows_kernel<T, R, false>), grid, block, 0, st, (const T*)nullptr, (T*)s, ld, sb, rows, cols, scale, total)
    if (dtype == UZ_BF16) {
      if (rpt == 1) UZ_ROWS_FWD(bf16_t, 1); else if (rpt == 2) UZ_ROWS_FWD(bf16_t, 2); else UZ_ROWS_FWD(bf16_t, 4);
    } else {
      if (rpt == 1) UZ_ROWS_FWD(float, 1); else if (rpt == 2) UZ_ROWS_FWD(float, 2); else UZ_ROWS_FWD(float, 4);
    }
#undef UZ_ROWS_FWD
  }
  UZ_LAUNCH_CHECK("uz_softmax_fwd");
  return UZ_OK;
}

extern "C" int uz_softmax_bwd(int dtype, const void* a, void* g, int ld, long long sb, int batch, int rows, int cols, int axis,
                              float scale, float* dot, int dot_given, void* stream) {
  UZ_REQUIRE(softmax_args_ok(dtype, g, ld, sb, batch, rows, cols, axis) && a && ((uintptr_t)a & 15) == 0,
             "uz_softmax_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  if (axis == 0) {
    UZ_REQUIRE(dot != nullptr, "uz_softmax_bwd: axis 0 needs the (batch, cols) fp32 buffer `dot`");
    if (!dot_given) {
      const dim3 grid(uz_cdiv(cols, 16 * vec), batch), block(256);
      if (dtype == UZ_BF16) hipLaunchKernelGGL(softmax_cols_dot_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)a, (const bf16_t*)g, ld, sb, rows, cols, dot);
      else hipLaunchKernelGGL(softmax_cols_dot_kernel<float>, grid, block, 0, st, (const float*)a, (const float*)g, ld, sb, rows, cols, dot);
      UZ_LAUNCH_CHECK("uz_softmax_bwd(dot)");
    }
    const long long chunks = (long long)batch * rows * (cols / vec);
    const dim3 grid(grid_for_chunks(chunks)), block(256);
    if (dtype == UZ_BF16) hipLaunchKernelGGL(softmax_cols_bwd_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)a, (bf16_t*)g, ld, sb, rows, cols, scale, (const float*)dot, chunks);
    else hipLaunchKernelGGL(softmax_cols_bwd_kernel<float>, grid, block, 0, st, (const float*)a, (float*)g, ld, sb, rows, cols, scale, (const float*)dot, chunks);
  } else {
    UZ_REQUIRE(cols <= 64 * vec * 4, "uz_softmax_bwd: rows longer than 64 * 4 sixteen-byte chunks");
    const long long total = (long long)batch * rows;
    const dim3 grid(grid_for_chunks(total * 64)), block(256);
    const int rpt = uz_cdiv(cols, 64 * vec);
#define UZ_ROWS_BWD(T, R) hipLaunchKernelGGL((softmax_rows_kernel<T, R, true>), grid, block, 0, st, (const T*)a, (T*)g, ld, sb, rows, cols, scale, total)
    if (dtype == UZ_BF16) {
      if (rpt == 1) UZ_ROWS_BWD(bf16_t, 1); else if (rpt == 2) UZ_ROWS_BWD(bf16_t, 2); else UZ_ROWS_BWD(bf16_t, 4);
    } else {
      if (rpt == 1) UZ_ROWS_BWD(float, 1); else if (rpt == 2) UZ_ROWS_BWD(float, 2); else UZ_ROWS_BWD(float, 4);
    }
#undef UZ_ROWS_BWD
  }
  UZ_LAUNCH_CHECK("uz_softmax_bwd");
  return UZ_OK;
}

extern "C" int uz_adaptive_avgpool_fwd(int dtype, const void* x, int ldx, int N, int Hi, int Wi, int C, void* y, int ldy, int Ho,
                                       int Wo, void* stream) {
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE((dtype == UZ_F32 || dtype == UZ_BF16) && x && y && N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0 &&
                 C % vec == 0 && ldx % vec == 0 && ldy % vec == 0 && ldx >= C && ldy >= C,
             "uz_adaptive_avgpool_fwd: bad arguments");
  const long long chunks = (long long)N * Ho * Wo * (C / vec);
  const dim3 grid(grid_for_chunks(chunks)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == UZ_BF16) hipLaunchKernelGGL(adaptive_pool_fwd_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, ldx, N, Hi, Wi, C, (bf16_t*)y, ldy, Ho, Wo, chunks);
  else hipLaunchKernelGGL(adaptive_pool_fwd_kernel<float>, grid, block, 0, st, (const float*)x, ldx, N, Hi, Wi, C, (float*)y, ldy, Ho, Wo, chunks);
  UZ_LAUNCH_CHECK("uz_adaptive_avgpool_fwd");
  return UZ_OK;
}

extern "C" int uz_adaptive_avgpool_bwd(int dtype, const void* g, int ldg, int N, int Hi, int Wi, int C, void* dx, int lddx, int Ho,
                                       int Wo, int accumulate, void* stream) {
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE((dtype == UZ_F32 || dtype == UZ_BF16) && g && dx && N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0 &&
                 C % vec == 0 && ldg % vec == 0 && lddx % vec == 0 && ldg >= C && lddx >= C,
             "uz_adaptive_avgpool_bwd: bad arguments");
  const long long chunks = (long long)N * Hi * Wi * (C / vec);
  const dim3 grid(grid_for_chunks(chunks)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == UZ_BF16) hipLaunchKernelGGL(adaptive_pool_bwd_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)g, ldg, N, Hi, Wi, C, (bf16_t*)dx, lddx, Ho, Wo, accumulate, chunks);
  else hipLaunchKernelGGL(adaptive_pool_bwd_kernel<float>, grid, block, 0, st, (const float*)g, ldg, N, Hi, Wi, C, (float*)dx, lddx, Ho, Wo, accumulate, chunks);
  UZ_LAUNCH_CHECK("uz_adaptive_avgpool_bwd");
  return UZ_OK;
}

extern "C" int uz_rowdot_f32(int dtype, const float* a, int lda, const void* b, int ldb, long long rows, int C, float* out,
                             void* stream) {
  UZ_REQUIRE((dtype == UZ_F32 || dtype == UZ_BF16) && a && b && out && rows > 0 && C > 0 && C % 4 == 0 && lda % 4 == 0 &&
                 lda >= C && ldb >= C && ((uintptr_t)a & 15) == 0,
             "uz_rowdot_f32: bad arguments");
  const dim3 grid(grid_for_chunks(rows * 64)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == UZ_BF16) hipLaunchKernelGGL(rowdot_kernel<bf16_t>, grid, block, 0, st, a, lda, (const bf16_t*)b, ldb, rows, C, out);
  else hipLaunchKernelGGL(rowdot_kernel<float>, grid, block, 0, st, a, lda, (const float*)b, ldb, rows, C, out);
  UZ_LAUNCH_CHECK("uz_rowdot_f32");
  return UZ_OK;
}

extern "C" int uz_cast_rows(int dtype, const float* src, int lds, void* dst, int ldd, long long rows, int C, int accumulate,
                            void* stream) {
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE((dtype == UZ_F32 || dtype == UZ_BF16) && src && dst && rows > 0 && C > 0 && C % vec == 0 && ldd % vec == 0 &&
                 lds % 4 == 0 && lds >= C && ldd >= C && (((uintptr_t)dst | (uintptr_t)src) & 15) == 0,
             "uz_cast_rows: bad arguments");
  const long long chunks = rows * (C / vec);
  const dim3 grid(grid_for_chunks(chunks)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == UZ_BF16) hipLaunchKernelGGL(cast_rows_kernel<bf16_t>, grid, block, 0, st, src, lds, (bf16_t*)dst, ldd, rows, C, accumulate, chunks);
  else hipLaunchKernelGGL(cast_rows_kernel<float>, grid, block, 0, st, src, lds, (float*)dst, ldd, rows, C, accumulate, chunks);
  UZ_LAUNCH_CHECK("uz_cast_rows");
  return UZ_OK;
}

extern "C" int uz_add_map(int dtype, const void* x, int ldx, const float* map, void* out, int ldo, long long P, int HW, int C,
                          void* stream) {
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE((dtype == UZ_F32 || dtype == UZ_BF16) && x && map && out && P > 0 && HW > 0 && P % HW == 0 && C > 0 && C % vec == 0 &&
                 ldx % vec == 0 && ldo % vec == 0 && ldx >= C && ldo >= C && ((uintptr_t)map & 15) == 0,
             "uz_add_map: bad arguments");
  const long long chunks = P * (C / vec);
  const dim3 grid(grid_for_chunks(chunks)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == UZ_BF16) hipLaunchKernelGGL(add_map_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, ldx, map, (bf16_t*)out, ldo, P, HW, C, chunks);
  else hipLaunchKernelGGL(add_map_kernel<float>, grid, block, 0, st, (const float*)x, ldx, map, (float*)out, ldo, P, HW, C, chunks);
  UZ_LAUNCH_CHECK("uz_add_map");
  return UZ_OK;
}

// ---- channel-wise cross attention of UCTransNet: InstanceNorm2d + softmax on the score planes -----------------------
// Attention_org.forward (unet_zoo/models/uctransnet.py:160-216): per (image, head) the (C, KV) plane of
// scores = Q^T K / sqrt(KV) is instance-normalised (nn.InstanceNorm2d(heads): mean and biased variance over the
// plane, no affine), softmax over KV, and the context layers of the heads are averaged.  One workgroup per (image,
// head); a plane is at most 128 x 240 fp32 = 123 KB and stays in L2 between the passes, a row lives in registers.
//   forward : pcat[b][c][h * KV + kv] = softmax_kv((s - mean) * rstd) / H (run dtype: the operand of the context product
//             over K = H * KV, which also takes the mean over heads) and its transpose pcat_t[b][h * KV + kv][c]
//   backward: from d(loss)/d(pcat) (fp32, (B, C, H * KV)) to d(loss)/d(scores) * scale as ds[b][h][c][kv] and its
//             transpose ds_t[b][h][kv][c] (run dtype: operands of dQ = K dS^T and dK = Q dS)
namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {   // 1024 threads; red: 16 floats
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) t += red[k];
  return t;
}

template <typename T, int RPT, bool BWD>
__global__ __launch_bounds__(1024) void chanattn_probs_kernel(const float* __restrict__ scores, const float* __restrict__ dpc,
                                                             int H, int C, int KV, float scale, float eps, T* __restrict__ o0,
                                                             T* __restrict__ o1) {
  __shared__ float red[16];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* pl = scores + ((long long)b * H + h) * C * KV;
  const int n = C * KV;
  // the plane passes: eight UNCONDITIONAL loads of a thread in flight (element 0 past the end, dropped by the select; values made
  // opaque so that the select cannot pull the loads back under a branch), added in the same order as one load per trip -- these
  // loops were 30 dependent L2 round trips per pass, and a load under `if (in range)` is waited for where the block ends
  float t = 0.f;
  for (int i0 = threadIdx.x; i0 < n; i0 += 8 * 1024) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = pl[i0 + k * 1024 < n ? i0 + k * 1024 : 0];
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(v[k]));
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (i0 + k * 1024 < n) t += v[k];
  }
  const float mean = block_sum(t, red) * scale / (float)n;
  t = 0.f;
  for (int i0 = threadIdx.x; i0 < n; i0 += 8 * 1024) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = pl[i0 + k * 1024 < n ? i0 + k * 1024 : 0];
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(v[k]));
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (i0 + k * 1024 < n) {
        const float d = v[k] * scale - mean;
        t += d * d;
      }
  }
  const float rstd = rsqrtf(block_sum(t, red) / (float)n + eps);
  const float invH = 1.f / (float)H;
  const int HK = H * KV;

  // one row: normalised scores and probabilities of this lane's RPT elements
  auto row = [&](int c, float (&sh)[RPT], float (&p)[RPT]) {
    float mx = -INFINITY;
    if constexpr (RPT <= 4 || !BWD) {
#pragma unroll
      for (int u = 0; u < RPT; ++u) sh[u] = pl[(long long)c * KV + (u * 64 + lane < KV ? u * 64 + lane : 0)];   // (unconditional, in flight together)
#pragma unroll
      for (int u = 0; u < RPT; ++u) asm volatile("" : "+v"(sh[u]));
#pragma unroll
      for (int u = 0; u < RPT; ++u) {
        const int kv = u * 64 + lane;
        sh[u] = kv < KV ? (sh[u] * scale - mean) * rstd : -INFINITY;
        mx = fmaxf(mx, sh[u]);
      }
    } else {   // (the backward at 16 elements per lane and 1024 threads: the batched form spills)
#pragma unroll
      for (int u = 0; u < RPT; ++u) {
        const int kv = u * 64 + lane;
        sh[u] = kv < KV ? (pl[(long long)c * KV + kv] * scale - mean) * rstd : -INFINITY;
        mx = fmaxf(mx, sh[u]);
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float z = 0.f;
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
      p[u] = u * 64 + lane < KV ? expf(sh[u] - mx) : 0.f;
      z += p[u];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) z += __shfl_xor(z, o);
    const float rz = 1.f / z;
#pragma unroll
    for (int u = 0; u < RPT; ++u) p[u] *= rz;
  };

  if constexpr (!BWD) {
    T* pcat = o0 + (long long)b * C * HK + h * KV;
    T* pcat_t = o1 + ((long long)b * HK + h * KV) * C;
    for (int c = wave; c < C; c += 16) {
      float sh[RPT], p[RPT];
      row(c, sh, p);
#pragma unroll
      for (int u = 0; u < RPT; ++u) {
        const int kv = u * 64 + lane;
        if (kv < KV) {
          const T v = (T)(p[u] * invH);
          pcat[(long long)c * HK + kv] = v;
          pcat_t[(long long)kv * C + c] = v;
        }
      }
    }
  } else {
    // d(loss)/d(P) = dpc / H; softmax: dsh = P * (dP - sum(P dP)); instance norm: ds = rstd * (dsh - mean(dsh) -
    // sh * mean(dsh * sh)) over the plane
    const float* dp = dpc + (long long)b * C * HK + h * KV;
    float a1 = 0.f, a2 = 0.f;
    for (int c = wave; c < C; c += 16) {
      float sh[RPT], p[RPT], g[RPT], dot = 0.f;
      row(c, sh, p);
      if constexpr (RPT <= 4) {
#pragma unroll
        for (int u = 0; u < RPT; ++u) g[u] = dp[(long long)c * HK + (u * 64 + lane < KV ? u * 64 + lane : 0)];   // (unconditional, in flight together)
#pragma unroll
        for (int u = 0; u < RPT; ++u) asm volatile("" : "+v"(g[u]));
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
          const int kv = u * 64 + lane;
          g[u] = kv < KV ? g[u] * invH : 0.f;
          dot += p[u] * g[u];
        }
      } else {   // (16 elements per lane at 1024 threads: the batched form spills)
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
          const int kv = u * 64 + lane;
          g[u] = kv < KV ? dp[(long long)c * HK + kv] * invH : 0.f;
          dot += p[u] * g[u];
        }
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) dot += __shfl_xor(dot, o);
#pragma unroll
      for (int u = 0; u < RPT; ++u) {
        if (u * 64 + lane < KV) {
          const float dsh = p[u] * (g[u] - dot);
          a1 += dsh;
          a2 += dsh * sh[u];
        }
      }
    }
    const float m1 = block_sum(a1, red) / (float)n;
    const float m2 = block_sum(a2, red) / (float)n;
    T* ds = o0 + ((long long)b * H + h) * C * KV;
    T* ds_t = o1 + ((long long)b * H + h) * KV * C;
    for (int c = wave; c < C; c += 16) {
      float sh[RPT], p[RPT], g[RPT], dot = 0.f;
      row(c, sh, p);
      if constexpr (RPT <= 4) {
#pragma unroll
        for (int u = 0; u < RPT; ++u) g[u] = dp[(long long)c * HK + (u * 64 + lane < KV ? u * 64 + lane : 0)];   // (unconditional, in flight together)
#pragma unroll
        for (int u = 0; u < RPT; ++u) asm volatile("" : "+v"(g[u]));
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
          const int kv = u * 64 + lane;
          g[u] = kv < KV ? g[u] * invH : 0.f;
          dot += p[u] * g[u];
        }
      } else {   // (16 elements per lane at 1024 threads: the batched form spills)
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
          const int kv = u * 64 + lane;
          g[u] = kv < KV ? dp[(long long)c * HK + kv] * invH : 0.f;
          dot += p[u] * g[u];
        }
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) dot += __shfl_xor(dot, o);
#pragma unroll
      for (int u = 0; u < RPT; ++u) {
        const int kv = u * 64 + lane;
        if (kv < KV) {
          const float dsh = p[u] * (g[u] - dot);
          const T v = (T)(rstd * (dsh - m1 - sh[u] * m2) * scale);
          ds[(long long)c * KV + kv] = v;
          ds_t[(long long)kv * C + c] = v;
        }
      }
    }
  }
}

template <bool BWD>
int chanattn_launch(int dtype, const float* scores, const float* dpc, int B, int H, int C, int KV, float scale, float eps,
                    void* o0, void* o1, hipStream_t st) {
  const dim3 grid(B * H), block(1024);
  const int rpt = uz_cdiv(KV, 64);
#define UZ_CA(T, R) hipLaunchKernelGGL((chanattn_probs_kernel<T, R, BWD>), grid, block, 0, st, scores, dpc, H, C, KV, scale, eps, (T*)o0, (T*)o1)
  if (dtype == UZ_BF16) {
    if (rpt <= 4) UZ_CA(bf16_t, 4); else UZ_CA(bf16_t, 16);
  } else {
    if (rpt <= 4) UZ_CA(float, 4); else UZ_CA(float, 16);
  }
#undef UZ_CA
  UZ_LAUNCH_CHECK("uz_chanattn_probs");
  return UZ_OK;
}

}  // namespace

extern "C" int uz_chanattn_probs_fwd(int dtype, const float* scores, int B, int H, int C, int KV, float scale, float eps,
                                     void* pcat, void* pcat_t, void* stream) {
  UZ_REQUIRE((dtype == UZ_F32 || dtype == UZ_BF16) && scores && pcat && pcat_t && B >= 1 && H >= 1 && C >= 1 && KV >= 1 &&
                 KV <= 1024 && (long long)B * H < (1LL << 31) && (long long)C * KV < (1LL << 31),
             "uz_chanattn_probs_fwd: bad arguments");
  return chanattn_launch<false>(dtype, scores, nullptr, B, H, C, KV, scale, eps, pcat, pcat_t, (hipStream_t)stream);
}

extern "C" int uz_chanattn_probs_bwd(int dtype, const float* scores, const float* dpc, int B, int H, int C, int KV, float scale,
                                     float eps, void* ds, void* ds_t, void* stream) {
  UZ_REQUIRE((dtype == UZ_F32 || dtype == UZ_BF16) && scores && dpc && ds && ds_t && B >= 1 && H >= 1 && C >= 1 && KV >= 1 &&
                 KV <= 1024 && (long long)B * H < (1LL << 31) && (long long)C * KV < (1LL << 31),
             "uz_chanattn_probs_bwd: bad arguments");
  return chanattn_launch<true>(dtype, scores, dpc, B, H, C, KV, scale, eps, ds, ds_t, (hipStream_t)stream);
}

// ---- nn.Dropout in training mode on a token / pixel map: out = x * [u >= p] / (1 - p), u = the caller's uniform draw ----
// (the Bernoulli draw stays torch's generator -- `torch.rand` -- so that a seed reproduces a run; mask, scale and the
// copy were four torch passes per dropout, 48 dropouts per UCTransNet step.)  The same kernel applies the mask to the
// gradient.
namespace {
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* x, int ldx, const float* u, float p, float scale, T* out, int ldo,
                                                      long long P, int C, long long chunks) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int cpr = C / VEC;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < chunks; i += (long long)gridDim.x * 256) {
    const int cc = (int)(i % cpr);
    const long long r = i / cpr;
    Vec16<T> v = ld16(x + r * ldx + cc * VEC);
    const float* up = u + r * C + cc * VEC;
#pragma unroll
    for (int e = 0; e < VEC; e += 4) {
      const f32x4 u4 = *reinterpret_cast<const f32x4*>(up + e);
#pragma unroll
      for (int k = 0; k < 4; ++k) v.v[e + k] = (T)(u4[k] >= p ? tof(v.v[e + k]) * scale : 0.f);
    }
    st16(out + r * ldo + cc * VEC, v);
  }
}
}  // namespace

extern "C" int uz_dropout(int dtype, const void* x, int ldx, const float* u, float p, void* out, int ldo, long long P, int C,
                          void* stream) {
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE((dtype == UZ_F32 || dtype == UZ_BF16) && x && u && out && P > 0 && C > 0 && C % vec == 0 && ldx % vec == 0 &&
                 ldo % vec == 0 && ldx >= C && ldo >= C && p >= 0.f && p < 1.f && ((uintptr_t)u & 15) == 0,
             "uz_dropout: bad arguments");
  const long long chunks = P * (C / vec);
  const dim3 grid(grid_for_chunks(chunks)), block(256);
  hipStream_t st = (hipStream_t)stream;
  const float scale = 1.f / (1.f - p);
  if (dtype == UZ_BF16) hipLaunchKernelGGL(dropout_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, ldx, u, p, scale, (bf16_t*)out, ldo, P, C, chunks);
  else hipLaunchKernelGGL(dropout_kernel<float>, grid, block, 0, st, (const float*)x, ldx, u, p, scale, (float*)out, ldo, P, C, chunks);
  UZ_LAUNCH_CHECK("uz_dropout");
  return UZ_OK;
}

// ---- CCA gate of UCTransNet (uctransnet.py:417-427): out = relu(x * s[n][c]), s = sigmoid(...) > 0 per (image, channel) --
// forward: out = relu(x * s).  backward, with m = g * [x > 0] (s > 0, so the ReLU mask is the sign of x):
//   mode 0: out = m * x          (its per-image column sums are d(loss)/d(s))
//   mode 1: out = m * s + a[n][c] (a = the gradient that reaches x through the global average behind s)
namespace {
template <typename T, int MODE>   // MODE 2: forward
__global__ __launch_bounds__(256) void chanscale_kernel(const T* g, int ldg, const T* x, int ldx, const float* s, const float* a,
                                                        T* out, int ldo, int HW, int C, long long chunks) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int cpr = C / VEC;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < chunks; i += (long long)gridDim.x * 256) {
    const int cc = (int)(i % cpr);
    const long long p = i / cpr;
    const long long n = p / HW;
    const Vec16<T> xv = ld16(x + p * ldx + cc * VEC);
    Vec16<T> o;
    if constexpr (MODE == 2) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) o.v[e] = (T)fmaxf(tof(xv.v[e]) * s[n * C + cc * VEC + e], 0.f);
    } else {
      const Vec16<T> gv = ld16(g + p * ldg + cc * VEC);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float xf = tof(xv.v[e]);
        const float m = xf > 0.f ? tof(gv.v[e]) : 0.f;
        o.v[e] = (T)(MODE == 0 ? m * xf : m * s[n * C + cc * VEC + e] + a[n * C + cc * VEC + e]);
      }
    }
    st16(out + p * ldo + cc * VEC, o);
  }
}
}  // namespace

extern "C" int uz_chanscale_relu(int dtype, int mode, const void* g, int ldg, const void* x, int ldx, const float* s, const float* a,
                                 int N, int HW, int C, void* out, int ldo, void* stream) {
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE((dtype == UZ_F32 || dtype == UZ_BF16) && mode >= 0 && mode <= 2 && x && out && N > 0 && HW > 0 && C > 0 && C % vec == 0 &&
                 ldx % vec == 0 && ldo % vec == 0 && ldx >= C && ldo >= C && (mode == 0 || s) && (mode != 1 || a) &&
                 (mode == 2 || (g && ldg % vec == 0 && ldg >= C)),
             "uz_chanscale_relu: bad arguments");
  const long long chunks = (long long)N * HW * (C / vec);
  const dim3 grid(grid_for_chunks(chunks)), block(256);
  hipStream_t st = (hipStream_t)stream;
#define UZ_CS(T, M) hipLaunchKernelGGL((chanscale_kernel<T, M>), grid, block, 0, st, (const T*)g, ldg, (const T*)x, ldx, s, a, (T*)out, ldo, HW, C, chunks)
  if (dtype == UZ_BF16) {
    if (mode == 0) UZ_CS(bf16_t, 0); else if (mode == 1) UZ_CS(bf16_t, 1); else UZ_CS(bf16_t, 2);
  } else {
    if (mode == 0) UZ_CS(float, 0); else if (mode == 1) UZ_CS(float, 1); else UZ_CS(float, 2);
  }
#undef UZ_CS
  UZ_LAUNCH_CHECK("uz_chanscale_relu");
  return UZ_OK;
}
