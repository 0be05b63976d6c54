// Additive attention gate of Attention-UNet for gfx950 (MI355X), forward and backward, NHWC.
//
// Reference: AttentionBlock.forward, unet_zoo/models/attention_unet.py:34-40
//     g1 = BN(W_g g);  x1 = BN(W_x x);  psi = sigmoid(BN(W_psi relu(g1 + x1)));  return psi * x
// (train-mode BatchNorm: three grid-wide reductions, so the gate is a short chain of
// bandwidth-bound passes).  The two 1x1 convolutions W_g, W_x run on the matrix cores
// (uz_conv_igemm, which also delivers their BatchNorm partial sums); this file holds the rest:
//   forward : uz_attn_psi_fwd  q[p] = b_psi + sum_c relu(G1 + X1)[p,c] * w_psi[c]   (+ partial sums of q)
//             uz_attn_gate_fwd out[p,c] = x[p,c] * sigmoid(sq*q[p] + hq)
//   backward: uz_attn_bwd_psi  dxd = dOut*psi, dZ[p] = (sum_c dOut*x) * psi(1-psi)  (+ partials of dZ, dZ*qhat)
//             uz_attn_bwd_reduce per-channel sums of dPre = dq*w_psi*[G1+X1>0] (and * ghat, * xhat, dw_psi)
//             uz_attn_bwd_apply  dg1raw, dx1raw from dPre and the reduced sums (BN backward)
// Nothing but the 1-channel q / dZ vectors is materialised: relu(G1+X1) and dPre are recomputed
// from the raw 1x1 conv outputs in every pass.  LPP = min(64, channels/VEC) lanes cooperate on one
// pixel (16-byte loads, xor-shuffle reductions); per-workgroup partial rows + fixed-order finalize.
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
__device__ __forceinline__ float sigmoidf_(float z) { return 1.0f / (1.0f + __expf(-z)); }

// block-wide sum of NV per-thread values -> row[0..NV) of the workgroup's partial row
template <int NV> __device__ __forceinline__ void block_sum_to_row(const float* v, float* red, float* row) {
  float w[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    w[i] = v[i];
    for (int o = 32; o > 0; o >>= 1) w[i] += __shfl_xor(w[i], o);
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int i = 0; i < NV; ++i) red[(threadIdx.x >> 6) * NV + i] = w[i];
  __syncthreads();
  if (threadIdx.x < NV) {
    float t = 0.f;
    for (int r = 0; r < (int)(blockDim.x >> 6); ++r) t += red[r * NV + threadIdx.x];
    row[threadIdx.x] = t;
  }
}

struct GateVecs {            // per-channel BatchNorm vectors [4][F]: scale, shift, mean, invstd
  const float* g;
  const float* x;
  const float* q;            // [4][1] for the psi BatchNorm
};

// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_psi_fwd_kernel(const T* __restrict__ g1, int ldg,
                                                           const T* __restrict__ x1, int ldx, GateVecs v,
                                                           const float* __restrict__ wpsi,
                                                           const float* __restrict__ bpsi_p, int P,
                                                           int F, float* __restrict__ q,
                                                           float* __restrict__ partial) {
  constexpr int VEC = ElemTraits<T>::VEC;
  __shared__ float red[256 * 2];
  const float bpsi = bpsi_p != nullptr ? bpsi_p[0] : 0.f;
  const int nchunks = F / VEC;
  const int LPP = nchunks < 64 ? nchunks : 64;   // power of two (host-checked)
  const int ppb = 256 / LPP, sub = threadIdx.x % LPP, pl = threadIdx.x / LPP;
  float acc2[2] = {0.f, 0.f};
  for (int p0 = blockIdx.x * ppb; p0 < P; p0 += gridDim.x * ppb) {
    const int p = p0 + pl;
    float dot = 0.f;
    if (p < P) {
      for (int ch = sub; ch < nchunks; ch += LPP) {
        const int c0 = ch * VEC;
        float a[VEC], b[VEC];
        load_f(g1 + (size_t)p * ldg + c0, a);
        load_f(x1 + (size_t)p * ldx + c0, b);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const float s = fmaf(a[i], v.g[c0 + i], v.g[F + c0 + i]) + fmaf(b[i], v.x[c0 + i], v.x[F + c0 + i]);
          dot = fmaf(fmaxf(s, 0.f), wpsi[c0 + i], dot);
        }
      }
    }
    for (int o = LPP >> 1; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
    if (sub == 0 && p < P) {
      const float qq = dot + bpsi;
      q[p] = qq;
      acc2[0] += qq;
      acc2[1] += qq * qq;
    }
  }
  block_sum_to_row<2>(acc2, red, partial + (size_t)blockIdx.x * 2);
}

template <typename T>
__global__ __launch_bounds__(256) void attn_gate_fwd_kernel(const T* __restrict__ x, int ldx,
                                                            const float* __restrict__ q, const float* __restrict__ vq,
                                                            long long total, int C, T* __restrict__ out, int ldo) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int CC = C / VEC;
  const float sq = vq[0], hq = vq[1];
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(idx % CC);
    const long long p = idx / CC;
    const float psi = sigmoidf_(fmaf(q[p], sq, hq));
    float a[VEC];
    load_f(x + (size_t)p * ldx + cc * VEC, a);
#pragma unroll
    for (int i = 0; i < VEC; ++i) a[i] *= psi;
    store_f(out + (size_t)p * ldo + cc * VEC, a);
  }
}

// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_psi_kernel(const T* __restrict__ dout, int ldd,
                                                           const T* __restrict__ x, int ldx,
                                                           const float* __restrict__ q, const float* __restrict__ vq,
                                                           int P, int C, T* __restrict__ dxd, int lddx,
                                                           float* __restrict__ dz, float* __restrict__ partial) {
  constexpr int VEC = ElemTraits<T>::VEC;
  __shared__ float red[256 * 2];
  const int nchunks = C / VEC;
  const int LPP = nchunks < 64 ? nchunks : 64;
  const int ppb = 256 / LPP, sub = threadIdx.x % LPP, pl = threadIdx.x / LPP;
  const float sq = vq[0], hq = vq[1], mq = vq[2], iq = vq[3];
  float acc2[2] = {0.f, 0.f};
  for (int p0 = blockIdx.x * ppb; p0 < P; p0 += gridDim.x * ppb) {
    const int p = p0 + pl;
    float dot = 0.f, psi = 0.f, qq = 0.f;
    if (p < P) {
      qq = q[p];
      psi = sigmoidf_(fmaf(qq, sq, hq));
      for (int ch = sub; ch < nchunks; ch += LPP) {
        const int c0 = ch * VEC;
        float d[VEC], xv[VEC];
        load_f(dout + (size_t)p * ldd + c0, d);
        load_f(x + (size_t)p * ldx + c0, xv);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          dot = fmaf(d[i], xv[i], dot);
          d[i] *= psi;
        }
        store_f(dxd + (size_t)p * lddx + c0, d);
      }
    }
    for (int o = LPP >> 1; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
    if (sub == 0 && p < P) {
      const float dzz = dot * psi * (1.f - psi);
      dz[p] = dzz;
      acc2[0] += dzz;
      acc2[1] += dzz * (qq - mq) * iq;
    }
  }
  block_sum_to_row<2>(acc2, red, partial + (size_t)blockIdx.x * 2);
}

struct AttnBwdArgs {
  const void* g1;
  const void* x1;
  const float* q;
  const float* dz;
  const float* wpsi;
  GateVecs v;
  const double* a01;   // [2]: sum dZ, sum dZ*qhat (pass reduce/apply)
  const double* tot;   // apply: [4F+1] totals of the reduce pass
  float* partial;      // reduce: [grid][4F+1]
  void* dg1;
  void* dx1;
  double inv_count;
  int P, F, ldg, ldx, lddg, lddx;
};

// PASS 1: per-channel sums (B0 = sum dPre, B1 = sum dPre*ghat, D1 = sum dPre*xhat, W = sum dq*S) and
//         sum dq;  PASS 2: dg1raw = sg (dPre - B0/P - ghat B1/P), dx1raw = sx (dPre - B0/P - xhat D1/P)
template <typename T, int PASS>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const AttnBwdArgs a) {
  constexpr int VEC = ElemTraits<T>::VEC;
  extern __shared__ __attribute__((aligned(16))) float red[];  // PASS 1: [256][4*VEC+1]
  const T* __restrict__ g1 = static_cast<const T*>(a.g1);
  const T* __restrict__ x1 = static_cast<const T*>(a.x1);
  T* __restrict__ dg1 = static_cast<T*>(a.dg1);
  T* __restrict__ dx1 = static_cast<T*>(a.dx1);
  const int F = a.F, nchunks = F / VEC;
  // one lane owns one channel chunk for the whole kernel: blockDim.x = (LPP lanes) x (ppb pixels),
  // blockIdx.y walks chunk groups when F/VEC > 64
  const int LPP = nchunks < 64 ? nchunks : 64;
  const int ppb = 256 / LPP, sub = threadIdx.x % LPP, pl = threadIdx.x / LPP;
  const int ch = blockIdx.y * LPP + sub;
  const int c0 = ch * VEC;
  const float sq = a.v.q[0], mq = a.v.q[2], iq = a.v.q[3];
  const float k0 = (float)(a.a01[0] * a.inv_count), k1 = (float)(a.a01[1] * a.inv_count);
  float sg[VEC], hg[VEC], mg[VEC], ig[VEC], sx[VEC], hx[VEC], mx[VEC], ix[VEC], wp[VEC];
  float tb0[VEC], tb1[VEC], td1[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    sg[i] = a.v.g[c0 + i]; hg[i] = a.v.g[F + c0 + i]; mg[i] = a.v.g[2 * F + c0 + i]; ig[i] = a.v.g[3 * F + c0 + i];
    sx[i] = a.v.x[c0 + i]; hx[i] = a.v.x[F + c0 + i]; mx[i] = a.v.x[2 * F + c0 + i]; ix[i] = a.v.x[3 * F + c0 + i];
    wp[i] = a.wpsi[c0 + i];
    if (PASS == 2) {
      tb0[i] = (float)(a.tot[c0 + i] * a.inv_count);
      tb1[i] = (float)(a.tot[F + c0 + i] * a.inv_count);
      td1[i] = (float)(a.tot[2 * F + c0 + i] * a.inv_count);
    }
  }
  float acc[4 * VEC + 1];
#pragma unroll
  for (int i = 0; i < 4 * VEC + 1; ++i) acc[i] = 0.f;

  for (int p0 = blockIdx.x * ppb; p0 < a.P; p0 += gridDim.x * ppb) {
    const int p = p0 + pl;
    if (p >= a.P) continue;
    const float qq = a.q[p];
    const float dq = sq * (a.dz[p] - k0 - (qq - mq) * iq * k1);
    float gv[VEC], xv[VEC], og[VEC], ox[VEC];
    load_f(g1 + (size_t)p * a.ldg + c0, gv);
    load_f(x1 + (size_t)p * a.ldx + c0, xv);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float s = fmaf(gv[i], sg[i], hg[i]) + fmaf(xv[i], sx[i], hx[i]);
      const float dpre = s > 0.f ? dq * wp[i] : 0.f;
      const float gh = (gv[i] - mg[i]) * ig[i], xh = (xv[i] - mx[i]) * ix[i];
      if (PASS == 1) {
        acc[i] += dpre;
        acc[VEC + i] += dpre * gh;
        acc[2 * VEC + i] += dpre * xh;
        acc[3 * VEC + i] += dq * fmaxf(s, 0.f);
      } else {
        og[i] = sg[i] * (dpre - tb0[i] - gh * tb1[i]);
        ox[i] = sx[i] * (dpre - tb0[i] - xh * td1[i]);
      }
    }
    if (PASS == 1) {
      if (sub == 0 && blockIdx.y == 0) acc[4 * VEC] += dq;
    } else {
      store_f(dg1 + (size_t)p * a.lddg + c0, og);
      store_f(dx1 + (size_t)p * a.lddx + c0, ox);
    }
  }
  if (PASS == 1) {
    constexpr int NV = 4 * VEC + 1;
#pragma unroll
    for (int i = 0; i < NV; ++i) red[threadIdx.x * NV + i] = acc[i];
    __syncthreads();
    // thread (sub, pl) finalises elements pl, pl+ppb, ... of chunk `sub`
    float* row = a.partial + (size_t)blockIdx.x * (4 * F + 1);
    for (int e = pl; e < NV; e += ppb) {
      float t = 0.f;
      for (int r = 0; r < ppb; ++r) t += red[(r * LPP + sub) * NV + e];
      if (e < 4 * VEC) row[(e / VEC) * F + c0 + (e % VEC)] = t;
      else if (sub == 0 && blockIdx.y == 0) row[4 * F] = t;
    }
  }
}

// out[e] (double) = sum over rows of partial[row][e]; optional fp32 copies f0 = out[off0..], ...
__global__ __launch_bounds__(1024) void sum_rows_kernel(const float* __restrict__ partial, int rows, int n,
                                                        double* __restrict__ out) {
  __shared__ double sh[32][33];
  const int el = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + el;
  double s = 0.0;
  if (e < n)
    for (int r = g; r < rows; r += 32) s += (double)partial[(size_t)r * n + e];
  sh[g][el] = s;
  __syncthreads();
  if (g == 0 && e < n) {
    double t = 0.0;
    for (int r = 0; r < 32; ++r) t += sh[r][el];
    out[e] = t;
  }
}

// fp32 results straight into (up to) two destinations: elements [0, n0) -> out0, [n0, n) -> out1
// (e.g. d gamma | d beta into the two parameters' gradient tensors); accumulation in double as above.
// EW elements x 1024 / EW row groups per workgroup: many rows of few columns take EW = 8.
template <int EW>
__global__ __launch_bounds__(1024) void sum_rows_f32_kernel(const float* __restrict__ partial, int ld, int rows, int n,
                                                            float* __restrict__ out0, int n0, float* __restrict__ out1) {
  constexpr int NG = 1024 / EW;
  __shared__ double sh[NG][EW + 1];
  const int el = threadIdx.x % EW, g = threadIdx.x / EW;
  const int e = blockIdx.x * EW + el;
  double s = 0.0;
  {   // eight rows per trip, unconditional loads, same order of additions
    const bool in = e < n;
    const float* base = partial + (in ? e : 0);
    for (int r = g; r < rows; r += 8 * NG) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = base[(size_t)(r + u * NG < rows ? r + u * NG : 0) * ld];
#pragma unroll
      for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(v[u]));
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (in && r + u * NG < rows) s += (double)v[u];
    }
  }
  sh[g][el] = s;
  __syncthreads();
  if (g == 0 && e < n) {
    double t = 0.0;
    for (int r = 0; r < NG; ++r) t += sh[r][el];
    if (e < n0) out0[e] = (float)t;
    else out1[e - n0] = (float)t;
  }
}

// The same sums for few rows of many columns (the window-attention d(bias) / d(tau) partials: 16 ... 147 rows of 24 576 ...
// 196 608 columns): a thread owns four consecutive columns (16-byte loads, a wave reads 1 KB of a row) and every RG-th row;
// the RG partial sums of a column meet in LDS and are added in a fixed order.  (The 32-column kernel above read 128-byte
// pieces of rows and left half its threads idle at 16 rows: 12.5 us for 12.6 MB.)
template <int RG>
__global__ __launch_bounds__(256) void sum_rows_f32_wide_kernel(const float* __restrict__ partial, int ld, int rows, int n,
                                                                float* __restrict__ out0, int n0, float* __restrict__ out1) {
  uz_sum_rows_wide_body<RG>(partial, ld, rows, n, out0, n0, out1, blockIdx.x);
}

// dx[lo pixel][c] = sum of the 2x2 fine pixels (backward of nearest-neighbour x2 upsampling)
template <typename T>
__global__ __launch_bounds__(256) void sum2x2_kernel(const T* __restrict__ du, int ldu, int N, int H, int W,
                                                     int C, T* __restrict__ dx, int lddx) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int CC = C / VEC;
  const long long total = (long long)N * H * W * CC;  // H, W: coarse grid
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(idx % CC);
    const long long u = idx / CC;
    const int w = (int)(u % W);
    const long long t = u / W;
    const int h = (int)(t % H);
    const int img = (int)(t / H);
    const size_t p00 = ((size_t)img * 2 * H + 2 * h) * (2 * W) + 2 * w;
    float s[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) s[i] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float v[VEC];
      load_f(du + (p00 + (k >> 1) * (2 * W) + (k & 1)) * ldu + cc * VEC, v);
#pragma unroll
      for (int i = 0; i < VEC; ++i) s[i] += v[i];
    }
    store_f(dx + (size_t)u * lddx + cc * VEC, s);
  }
}

int pix_grid(int P, int ppb) {
  long long g = ((long long)P + ppb - 1) / ppb;
  if (g > UZ_NUM_CU * 4) g = UZ_NUM_CU * 4;
  return g < 1 ? 1 : (int)g;
}
bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
int lpp_of(int nchunks) { return nchunks < 64 ? nchunks : 64; }

int check_gate(const char* who, int dtype, int P, int CH) {
  UZ_REQUIRE(dtype == UZ_F32 || dtype == UZ_BF16, "%s: bad dtype", who);
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(P > 0 && CH > 0 && CH % vec == 0, "%s: channels=%d must be a multiple of %d", who, CH, vec);
  const int nch = CH / vec;
  UZ_REQUIRE(pow2(nch) || (nch > 64 && nch % 64 == 0), "%s: channels/%d = %d must be a power of two or a multiple of 64", who, vec, nch);
  return UZ_OK;
}

}  // namespace

extern "C" int uz_attn_grid(int dtype, int P, int channels) {
  const int rc = check_gate("uz_attn_grid", dtype, P, channels);
  if (rc != UZ_OK) return rc;
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  return pix_grid(P, 256 / lpp_of(channels / vec));
}

extern "C" int uz_attn_psi_fwd(int dtype, const void* g1raw, int ldg, const void* x1raw, int ldx,
                               const float* vec_g, const float* vec_x, const float* wpsi, const float* bpsi,
                               int P, int F, float* q, float* partial, void* stream) {
  int rc = check_gate("uz_attn_psi_fwd", dtype, P, F);
  if (rc != UZ_OK) return rc;
  UZ_REQUIRE(g1raw && x1raw && vec_g && vec_x && wpsi && q && partial, "uz_attn_psi_fwd: null pointer");
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(pow2(lpp_of(F / vec)), "uz_attn_psi_fwd: F");
  const int grid = pix_grid(P, 256 / lpp_of(F / vec));
  GateVecs v{vec_g, vec_x, nullptr};
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UZ_BF16)
    hipLaunchKernelGGL((attn_psi_fwd_kernel<bf16_t>), dim3(grid), dim3(256), 0, s, (const bf16_t*)g1raw, ldg, (const bf16_t*)x1raw, ldx, v, wpsi, bpsi, P, F, q, partial);
  else
    hipLaunchKernelGGL((attn_psi_fwd_kernel<float>), dim3(grid), dim3(256), 0, s, (const float*)g1raw, ldg, (const float*)x1raw, ldx, v, wpsi, bpsi, P, F, q, partial);
  UZ_LAUNCH_CHECK("uz_attn_psi_fwd");
  return UZ_OK;
}

extern "C" int uz_attn_gate_fwd(int dtype, const void* x, int ldx, const float* q, const float* vec_q, int P,
                                int C, void* out, int ldo, void* stream) {
  UZ_REQUIRE(dtype == UZ_F32 || dtype == UZ_BF16, "uz_attn_gate_fwd: bad dtype");
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(x && q && vec_q && out && P > 0 && C % vec == 0 && ldx % vec == 0 && ldo % vec == 0, "uz_attn_gate_fwd: bad args");
  const long long total = (long long)P * (C / vec);
  long long g = (total + 255) / 256;
  if (g > UZ_NUM_CU * 8) g = UZ_NUM_CU * 8;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UZ_BF16)
    hipLaunchKernelGGL((attn_gate_fwd_kernel<bf16_t>), dim3((unsigned)g), dim3(256), 0, s, (const bf16_t*)x, ldx, q, vec_q, total, C, (bf16_t*)out, ldo);
  else
    hipLaunchKernelGGL((attn_gate_fwd_kernel<float>), dim3((unsigned)g), dim3(256), 0, s, (const float*)x, ldx, q, vec_q, total, C, (float*)out, ldo);
  UZ_LAUNCH_CHECK("uz_attn_gate_fwd");
  return UZ_OK;
}

extern "C" int uz_attn_bwd_psi(int dtype, const void* dout, int ldd, const void* x, int ldx, const float* q,
                               const float* vec_q, int P, int C, void* dxd, int lddx, float* dz,
                               float* partial, void* stream) {
  int rc = check_gate("uz_attn_bwd_psi", dtype, P, C);
  if (rc != UZ_OK) return rc;
  UZ_REQUIRE(dout && x && q && vec_q && dxd && dz && partial, "uz_attn_bwd_psi: null pointer");
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  const int grid = pix_grid(P, 256 / lpp_of(C / vec));
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UZ_BF16)
    hipLaunchKernelGGL((attn_bwd_psi_kernel<bf16_t>), dim3(grid), dim3(256), 0, s, (const bf16_t*)dout, ldd, (const bf16_t*)x, ldx, q, vec_q, P, C, (bf16_t*)dxd, lddx, dz, partial);
  else
    hipLaunchKernelGGL((attn_bwd_psi_kernel<float>), dim3(grid), dim3(256), 0, s, (const float*)dout, ldd, (const float*)x, ldx, q, vec_q, P, C, (float*)dxd, lddx, dz, partial);
  UZ_LAUNCH_CHECK("uz_attn_bwd_psi");
  return UZ_OK;
}

template <typename T>
static int attn_bwd_launch(int pass, const AttnBwdArgs& a, int grid_x, hipStream_t s) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int nch = a.F / VEC;
  const dim3 grid(grid_x, nch > 64 ? nch / 64 : 1);
  if (pass == 1) {
    hipLaunchKernelGGL((attn_bwd_kernel<T, 1>), grid, dim3(256), (size_t)256 * (4 * VEC + 1) * sizeof(float), s, a);
  } else {
    hipLaunchKernelGGL((attn_bwd_kernel<T, 2>), grid, dim3(256), 0, s, a);
  }
  UZ_LAUNCH_CHECK("uz_attn_bwd");
  return UZ_OK;
}

extern "C" int uz_attn_bwd_reduce(int dtype, const void* g1raw, int ldg, const void* x1raw, int ldx,
                                  const float* q, const float* dz, const float* wpsi, const float* vec_g,
                                  const float* vec_x, const float* vec_q, const double* a01, int P, int F,
                                  float* partial, void* stream) {
  int rc = check_gate("uz_attn_bwd_reduce", dtype, P, F);
  if (rc != UZ_OK) return rc;
  UZ_REQUIRE(g1raw && x1raw && q && dz && wpsi && vec_g && vec_x && vec_q && a01 && partial, "uz_attn_bwd_reduce: null pointer");
  AttnBwdArgs a{};
  a.g1 = g1raw; a.x1 = x1raw; a.q = q; a.dz = dz; a.wpsi = wpsi;
  a.v = GateVecs{vec_g, vec_x, vec_q};
  a.a01 = a01; a.partial = partial; a.inv_count = 1.0 / (double)P;
  a.P = P; a.F = F; a.ldg = ldg; a.ldx = ldx;
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  const int grid = pix_grid(P, 256 / lpp_of(F / vec));
  // rows of other blockIdx.y groups write disjoint columns of the same row; the dq column by y == 0
  hipStream_t s = (hipStream_t)stream;
  return dtype == UZ_BF16 ? attn_bwd_launch<bf16_t>(1, a, grid, s) : attn_bwd_launch<float>(1, a, grid, s);
}

extern "C" int uz_attn_bwd_apply(int dtype, const void* g1raw, int ldg, const void* x1raw, int ldx,
                                 const float* q, const float* dz, const float* wpsi, const float* vec_g,
                                 const float* vec_x, const float* vec_q, const double* a01,
                                 const double* totals, int P, int F, void* dg1raw, int lddg, void* dx1raw,
                                 int lddx, void* stream) {
  int rc = check_gate("uz_attn_bwd_apply", dtype, P, F);
  if (rc != UZ_OK) return rc;
  UZ_REQUIRE(g1raw && x1raw && q && dz && wpsi && vec_g && vec_x && vec_q && a01 && totals && dg1raw && dx1raw,
             "uz_attn_bwd_apply: null pointer");
  AttnBwdArgs a{};
  a.g1 = g1raw; a.x1 = x1raw; a.q = q; a.dz = dz; a.wpsi = wpsi;
  a.v = GateVecs{vec_g, vec_x, vec_q};
  a.a01 = a01; a.tot = totals; a.dg1 = dg1raw; a.dx1 = dx1raw; a.inv_count = 1.0 / (double)P;
  a.P = P; a.F = F; a.ldg = ldg; a.ldx = ldx; a.lddg = lddg; a.lddx = lddx;
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  const int grid = pix_grid(P, 256 / lpp_of(F / vec));
  hipStream_t s = (hipStream_t)stream;
  return dtype == UZ_BF16 ? attn_bwd_launch<bf16_t>(2, a, grid, s) : attn_bwd_launch<float>(2, a, grid, s);
}

extern "C" int uz_sum_rows(const float* partial, int rows, int n, double* out, void* stream) {
  UZ_REQUIRE(partial && out && rows > 0 && n > 0, "uz_sum_rows: bad args");
  hipLaunchKernelGGL(sum_rows_kernel, dim3(uz_cdiv(n, 32)), dim3(1024), 0, (hipStream_t)stream, partial, rows, n, out);
  UZ_LAUNCH_CHECK("uz_sum_rows");
  return UZ_OK;
}

extern "C" int uz_sum_rows_f32_ld(const float* partial, int ld, int rows, int n, float* out0, int n0, float* out1,
                                  void* stream) {
  UZ_REQUIRE(partial && out0 && rows > 0 && n > 0 && ld >= n && n0 >= 0 && n0 <= n && (out1 || n0 == n),
             "uz_sum_rows_f32: bad args");
  if (const int rg = uz_sum_rows_wide_rg(partial, ld, rows, n)) {
    if (rg == 16)
      hipLaunchKernelGGL(sum_rows_f32_wide_kernel<16>, dim3(uz_cdiv(n, 64)), dim3(256), 0, (hipStream_t)stream, partial, ld, rows,
                         n, out0, n0, out1);
    else
      hipLaunchKernelGGL(sum_rows_f32_wide_kernel<4>, dim3(uz_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, partial, ld, rows,
                         n, out0, n0, out1);
  } else if (rows >= 256 && n <= 2048)
    hipLaunchKernelGGL(sum_rows_f32_kernel<8>, dim3(uz_cdiv(n, 8)), dim3(1024), 0, (hipStream_t)stream, partial, ld, rows, n,
                       out0, n0, out1);
  else
    hipLaunchKernelGGL(sum_rows_f32_kernel<32>, dim3(uz_cdiv(n, 32)), dim3(1024), 0, (hipStream_t)stream, partial, ld, rows, n,
                       out0, n0, out1);
  UZ_LAUNCH_CHECK("uz_sum_rows_f32");
  return UZ_OK;
}

extern "C" int uz_sum_rows_f32(const float* partial, int rows, int n, float* out0, int n0, float* out1, void* stream) {
  return uz_sum_rows_f32_ld(partial, n, rows, n, out0, n0, out1, stream);
}

extern "C" int uz_sum2x2(int dtype, const void* du, int ldu, int N, int H, int W, int C, void* dx, int lddx,
                         void* stream) {
  UZ_REQUIRE(dtype == UZ_F32 || dtype == UZ_BF16, "uz_sum2x2: bad dtype");
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(du && dx && N > 0 && H > 0 && W > 0 && C % vec == 0 && ldu % vec == 0 && lddx % vec == 0 && ldu >= C && lddx >= C,
             "uz_sum2x2: bad args");
  const long long total = (long long)N * H * W * (C / vec);
  long long g = (total + 255) / 256;
  if (g > UZ_NUM_CU * 8) g = UZ_NUM_CU * 8;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UZ_BF16)
    hipLaunchKernelGGL((sum2x2_kernel<bf16_t>), dim3((unsigned)g), dim3(256), 0, s, (const bf16_t*)du, ldu, N, H, W, C, (bf16_t*)dx, lddx);
  else
    hipLaunchKernelGGL((sum2x2_kernel<float>), dim3((unsigned)g), dim3(256), 0, s, (const float*)du, ldu, N, H, W, C, (float*)dx, lddx);
  UZ_LAUNCH_CHECK("uz_sum2x2");
  return UZ_OK;
}
