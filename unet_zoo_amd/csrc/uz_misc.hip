// Small streaming kernels added in round 2 (gfx950): residual sums with ReLU (MultiResUNet / ResPath blocks,
// unet_zoo/models/multiresunet.py:79-82, 127-137), 16 bytes per lane, one pixel row of channels per group of lanes.
#include "uz_common.h"

namespace {

template <typename T> __device__ __forceinline__ void ldf(const T* p, float* v) {
  const Vec16<T> r = ld16(p);
#pragma unroll
  for (int i = 0; i < ElemTraits<T>::VEC; ++i) v[i] = (float)r.v[i];
}
template <typename T> __device__ __forceinline__ void stf(T* p, const float* v) {
  Vec16<T> r;
#pragma unroll
  for (int i = 0; i < ElemTraits<T>::VEC; ++i) r.v[i] = (T)v[i];
  st16(p, r);
}

// out = relu(a + b)   (b == nullptr: relu(a))
template <typename T>
__global__ __launch_bounds__(256) void add_relu_kernel(const T* __restrict__ a, int lda, const T* __restrict__ b, int ldb,
                                                       T* __restrict__ out, int ldo, long long P, int C) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int CC = C / VEC;
  const long long total = P * CC;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int c0 = (int)(idx % CC) * VEC;
    const long long p = idx / CC;
    float u[VEC], v[VEC];
    ldf(a + p * lda + c0, u);
    if (b != nullptr) {
      ldf(b + p * ldb + c0, v);
#pragma unroll
      for (int i = 0; i < VEC; ++i) u[i] += v[i];
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) u[i] = fmaxf(u[i], 0.f);
    stf(out + p * ldo + c0, u);
  }
}

// dx = g where the forward's output is positive, else 0
template <typename T>
__global__ __launch_bounds__(256) void relu_bwd_kernel(const T* __restrict__ out, int ldo, const T* __restrict__ g, int ldg,
                                                       T* __restrict__ dx, int lddx, long long P, int C) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int CC = C / VEC;
  const long long total = P * CC;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int c0 = (int)(idx % CC) * VEC;
    const long long p = idx / CC;
    float o[VEC], v[VEC];
    ldf(out + p * ldo + c0, o);
    ldf(g + p * ldg + c0, v);
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = o[i] > 0.f ? v[i] : 0.f;
    stf(dx + p * lddx + c0, v);
  }
}

int check2(const char* fn, int dtype, long long P, int C, int l0, int l1, int l2, bool has1) {
  UZ_REQUIRE(dtype == UZ_F32 || dtype == UZ_BF16, "%s: bad dtype", fn);
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(P > 0 && C > 0 && C % vec == 0, "%s: C=%d must be a positive multiple of %d", fn, C, vec);
  UZ_REQUIRE(l0 % vec == 0 && l0 >= C && l2 % vec == 0 && l2 >= C && (!has1 || (l1 % vec == 0 && l1 >= C)), "%s: bad ld", fn);
  return UZ_OK;
}

int grid_for(long long units) {
  long long g = (units + 255) / 256;
  const long long cap = (long long)UZ_NUM_CU * 8;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int uz_add_relu(int dtype, const void* a, int lda, const void* b, int ldb, void* out, int ldo, long long P,
                           int C, void* stream) {
  const int rc = check2("uz_add_relu", dtype, P, C, lda, ldb, ldo, b != nullptr);
  if (rc != UZ_OK) return rc;
  UZ_REQUIRE(a && out, "uz_add_relu: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  const int g = grid_for(P * (C / vec));
  if (dtype == UZ_BF16)
    hipLaunchKernelGGL(add_relu_kernel<bf16_t>, dim3(g), dim3(256), 0, s, (const bf16_t*)a, lda, (const bf16_t*)b, ldb, (bf16_t*)out, ldo, P, C);
  else
    hipLaunchKernelGGL(add_relu_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)a, lda, (const float*)b, ldb, (float*)out, ldo, P, C);
  UZ_LAUNCH_CHECK("uz_add_relu");
  return UZ_OK;
}

extern "C" int uz_relu_bwd(int dtype, const void* out, int ldo, const void* g, int ldg, void* dx, int lddx, long long P,
                           int C, void* stream) {
  const int rc = check2("uz_relu_bwd", dtype, P, C, ldo, ldg, lddx, true);
  if (rc != UZ_OK) return rc;
  UZ_REQUIRE(out && g && dx, "uz_relu_bwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  const int gr = grid_for(P * (C / vec));
  if (dtype == UZ_BF16)
    hipLaunchKernelGGL(relu_bwd_kernel<bf16_t>, dim3(gr), dim3(256), 0, s, (const bf16_t*)out, ldo, (const bf16_t*)g, ldg, (bf16_t*)dx, lddx, P, C);
  else
    hipLaunchKernelGGL(relu_bwd_kernel<float>, dim3(gr), dim3(256), 0, s, (const float*)out, ldo, (const float*)g, ldg, (float*)dx, lddx, P, C);
  UZ_LAUNCH_CHECK("uz_relu_bwd");
  return UZ_OK;
}

// ---------------------------------------------------------------------------------------------
// Input pipeline (unet_zoo/data/datasets.py:40-59): transforms.Resize((512, 512)) of a PIL image -- Pillow's
// two-pass antialiased BILINEAR resample of 8-bit images in 22-bit fixed point, restated bit for bit --, ToTensor,
// Normalize(mean, std) for the image, `> 0.5` for the mask.  The coefficient tables (Pillow's precompute_coeffs +
// normalize_coeffs_8bpc, src/libImaging/Resample.c) come from the host; the passes are integer arithmetic.
// ---------------------------------------------------------------------------------------------
namespace {
constexpr int PIL_PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ int pil_clip8(int v) {
  v >>= PIL_PRECISION_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// dst[h][x][c] = clip8(2^21 + sum_t src[h][xmin(x) + t][c] * kk[x][t])      (ImagingResampleHorizontal_8bpc)
__global__ __launch_bounds__(256) void pil_resample_h_kernel(const unsigned char* __restrict__ src, int H, int Win, int C,
                                                             const int* __restrict__ bounds, const int* __restrict__ kk,
                                                             int ksize, int Wout, unsigned char* __restrict__ dst) {
  const long long total = (long long)H * Wout;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(idx % Wout);
    const long long h = idx / Wout;
    const int xmin = bounds[2 * x], xmax = bounds[2 * x + 1];
    const int* k = kk + (long long)x * ksize;
    const unsigned char* row = src + (h * Win + xmin) * C;
    for (int c = 0; c < C; ++c) {
      int ss = 1 << (PIL_PRECISION_BITS - 1);
      for (int t = 0; t < xmax; ++t) ss += (int)row[t * C + c] * k[t];
      dst[(h * Wout + x) * C + c] = (unsigned char)pil_clip8(ss);
    }
  }
}

// vertical pass fused with ToTensor + Normalize (mode 0: out[c][y][x] = (p / 255 - mean[c]) / std[c], fp32 in torch's
// operation order) or with the mask threshold (mode 1: out = p / 255 > 0.5)      (ImagingResampleVertical_8bpc)
__global__ __launch_bounds__(256) void pil_resample_v_kernel(const unsigned char* __restrict__ src, int Hin, int W, int C,
                                                             const int* __restrict__ bounds, const int* __restrict__ kk,
                                                             int ksize, int Hout, float m0, float m1, float m2, float s0,
                                                             float s1, float s2, int mode, float* __restrict__ out) {
  const long long total = (long long)Hout * W;
  const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(idx % W);
    const int y = (int)(idx / W);
    const int ymin = bounds[2 * y], ymax = bounds[2 * y + 1];
    const int* k = kk + (long long)y * ksize;
    for (int c = 0; c < C; ++c) {
      int ss = 1 << (PIL_PRECISION_BITS - 1);
      for (int t = 0; t < ymax; ++t) ss += (int)src[((long long)(ymin + t) * W + x) * C + c] * k[t];
      const float v = (float)pil_clip8(ss) / 255.0f;
      out[((long long)c * Hout + y) * W + x] = mode == 0 ? (v - mean[c]) / sd[c] : (v > 0.5f ? 1.0f : 0.0f);
    }
  }
}
// ---- measurement: the clock the chip holds under a dense MFMA stream ---------------------------------------------------
// Eight waves per CU (two per SIMD) issue v_mfma_f32_16x16x32_bf16 back to back on pseudo-random operands kept in
// registers; wave 0 of each workgroup stamps s_memtime (shader clock) and s_memrealtime (100 MHz) around the loop.
// clock = d(memtime) / d(memrealtime) x 100 MHz.  MI355X_MICROARCH.md, "DVFS give-back" (5), (6): this is the quantity
// that differs between devices (1.51-1.69 GHz measured there on one binary), and with it every MFMA-bound kernel.
__global__ __launch_bounds__(512, 2) void clock_probe_kernel(unsigned long long* __restrict__ out, int iters, unsigned seed) {
  const int tid = threadIdx.x;
  unsigned h = (blockIdx.x * 512u + tid) * 2654435761u + seed;
  bf16x8 a[8], b[8];   // eight independent operand pairs: every MFMA of the loop body toggles its operand lines
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      h = h * 1664525u + 1013904223u;
      a[k][e] = (bf16_t)(((int)(h >> 8) & 0xffff) * (2.0f / 65536.0f) - 1.0f);
      h = h * 1664525u + 1013904223u;
      b[k][e] = (bf16_t)(((int)(h >> 8) & 0xffff) * (2.0f / 65536.0f) - 1.0f);
    }
  f32x4 acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[k], b[k], acc[k], 0, 0, 0);
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int k = 1; k < 8; ++k) s += acc[k];
  asm volatile("" ::"v"(s));
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) {
    out[2 * blockIdx.x + 0] = t1 - t0;
    out[2 * blockIdx.x + 1] = r1 - r0;
  }
}
}  // namespace

extern "C" int uz_clock_probe(int iters, void* out_pairs, int workgroups, void* stream) {
  UZ_REQUIRE(out_pairs != nullptr && iters > 0 && workgroups > 0 && workgroups <= 4096, "uz_clock_probe: bad arguments");
  hipLaunchKernelGGL(clock_probe_kernel, dim3(workgroups), dim3(512), 0, (hipStream_t)stream,
                     static_cast<unsigned long long*>(out_pairs), iters, 12345u);
  UZ_LAUNCH_CHECK("uz_clock_probe");
  return UZ_OK;
}

extern "C" int uz_pil_resample_h_u8(const void* src, int H, int Win, int C, const int* bounds, const int* kk, int ksize,
                                    int Wout, void* dst, void* stream) {
  UZ_REQUIRE(src && bounds && kk && dst && H > 0 && Win > 0 && Wout > 0 && C >= 1 && C <= 4 && ksize >= 1,
             "uz_pil_resample_h_u8: bad arguments");
  hipLaunchKernelGGL(pil_resample_h_kernel, dim3(grid_for((long long)H * Wout)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned char*)src, H, Win, C, bounds, kk, ksize, Wout, (unsigned char*)dst);
  UZ_LAUNCH_CHECK("uz_pil_resample_h_u8");
  return UZ_OK;
}

extern "C" int uz_pil_resample_v_f32(const void* src, int Hin, int W, int C, const int* bounds, const int* kk, int ksize,
                                     int Hout, const float* mean_host, const float* std_host, int mode, float* out,
                                     void* stream) {
  UZ_REQUIRE(src && bounds && kk && out && Hin > 0 && W > 0 && Hout > 0 && C >= 1 && C <= 3 && ksize >= 1 &&
                 (mode == 0 || mode == 1), "uz_pil_resample_v_f32: bad arguments");
  UZ_REQUIRE(mode == 1 || (mean_host && std_host), "uz_pil_resample_v_f32: mode 0 needs mean and std");
  float m[3] = {0.f, 0.f, 0.f}, s[3] = {1.f, 1.f, 1.f};
  if (mode == 0)
    for (int c = 0; c < C; ++c) {
      m[c] = mean_host[c];
      s[c] = std_host[c];
    }
  hipLaunchKernelGGL(pil_resample_v_kernel, dim3(grid_for((long long)Hout * W)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned char*)src, Hin, W, C, bounds, kk, ksize, Hout, m[0], m[1], m[2], s[0], s[1], s[2], mode,
                     out);
  UZ_LAUNCH_CHECK("uz_pil_resample_v_f32");
  return UZ_OK;
}
