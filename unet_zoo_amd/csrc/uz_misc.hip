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
