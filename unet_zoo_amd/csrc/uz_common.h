// Internal helpers shared by the gfx950 kernels of libunetzoo_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/unetzoo_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

// ---- measurement hook (uz_profile_arm / uz_profile_disarm, include/unetzoo_hip.h) --------------------------------------
// While a thread is armed, the kernel launches of this library record the caller's event pair at the kernel's own begin
// and end (hipExtLaunchKernelGGL): the first launch records both, later launches of the same armed scope move only the
// end event -- the pair then brackets first-kernel-begin to last-kernel-end without the dispatch latency that events
// recorded around a launch include.  Unarmed (always so inside graph capture) launches are plain <<< >>> launches.
#include <hip/hip_ext.h>
bool uz_prof_take(hipEvent_t* e0, hipEvent_t* e1);   // true while armed; *e0 = nullptr from the second launch on
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kern, grid, block, shm, stream, ...)                                              \
  do {                                                                                                       \
    hipEvent_t uz_e0_, uz_e1_;                                                                               \
    if (uz_prof_take(&uz_e0_, &uz_e1_))                                                                      \
      hipExtLaunchKernelGGL(kern, dim3(grid), dim3(block), (shm), (stream), uz_e0_, uz_e1_, 0, __VA_ARGS__);   \
    else                                                                                                     \
      kern<<<dim3(grid), dim3(block), (shm), (stream)>>>(__VA_ARGS__);                                       \
  } while (0)

#define UZ_WAVE 64
#define UZ_NUM_CU_HW 256
#define UZ_NUM_XCD 8
// CUs the plans size their grids for: all 256, or fewer while uz_set_cu_reserve(n) holds some back for the kernels of
// another stream (RCCL's all-reduce beside the backward: the convolution / GEMM / weight-gradient kernels are persistent
// grids of one 160 KB workgroup per CU, which leave a collective no CU to start on until the next kernel boundary)
int uz_num_cu();
#define UZ_NUM_CU (uz_num_cu())

void uz_set_error(const char* fmt, ...);

#define UZ_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      uz_set_error(__VA_ARGS__);         \
      return UZ_EINVAL;                  \
    }                                    \
  } while (0)

#define UZ_LAUNCH_CHECK(name)                                               \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      uz_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));  \
      return (int)e__;                                                      \
    }                                                                       \
  } while (0)

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
  static constexpr int VEC = 4;  // elements per 16 bytes
};
template <> struct ElemTraits<bf16_t> {
  static constexpr int VEC = 8;
};

// 16-byte vector of T, loaded/stored as one dwordx4.
template <typename T> struct alignas(16) Vec16 {
  T v[ElemTraits<T>::VEC];
};

template <typename T> __device__ __forceinline__ Vec16<T> ld16(const T* p) {
  return *reinterpret_cast<const Vec16<T>*>(p);
}
template <typename T> __device__ __forceinline__ void st16(T* p, const Vec16<T>& v) {
  *reinterpret_cast<Vec16<T>*>(p) = v;
}
template <typename T> __device__ __forceinline__ Vec16<T> zero16() {
  Vec16<T> z;
#pragma unroll
  for (int i = 0; i < ElemTraits<T>::VEC; ++i) z.v[i] = (T)0.0f;
  return z;
}

static inline int uz_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Row sums of few rows of many columns (uz_sum_rows_f32's wide form, uz_attn.hip; the same body serves the batched launch of
// uz_colsum.hip, so one buffer's sums do not depend on which launch formed them): workgroup `block` of 256 threads owns
// 256 / RG column quads, thread (cq, g) every RG-th row of its quad; the RG partial sums meet in LDS, fixed order.
template <int RG>
__device__ __forceinline__ void uz_sum_rows_wide_body(const float* __restrict__ partial, int ld, int rows, int n,
                                                      float* __restrict__ out0, int n0, float* __restrict__ out1, int block) {
  constexpr int CQ = 256 / RG;   // column quads per workgroup
  __shared__ double sh[RG][CQ][4 + 1];
  const int cq = threadIdx.x % CQ, g = threadIdx.x / CQ;
  const int e = (block * CQ + cq) * 4;
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  if (e < n) {
    int r = g;
    for (; r + 3 * RG < rows; r += 4 * RG) {   // four independent loads in flight
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(partial + (size_t)(r + u * RG) * ld + e);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s[0] += (double)v[u].x;
        s[1] += (double)v[u].y;
        s[2] += (double)v[u].z;
        s[3] += (double)v[u].w;
      }
    }
    for (; r < rows; r += RG) {
      const float4 v = *reinterpret_cast<const float4*>(partial + (size_t)r * ld + e);
      s[0] += (double)v.x;
      s[1] += (double)v.y;
      s[2] += (double)v.z;
      s[3] += (double)v.w;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) sh[g][cq][i] = s[i];
  __syncthreads();
  if (g == 0 && e < n) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      double t = 0.0;
      for (int r = 0; r < RG; ++r) t += sh[r][cq][i];
      const int idx = e + i;
      if (idx < n0) out0[idx] = (float)t;
      else out1[idx - n0] = (float)t;
    }
  }
}
// which form uz_sum_rows_f32_ld takes for a buffer: 16 / 4 = the wide form with that many row groups, 0 = the column forms
static inline int uz_sum_rows_wide_rg(const float* partial, int ld, int rows, int n) {
  if (rows <= 256 && n >= 16384 && n % 4 == 0 && ld % 4 == 0 && ((uintptr_t)partial & 15) == 0) return rows >= 64 ? 16 : 4;
  return 0;
}

// Tuning / ablation switches for in-process A/B measurements (tools/kbench.py).  They exist only in a library
// built with -DUZ_ABLATE (make ABLATE=1): the shipped libunetzoo_hip.so never reads the environment, its flags are
// the constant 0, so no environment variable can change what a kernel computes or how it is launched.
#include <stdlib.h>
// Device code reads a kernel's `flags` argument through UZ_KFLAGS(a): the argument itself in the ablation build, the
// literal 0 in the shipped build, so that every ablation branch (and the register copies its control flow costs: 32
// v_mov_b64 per (tap, slab) unit in the direct convolution) is compiled out of the shipped kernels.
#ifdef UZ_ABLATE
#define UZ_KFLAGS(a) ((a).flags)
#else
#define UZ_KFLAGS(a) 0
#endif
#ifdef UZ_ABLATE
static inline int uz_tune_flags() {
  const char* e = getenv("UZ_TUNE");
  return e ? atoi(e) : 0;
}
static inline const char* uz_ablate_env(const char* name) { return getenv(name); }
#else
static inline int uz_tune_flags() { return 0; }
static inline const char* uz_ablate_env(const char*) { return nullptr; }
#endif

// direct 3x3 convolution (uz_conv3x3.hip), dispatched from uz_conv_igemm()
struct UzDirectPlan {
  int tw, bn, bres, th_n, tw_n, ntiles, tiles_n, grid_m;
  int ppcfg;   // bres == 3: the ping-pong configuration (UZ_PP_*)
  int ksplit, cps;   // bres == 3: split-K plan of the ping-pong kernel (1: none)
};
int uz_direct_plan(const uz_conv_desc* d, UzDirectPlan* p);
// operands of the BatchNorm-backward reduction fused into the epilogue (uz_conv_igemm_bnred)
struct UzBnRed {
  const void* y;
  int ldy;
  const float *scale, *shift, *mean, *invstd;
};
// per-input-channel (scale, shift) of the BatchNorm + ReLU applied to x on its way into LDS (uz_conv_igemm_xf, uz_wgrad_xf)
struct UzXf {
  const float *scale, *shift;
};
int uz_direct_launch(const uz_conv_desc* d, const UzDirectPlan& p, const void* x, const void* w,
                     const float* bias, void* y, float* stats, hipStream_t s, const UzBnRed* br = nullptr,
                     float* part = nullptr,   // part: split-K partial tiles (ping-pong plans with ksplit > 1)
                     const UzXf* xf = nullptr);

// direct 3x3 convolution, ping-pong schedule on 512-pixel x 128-channel tiles (uz_conv3x3_pp.hip); uz_direct_plan()
// hands the descriptors it takes over with bres = 3
enum { UZ_PP_512 = 0,      // 16 x 32 pixels x 128 channels
       UZ_PP_512X64 = 1,   // 16 x 32 pixels x 64 channels
       UZ_PP_256 = 2,      // 8 x 32 pixels x 128 channels
       UZ_PP_256W16 = 3,   // 16 x 16 pixels x 128 channels
       UZ_PP_128W16 = 4 }; // 8 x 16 pixels x 128 channels
struct UzPpPlan {
  int cfg, bn, th_n, tw_n, ntiles, tiles_n, grid_m;
  int ksplit, cps;   // split-K over the channel slabs (only with a workspace): ksplit ranges of cps slabs
};
int uz_pp_plan(const uz_conv_desc* d, UzPpPlan* p);
int uz_pp_launch(const uz_conv_desc* d, const UzPpPlan& p, const void* x, const void* w, const float* bias, void* y,
                 float* stats, hipStream_t s, const UzBnRed* br = nullptr, float* part = nullptr, const UzXf* xf = nullptr);
int uz_pp_xf_channels(const UzPpPlan& p);   // input channels the XF form of this plan's configuration takes (0: none)

// 3x3 weight gradient with LDS-DMA pipeline (uz_wgrad3x3.hip), dispatched from uz_wgrad()
struct UzWgrad2Plan {
  int big, one_tap, gather, kw, kr, tiles_i, tiles_j, kg, units, upb, split, nslabs, H, W;
  int wide9;   // nine taps on a 128 (dy) x 64 (x) channel tile
  int v9, bi;  // v9 = 1: the row-walk nine-tap kernel of uz_wgrad9.hip on bi x 64 channel tiles takes the descriptor
};
// row-walk nine-tap weight gradient (uz_wgrad9.hip); uz_wgrad3x3_plan() tries it first
int uz_wgrad9_plan(const uz_wgrad_desc* d, UzWgrad2Plan* p);
int uz_wgrad9_launch(const uz_wgrad_desc* d, const UzWgrad2Plan& p, const void* L, const void* R, float* slab, hipStream_t s,
                     const UzXf* xf = nullptr);   // xf: R is read through a BatchNorm + ReLU (uz_wgrad_xf)
const char* uz_wgrad9_name(const UzWgrad2Plan& p);
// 2 x 2 gather (ConvTranspose2d k2 s2 / PatchExpand) weight gradient, four taps per workgroup (uz_wgrad_g4.hip): v9 = 2
int uz_wgrad_g4_plan(const uz_wgrad_desc* d, UzWgrad2Plan* p);
int uz_wgrad_g4_launch(const uz_wgrad_desc* d, const UzWgrad2Plan& p, const void* L, const void* R, float* slab, hipStream_t s);
int uz_wgrad3x3_plan(const uz_wgrad_desc* d, UzWgrad2Plan* p, int batch = 1);   // batch > 1: uz_wgrad_batched (one-tap only)
int uz_wgrad3x3_launch(const uz_wgrad_desc* d, const UzWgrad2Plan& p, const void* L, const void* R,
                       float* slab, hipStream_t s, int batch = 1, long long lb_bytes = 0, long long rb_bytes = 0,
                       long long slab_stride = 0,   // floats between the problems' slabs (0: dense)
                       int batch2 = 1, long long lb2_bytes = 0, long long rb2_bytes = 0,   // batch = ALL problems (outer * inner)
                       const UzXf* xf = nullptr);   // row-walk plans only (uz_wgrad_xf)

// several one-tap problems in one launch (uz_wgrad_multi): plans from uz_wgrad3x3_plan() with one_tap && !gather && !v9, all
// of one tile shape (big); at most uz_wgrad3x3_multi_max() per launch
struct UzWgradMultiItem {
  const uz_wgrad_desc* d;
  UzWgrad2Plan p;
  const void *L, *R;
  float* slab;
};
int uz_wgrad3x3_multi_max();
int uz_wgrad3x3_multi_launch(const UzWgradMultiItem* items, int n, hipStream_t s);

// LDS-DMA pixel-major GEMM (uz_gemm_dma.hip): 1x1 / ConvTranspose fwd + dgrad, dispatched from uz_conv_igemm()
struct UzGemmPlan {
  int bn, bm, nst, tiles_m, tiles_n, grid_m;
  int ksplit, cps;   // split-K over the 128-byte K slabs (only with a workspace): ksplit ranges of cps slabs
};
int uz_gemm_dma_plan(const uz_conv_desc* d, UzGemmPlan* p);
long long uz_gemm_dma_workspace_bytes(const uz_conv_desc* d);   // fp32 partial tiles of a split-K plan, else 0
// part: the split-K workspace (ksplit x M x Nout floats) -- the caller then runs the reduce pass; without it the unsplit plan
int uz_gemm_dma_launch(const uz_conv_desc* d, const UzGemmPlan& p, const void* x, const void* w,
                       const float* bias, void* y, float* stats, hipStream_t s, const void* res = nullptr,
                       int ldres = 0, const UzBnRed* br = nullptr, float* part = nullptr);
