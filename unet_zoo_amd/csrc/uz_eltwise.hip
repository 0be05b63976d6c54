// Bandwidth-bound kernels of the UNet hot path for gfx950 (MI355X): BatchNorm statistics
// finalisation, BN-apply + ReLU (+ 2x2 max-pool) forward, its two-pass backward, the 1x1 output
// convolution, weight re-packing and the input im2col.  All activations NHWC with 16-byte
// (8 x bf16 / 4 x fp32) accesses per lane; per-channel reductions go wave-shuffle -> LDS -> one
// atomic per block and channel.
//
// Reference call sites replaced: nn.BatchNorm2d / nn.ReLU / nn.MaxPool2d in DoubleConv and
// DownSample (unet_zoo/models/common_layers.py:29-33, 90-95), OutConv (:125), and their
// autograd backward (unet_zoo/utils/training_loop.py:119).
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

// ------------------------------------------------------------------------------------------
// BN finalize: reduce the conv kernel's partial rows and derive scale/shift + running stats.
// One block of 1024 threads per EW channels: thread (c = t % EW, g = t / EW) sums rows g, g + 1024/EW, ...
// (EW = 8 when the rows are many: the kernel is a chain of dependent loads, 128 row groups shorten it)
// ------------------------------------------------------------------------------------------
template <int EW>
__global__ __launch_bounds__(1024) void bn_finalize_kernel(
    const float* __restrict__ part, int grid_m, int C, double count, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, float momentum, float* running_mean,
    float* running_var, float* scale, float* shift, float* mean, float* invstd) {
  constexpr int NG = 1024 / EW;
  __shared__ double sh[2][NG][EW + 1];
  const int cl = threadIdx.x % EW, g = threadIdx.x / EW;
  const int c = blockIdx.x * EW + cl;
  double a1 = 0.0, a2 = 0.0;
  const int cq = c < C ? c : 0;
  // the tail's operands, requested before the reduction (they were three dependent round trips behind `if (g == 0)`, in a
  // kernel that is nothing but latency); unconditional, made opaque below so that they stay in front
  float gmv = gamma[cq], btv = beta[cq];
  float rmv = (running_mean != nullptr ? running_mean : gamma)[cq], rvv = (running_var != nullptr ? running_var : gamma)[cq];
  // four row groups per trip, unconditional loads (row 0 past the end, dropped by the select), added in the same order as one
  // group per trip: the sums are bit-identical, the dependent round trips a quarter
  for (int r = g; r < grid_m; r += 4 * NG) {
    float p1[4], p2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int rr = r + k * NG < grid_m ? r + k * NG : 0;
      p1[k] = part[((size_t)rr * 2 + 0) * C + cq];
      p2[k] = part[((size_t)rr * 2 + 1) * C + cq];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(p1[k]), "+v"(p2[k]));
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (r + k * NG < grid_m && c < C) {
        a1 += (double)p1[k];
        a2 += (double)p2[k];
      }
  }
  asm volatile("" : "+v"(gmv), "+v"(btv), "+v"(rmv), "+v"(rvv));
  sh[0][g][cl] = a1;
  sh[1][g][cl] = a2;
  __syncthreads();
  // fixed-order pairwise tree over the NG row groups (the pairing does not depend on timing: deterministic); a single
  // thread walking the NG partial sums cost 4 of this kernel's 5 microseconds
#pragma unroll
  for (int st = NG / 2; st > 0; st >>= 1) {
    if (g < st) {
      sh[0][g][cl] += sh[0][g + st][cl];
      sh[1][g][cl] += sh[1][g + st][cl];
    }
    __syncthreads();
  }
  if (g == 0 && c < C) {
    const double t1 = sh[0][0][cl], t2 = sh[1][0][cl];
    const double m = t1 / count;
    double var = t2 / count - m * m;
    if (var < 0.0) var = 0.0;
    const double istd = 1.0 / sqrt(var + (double)eps);
    const float sc = (float)((double)gmv * istd);
    scale[c] = sc;
    shift[c] = (float)((double)btv - m * (double)gmv * istd);
    mean[c] = (float)m;
    invstd[c] = (float)istd;
    if (running_mean != nullptr) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      running_mean[c] = (float)((1.0 - (double)momentum) * (double)rmv + (double)momentum * m);
      running_var[c] = (float)((1.0 - (double)momentum) * (double)rvv + (double)momentum * unbiased);
    }
  }
}

// ------------------------------------------------------------------------------------------
// A finalize INSIDE its consumer's launch (uz_bn_relu_add_apply_fin, uz_bn_relu_bwd_apply_fin; round 5).  The finalize
// kernels above are 5 us launches between two large kernels, 36 of them per unet step and 448 per u2net step, and they
// cannot be batched: each sits on the dependency chain between the kernel that produced the partial rows and the element
// pass that needs the result.  Here the first `nfin` workgroups of the element pass do the finalize's work -- the same row
// groups and the same pairing tree as bn_finalize_kernel<EW> / bn_bwd_finalize_kernel<EW>, 256 threads standing in for its
// 1024, so the same bits -- publish the vectors and add 1 to a flag with release order; every workgroup then waits for
// flag == nfin with acquire order before it reads them.  Workgroups are dispatched in index order, so the finalizing
// ones are resident before any waiting one: the wait cannot starve them.  The flag must be zero at launch (the caller's
// per-step memset of its flag arena) and is left at nfin.
// ------------------------------------------------------------------------------------------
constexpr int FIN_SH_DOUBLES = 2 * 128 * 9;   // EW = 8: [2][128][9]; EW = 32: [2][32][33] is smaller

template <int EW>
__device__ __forceinline__ void fin_row_sums(const float* __restrict__ part, int rows, int C, int blk, int tid, double* sh) {
  constexpr int NG = 1024 / EW, LD = EW + 1, GS = 256 / EW;
  // thread tid stands for the finalize kernel's threads tid + 256 k, k = 0..3: the same channel (256 is a multiple of EW), row
  // groups tid / EW + k GS.  The loads of the four are issued together (one after the other they were 8 memory round trips on
  // the critical path of the whole launch); each keeps its own order of additions.
  const int cl = tid % EW, g0 = tid / EW, c = blk * EW + cl;
  double a1[4] = {0.0, 0.0, 0.0, 0.0}, a2[4] = {0.0, 0.0, 0.0, 0.0};
  if (c < C) {
    for (int r0 = g0; r0 < rows; r0 += NG) {
      float x1[4], x2[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int r = r0 + k * GS;
        const int rc = r < rows ? r : 0;
        x1[k] = part[((size_t)rc * 2 + 0) * C + c];
        x2[k] = part[((size_t)rc * 2 + 1) * C + c];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (r0 + k * GS < rows) {
          a1[k] += (double)x1[k];
          a2[k] += (double)x2[k];
        }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    sh[(0 * NG + g0 + k * GS) * LD + cl] = a1[k];
    sh[(1 * NG + g0 + k * GS) * LD + cl] = a2[k];
  }
  __syncthreads();
#pragma unroll
  for (int st = NG / 2; st > 0; st >>= 1) {
    for (int v = tid; v < st * EW; v += 256) {
      const int vl = v % EW, g = v / EW;
      sh[(0 * NG + g) * LD + vl] += sh[(0 * NG + g + st) * LD + vl];
      sh[(1 * NG + g) * LD + vl] += sh[(1 * NG + g + st) * LD + vl];
    }
    __syncthreads();
  }
}

struct BnFin {
  const float* part;   // [rows][2][C] as the producing kernel left them
  int rows, ew;        // ew: 8 or 32, the choice uz_bn_finalize / uz_bn_bwd_finalize make for this shape
  double count;
  const float *gamma, *beta;
  float eps, momentum;
  float *running_mean, *running_var;   // may be null
  float* vec;          // forward: [4][C] = scale, shift, mean, invstd (out)
  double* sums;        // backward: [2][C] (out)
  float *dgamma, *dbeta;   // backward (out, may be null)
  int* flag;
};

// The hand-over uses no cache-wide fence.  (Measured: an acquire LOAD in the poll loop -- a cache invalidate per poll -- ran
// the unet step at 10.3 ms instead of 6.3; relaxed polls + one agent-scope release / acquire fence per workgroup still cost
// ~17 us per launch: 2048 workgroups each invalidating their XCD's L2 under the streaming loads of the others.)  Instead the
// few words handed over are themselves written and read at agent scope -- relaxed atomic stores / loads, i.e. sc1 accesses
// that are coherent across the XCDs' L2s -- and ordered by waiting for the stores' completion before the flag add.
#ifndef UZ_FIN_SLEEP
#define UZ_FIN_SLEEP 4
#endif
template <typename V> __device__ __forceinline__ void fin_store(V* p, V v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename V> __device__ __forceinline__ V fin_load(const V* p) {
  return __hip_atomic_load(const_cast<V*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void fin_publish(int* flag, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores have reached the coherence point
  __syncthreads();
  if (tid == 0) __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void fin_wait(int* flag, int nfin, int tid) {
  if (tid == 0)
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nfin) __builtin_amdgcn_s_sleep(UZ_FIN_SLEEP);
  __syncthreads();
}

// forward: what bn_finalize_kernel<EW> block `blk` does
template <int EW>
__device__ __forceinline__ void fin_forward(const BnFin& f, int C, int blk, int tid, double* sh) {
  constexpr int NG = 1024 / EW, LD = EW + 1;
  fin_row_sums<EW>(f.part, f.rows, C, blk, tid, sh);
  const int c = blk * EW + tid;
  if (tid < EW && c < C) {
    const double t1 = sh[(0 * NG) * LD + tid], t2 = sh[(1 * NG) * LD + tid];
    const double m = t1 / f.count;
    double var = t2 / f.count - m * m;
    if (var < 0.0) var = 0.0;
    const double istd = 1.0 / sqrt(var + (double)f.eps);
    fin_store(f.vec + c, (float)((double)f.gamma[c] * istd));          // scale, shift: read by the waiting workgroups
    fin_store(f.vec + C + c, (float)((double)f.beta[c] - m * (double)f.gamma[c] * istd));
    f.vec[2 * C + c] = (float)m;                                       // mean, invstd: for the backward's launches
    f.vec[3 * C + c] = (float)istd;
    if (f.running_mean != nullptr) {
      const double unbiased = f.count > 1.0 ? var * f.count / (f.count - 1.0) : var;
      f.running_mean[c] = (float)((1.0 - (double)f.momentum) * (double)f.running_mean[c] + (double)f.momentum * m);
      f.running_var[c] = (float)((1.0 - (double)f.momentum) * (double)f.running_var[c] + (double)f.momentum * unbiased);
    }
  }
}

// backward: what bn_bwd_finalize_kernel<EW> block `blk` does
template <int EW>
__device__ __forceinline__ void fin_backward(const BnFin& f, int C, int blk, int tid, double* sh) {
  constexpr int NG = 1024 / EW, LD = EW + 1;
  fin_row_sums<EW>(f.part, f.rows, C, blk, tid, sh);
  const int c = blk * EW + tid;
  if (tid < EW && c < C) {
    const double t1 = sh[(0 * NG) * LD + tid], t2 = sh[(1 * NG) * LD + tid];
    fin_store(f.sums + c, t1);
    fin_store(f.sums + C + c, t2);
    if (f.dgamma != nullptr) {
      f.dbeta[c] = (float)t1;
      f.dgamma[c] = (float)t2;
    }
  }
}

__global__ void bn_eval_scale_kernel(int C, const float* gamma, const float* beta, const float* rm,
                                     const float* rv, float eps, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    const float istd = 1.0f / sqrtf(rv[c] + eps);
    const float sc = gamma[c] * istd;
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
  }
}

// ------------------------------------------------------------------------------------------
// BN apply + ReLU (+ pool).  One thread = one pixel (POOL=false) or one 2x2 window (POOL=true)
// x VEC channels; consecutive threads walk the channel chunks of one pixel/window first, so each
// wave-instruction reads whole contiguous pixel rows.
// ------------------------------------------------------------------------------------------
template <typename T, bool POOL, bool FIN = false>
__global__ __launch_bounds__(256) void bn_relu_apply_kernel(
    const T* __restrict__ y, int ldy, const float* scale, const float* shift,
    int N, int H, int W, int C, T* __restrict__ act, int lda, T* __restrict__ pooled, int ldp,
    const T* __restrict__ res, int ldr, int pool_ceil, const BnFin fin) {
  const float* fin_tab = nullptr;
  if constexpr (FIN) {   // scale = fin.vec, shift = fin.vec + C: written by the first workgroups of THIS launch
    __shared__ double fin_sh[FIN_SH_DOUBLES];
    const int nfin = (C + fin.ew - 1) / fin.ew;
    if ((int)blockIdx.x < nfin) {
      if (fin.ew == 8) fin_forward<8>(fin, C, blockIdx.x, threadIdx.x, fin_sh);
      else fin_forward<32>(fin, C, blockIdx.x, threadIdx.x, fin_sh);
      fin_publish(fin.flag, threadIdx.x);
    }
    fin_wait(fin.flag, nfin, threadIdx.x);
    // scale | shift (2 C floats, written a moment ago by other workgroups of this launch) by coherent loads into LDS once;
    // the loop below reads them there (coherent loads in the loop: 133 us instead of 43 at 64 channels x 1 M pixels)
    float* tab = reinterpret_cast<float*>(fin_sh);
    for (int i = threadIdx.x; i < 2 * C; i += 256) tab[i] = fin_load(fin.vec + i);
    __syncthreads();
    fin_tab = tab;
  }
  // res != nullptr: act = relu(bn(y)) + res (the RSU residual, u2net.py:74), pooled = maxpool(act).
  // Windows are enumerated on the ceil grid so that every pixel is visited once; a window clipped by an odd
  // border is pooled only in ceil mode (MaxPool2d(2, 2, ceil_mode=True), u2net.py:30) and dropped in floor mode.
  constexpr int VEC = ElemTraits<T>::VEC;
  const int CC = C / VEC;
  const bool relu = !(pool_ceil & 2);   // bit 1 of the flag word: BatchNorm WITHOUT the ReLU (resunet's skip branch)
  const bool rev = (pool_ceil & 4) != 0;   // bit 2: walk the tensor from its END (the part its producer wrote last)
  pool_ceil &= 1;
  const int Ho = (H + 1) >> 1, Wo = (W + 1) >> 1;
  const int Hp = pool_ceil ? Ho : H >> 1, Wp = pool_ceil ? Wo : W >> 1;
  const long long total = POOL ? (long long)N * Ho * Wo * CC : (long long)N * H * W * CC;
  for (long long idx0 = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx0 < total;
       idx0 += (long long)gridDim.x * blockDim.x) {
    const long long idx = rev ? total - 1 - idx0 : idx0;
    const int cc = (int)(idx % CC);
    const long long u = idx / CC;
    const int c0 = cc * VEC;
    float sc[VEC], sh[VEC];
    if constexpr (FIN) {
#pragma unroll
      for (int i = 0; i < VEC; i += 4) {
        *reinterpret_cast<float4*>(sc + i) = *reinterpret_cast<const float4*>(fin_tab + c0 + i);
        *reinterpret_cast<float4*>(sh + i) = *reinterpret_cast<const float4*>(fin_tab + C + c0 + i);
      }
    } else {
#pragma unroll
      for (int i = 0; i < VEC; i += 4) {
        *reinterpret_cast<float4*>(sc + i) = *reinterpret_cast<const float4*>(scale + c0 + i);
        *reinterpret_cast<float4*>(sh + i) = *reinterpret_cast<const float4*>(shift + c0 + i);
      }
    }
    if constexpr (!POOL) {
      float v[VEC];
      load_f(y + (size_t)u * ldy + c0, v);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        v[i] = fmaf(v[i], sc[i], sh[i]);
        if (relu) v[i] = fmaxf(v[i], 0.f);
      }
      if (res != nullptr) {
        float r[VEC];
        load_f(res + (size_t)u * ldr + c0, r);
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] += r[i];
      }
      store_f(act + (size_t)u * lda + c0, v);
    } else {
      const int wo = (int)(u % Wo);
      const long long t = u / Wo;
      const int ho = (int)(t % Ho);
      const int img = (int)(t / Ho);
      const size_t p00 = ((size_t)img * H + 2 * ho) * W + 2 * wo;
      float m[VEC];
      // the four taps' loads are issued together, UNCONDITIONALLY (a clipped tap reads tap 0's address and is dropped by a
      // select): a load inside `if (tap inside)` is followed by s_waitcnt vmcnt(0) at the block's end (hipcc 7.2), which ran
      // the taps one memory round trip after the other
      bool ok[4];
      size_t pp[4];
      Vec16<T> vr[4], rr[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        ok[k] = 2 * ho + (k >> 1) < H && 2 * wo + (k & 1) < W;   // clipped window (tap 0 is always inside)
        pp[k] = p00 + (k >> 1) * W + (k & 1);
        // the dummy address is PIXEL 0 (of the tensor), not tap 0: a dummy the compiler can prove equal to an address it
        // has already loaded is turned back into `if (differs) load` (with its wait); the same for a missing residual -- a
        // load under `if (res != nullptr)`, uniform as that is, gets the wait too
        vr[k] = ld16(ok[k] ? y + pp[k] * ldy + c0 : y + c0);
        rr[k] = ld16(res != nullptr && ok[k] ? res + pp[k] * ldr + c0 : y + c0);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float v[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          v[i] = fmaf((float)vr[k].v[i], sc[i], sh[i]);
          if (relu) v[i] = fmaxf(v[i], 0.f);
          if (res != nullptr) v[i] = (float)(T)(v[i] + (float)rr[k].v[i]);  // the pool sees the stored value
          m[i] = (k == 0) ? v[i] : (ok[k] ? fmaxf(m[i], v[i]) : m[i]);
        }
        if (ok[k]) store_f(act + pp[k] * lda + c0, v);
      }
      if (ho < Hp && wo < Wp) store_f(pooled + (((size_t)img * Hp + ho) * Wp + wo) * ldp + c0, m);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Backward of BN(train) + ReLU (+ pool): shared per-thread work.
// For its pixel (or 2x2 window) and VEC channels the thread forms dz = g * [act > 0] and xhat.
// PASS 1 accumulates S0 += dz, S1 += dz*xhat; PASS 2 writes dy = scale*(dz - S0/N - xhat*S1/N).
// The pool gradient goes to the FIRST maximum of the window in (0,0),(0,1),(1,0),(1,1) order,
// which is the element ATen's max_pool2d records (strict '>' scan).
// ------------------------------------------------------------------------------------------
struct BnBwdArgs {
  const void* y;
  const void* g0;
  const void* g1;
  const void* gp;
  void* dy;
  const float* scale;
  const float* shift;
  const float* mean;
  const float* invstd;
  double* sums;        // pass 2 in: [2][C] totals
  float* partials;     // pass 1 out: [gridDim.x][2][C]
  float* dgamma;
  float* dbeta;
  double inv_count;
  int N, H, W, C, ldy, ldg0, ldg1, ldgp, lddy, pool_ceil;
  BnFin fin;           // FIN launches of pass 2: the totals are formed by the first workgroups of the launch itself
};

template <typename T, bool POOL, int PASS, bool FIN = false>
__global__ __launch_bounds__(256) void bn_relu_bwd_kernel(const BnBwdArgs a) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const double* fin_tab = nullptr;
  if constexpr (FIN) {
    static_assert(PASS == 2, "the fused finalize belongs to the apply pass");
    __shared__ double fin_sh[FIN_SH_DOUBLES];
    const int nfin = (a.C + a.fin.ew - 1) / a.fin.ew;
    const int wg = blockIdx.y * gridDim.x + blockIdx.x, tid = threadIdx.y * blockDim.x + threadIdx.x;
    if (wg < nfin) {
      if (a.fin.ew == 8) fin_backward<8>(a.fin, a.C, wg, tid, fin_sh);
      else fin_backward<32>(a.fin, a.C, wg, tid, fin_sh);
      fin_publish(a.fin.flag, tid);
    }
    fin_wait(a.fin.flag, nfin, tid);
    // the totals of this workgroup's channel slice by coherent loads into LDS once (every thread loading its own 16 doubles
    // that way was 67 M loads past the caches per launch)
    const int cw = blockDim.x * VEC, cb = blockIdx.y * cw;
    for (int i = tid; i < 2 * cw; i += 256) {
      const int which = i / cw, j = i - which * cw;
      fin_sh[i] = cb + j < a.C ? fin_load(a.sums + (size_t)which * a.C + cb + j) : 0.0;
    }
    __syncthreads();
    fin_tab = fin_sh;
  }
  // a 2x2 pooling window, or (no pool) four pixels one grid stride apart: either way a thread has its 8-12
  // sixteen-byte loads of an iteration in flight together (one pixel per iteration ran the reduce pass at 3.2 TB/s)
  constexpr int NPIX = 4;
  extern __shared__ __attribute__((aligned(16))) float red[];  // PASS 1: [blockDim.y][blockDim.x][2*VEC]
  const T* __restrict__ y = static_cast<const T*>(a.y);
  const T* __restrict__ g0 = static_cast<const T*>(a.g0);
  const T* __restrict__ g1 = static_cast<const T*>(a.g1);
  const T* __restrict__ gp = static_cast<const T*>(a.gp);
  T* __restrict__ dy = static_cast<T*>(a.dy);
  const int CC = a.C / VEC;
  const int cc = blockIdx.y * blockDim.x + threadIdx.x;  // channel chunk of this thread
  const bool cok = cc < CC;
  const int c0 = (cok ? cc : 0) * VEC;
  const int Ho = (a.H + 1) >> 1, Wo = (a.W + 1) >> 1;   // window grid (ceil): every pixel in exactly one window
  const bool norelu = (a.pool_ceil & 2) != 0;           // bit 1: the forward had no ReLU (non-pool launches only)
  const int Hp = (a.pool_ceil & 1) ? Ho : a.H >> 1, Wp = (a.pool_ceil & 1) ? Wo : a.W >> 1;
  const long long units = POOL ? (long long)a.N * Ho * Wo : (long long)a.N * a.H * a.W;

  float sc[VEC], sh[VEC], mu[VEC], is[VEC], k0[VEC], k1[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    sc[i] = a.scale[c0 + i];
    sh[i] = a.shift[c0 + i];
    mu[i] = a.mean[c0 + i];
    is[i] = a.invstd[c0 + i];
    if (PASS == 2) {
      if constexpr (FIN) {
        k0[i] = (float)(fin_tab[threadIdx.x * VEC + i] * a.inv_count);
        k1[i] = (float)(fin_tab[blockDim.x * VEC + threadIdx.x * VEC + i] * a.inv_count);
      } else {
        k0[i] = (float)(a.sums[c0 + i] * a.inv_count);
        k1[i] = (float)(a.sums[a.C + c0 + i] * a.inv_count);
      }
    }
  }
  float S0[VEC], S1[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) S0[i] = S1[i] = 0.f;

  const long long ustride = (long long)gridDim.x * blockDim.y;
  // bit 2 of the flag word: walk the units from the END -- the part of y / g the producing kernel wrote last and the
  // Infinity Cache still holds (sums are per workgroup and fixed-order either way; the walk changes WHICH units a
  // workgroup sums, so the partial rows -- not the totals' determinism -- differ between the two walks)
  const bool rev = (a.pool_ceil & 4) != 0;
  for (long long u0 = (long long)blockIdx.x * blockDim.y + threadIdx.y; u0 < units && cok;
       u0 += (POOL ? 1 : NPIX) * ustride) {
    size_t pix[NPIX];   // pixel of slot k; slots outside the map / past the end read a clamped pixel and are zeroed
    bool in[NPIX];
    size_t gpoff = 0;
    bool has_pool = false;
    if constexpr (POOL) {
      const long long u = rev ? units - 1 - u0 : u0;
      const int wo = (int)(u % Wo);
      const long long t = u / Wo;
      const int ho = (int)(t % Ho);
      const int img = (int)(t / Ho);
#pragma unroll
      for (int k = 0; k < NPIX; ++k) {
        const int hh = 2 * ho + (k >> 1), ww = 2 * wo + (k & 1);
        in[k] = hh < a.H && ww < a.W;
        pix[k] = ((size_t)img * a.H + min(hh, a.H - 1)) * a.W + min(ww, a.W - 1);
      }
      has_pool = ho < Hp && wo < Wp;
      gpoff = (((size_t)img * Hp + ho) * Wp + wo) * a.ldgp;
    } else {
#pragma unroll
      for (int k = 0; k < NPIX; ++k) {
        const long long q = u0 + k * ustride;
        in[k] = q < units;
        pix[k] = (size_t)(in[k] ? (rev ? units - 1 - q : q) : units - 1);
      }
    }
#ifndef UZ_BN_BRANCHY
    // every load of the iteration is issued before the first use: a branch per slot put an s_waitcnt vmcnt(0)
    // behind each slot's loads and the four slots ran one memory latency after the other
    Vec16<T> yr[NPIX], g0r[NPIX], g1r[NPIX], gpr;
#pragma unroll
    for (int k = 0; k < NPIX; ++k) {
      yr[k] = ld16(y + pix[k] * a.ldy + c0);
      if constexpr (POOL) {
        // (the pooled loop is not unswitched on the two pointers, and a load under `if (g != nullptr)`, uniform as that is,
        // is waited for at the block's end: a missing gradient reads pixel 0 of y instead -- a cached line -- and is dropped)
        g0r[k] = ld16(g0 != nullptr ? g0 + pix[k] * a.ldg0 + c0 : y + c0);
        g1r[k] = ld16(g1 != nullptr ? g1 + pix[k] * a.ldg1 + c0 : y + c0);
      } else {
        if (g0 != nullptr) g0r[k] = ld16(g0 + pix[k] * a.ldg0 + c0);
        if (g1 != nullptr) g1r[k] = ld16(g1 + pix[k] * a.ldg1 + c0);
      }
    }
    // ... the pool's gradient among them (a window without a pooled pixel reads element 0 and does not use it): as a load
    // behind `if (has_pool)` it was a second memory round trip per window, after the first had been waited for
    if constexpr (POOL) gpr = ld16(gp != nullptr ? gp + (has_pool ? gpoff : 0) + c0 : y + c0);
    float yv[NPIX][VEC], gv[NPIX][VEC];
#pragma unroll
    for (int k = 0; k < NPIX; ++k)
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        yv[k][i] = in[k] ? (float)yr[k].v[i] : 0.f;
        float gsum = g0 != nullptr ? (float)g0r[k].v[i] : 0.f;
        if (g1 != nullptr) gsum += (float)g1r[k].v[i];
        gv[k][i] = in[k] ? gsum : 0.f;
      }
#else
    float yv[NPIX][VEC], gv[NPIX][VEC];
#pragma unroll
    for (int k = 0; k < NPIX; ++k) {
      const size_t p = pix[k];
      if (!in[k]) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) yv[k][i] = gv[k][i] = 0.f;
        continue;
      }
      load_f(y + p * a.ldy + c0, yv[k]);
      if (g0 != nullptr) {
        load_f(g0 + p * a.ldg0 + c0, gv[k]);
      } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) gv[k][i] = 0.f;
      }
      if (g1 != nullptr) {
        float t[VEC];
        load_f(g1 + p * a.ldg1 + c0, t);
#pragma unroll
        for (int i = 0; i < VEC; ++i) gv[k][i] += t[i];
      }
    }
#endif
    if constexpr (POOL) {
      if (gp != nullptr && has_pool) {
        float gpv[VEC];
#ifndef UZ_BN_BRANCHY
#pragma unroll
        for (int i = 0; i < VEC; ++i) gpv[i] = (float)gpr.v[i];
#else
        load_f(gp + gpoff + c0, gpv);
#endif
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          float best = fmaf(yv[0][i], sc[i], sh[i]);
          if (!norelu) best = fmaxf(best, 0.f);
          int bk = 0;
#pragma unroll
          for (int k = 1; k < 4; ++k) {
            float v = fmaf(yv[k][i], sc[i], sh[i]);
            if (!norelu) v = fmaxf(v, 0.f);
            if (in[k] && v > best) {
              best = v;
              bk = k;
            }
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) gv[k][i] += (k == bk) ? gpv[i] : 0.f;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NPIX; ++k) {
      if (!in[k]) continue;
      float out[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float pre = fmaf(yv[k][i], sc[i], sh[i]);
        const float dz = (norelu || pre > 0.f) ? gv[k][i] : 0.f;
        const float xh = (yv[k][i] - mu[i]) * is[i];
        if (PASS == 1) {
          S0[i] += dz;
          S1[i] += dz * xh;
        } else {
          out[i] = sc[i] * (dz - k0[i] - xh * k1[i]);
        }
      }
      if (PASS == 2) {
        store_f(dy + pix[k] * a.lddy + c0, out);
      }
    }
  }

  if (PASS == 1) {
    // reduce over threadIdx.y through LDS, then one double atomic per channel and block
    float* mine = red + ((size_t)threadIdx.y * blockDim.x + threadIdx.x) * (2 * VEC);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      mine[i] = S0[i];
      mine[VEC + i] = S1[i];
    }
    __syncthreads();
    // thread (x, y) finalises element e = y, y + blockDim.y, ... of chunk x
    for (int e = threadIdx.y; e < 2 * VEC; e += blockDim.y) {
      float t = 0.f;
      for (int r = 0; r < (int)blockDim.y; ++r) t += red[((size_t)r * blockDim.x + threadIdx.x) * (2 * VEC) + e];
      if (cok) {
        const int which = e / VEC, ch = c0 + (e % VEC);
        a.partials[((size_t)blockIdx.x * 2 + which) * a.C + ch] = t;  // one deterministic row per block
      }
    }
  }
}

// totals[2][C] (double) = sum over the partial rows; dbeta = totals[0], dgamma = totals[1]
template <int EW>
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ part, int rows,
                                                               int C, double* __restrict__ sums,
                                                               float* dgamma, float* dbeta) {
  constexpr int NG = 1024 / EW;
  __shared__ double sh[2][NG][EW + 1];
  const int cl = threadIdx.x % EW, g = threadIdx.x / EW;
  const int c = blockIdx.x * EW + cl;
  double a1 = 0.0, a2 = 0.0;
  const int cq = c < C ? c : 0;
  // four row groups per trip, unconditional loads, same order of additions (see bn_finalize_kernel)
  for (int r = g; r < rows; r += 4 * NG) {
    float p1[4], p2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int rr = r + k * NG < rows ? r + k * NG : 0;
      p1[k] = part[((size_t)rr * 2 + 0) * C + cq];
      p2[k] = part[((size_t)rr * 2 + 1) * C + cq];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(p1[k]), "+v"(p2[k]));
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (r + k * NG < rows && c < C) {
        a1 += (double)p1[k];
        a2 += (double)p2[k];
      }
  }
  sh[0][g][cl] = a1;
  sh[1][g][cl] = a2;
  __syncthreads();
  // fixed-order pairwise tree over the NG row groups (the pairing does not depend on timing: deterministic); a single
  // thread walking the NG partial sums cost 4 of this kernel's 5 microseconds
#pragma unroll
  for (int st = NG / 2; st > 0; st >>= 1) {
    if (g < st) {
      sh[0][g][cl] += sh[0][g + st][cl];
      sh[1][g][cl] += sh[1][g + st][cl];
    }
    __syncthreads();
  }
  if (g == 0 && c < C) {
    const double t1 = sh[0][0][cl], t2 = sh[1][0][cl];
    sums[c] = t1;
    sums[C + c] = t2;
    if (dgamma != nullptr) {
      dbeta[c] = (float)t1;
      dgamma[c] = (float)t2;
    }
  }
}

// ------------------------------------------------------------------------------------------
// OutConv: 1x1 convolution to Kout <= 8 logits, NCHW fp32 output.  LPP = C/VEC lanes cooperate
// on one pixel (one 16-byte load each) and combine with xor-shuffles.
// ------------------------------------------------------------------------------------------
constexpr int OUTCONV_MAXK = 8;
__host__ __device__ inline int lanes_per_pixel(int chunks) {  // power of two >= chunks
  int l = 1;
  while (l < chunks) l <<= 1;
  return l;
}

// XF: x holds the RAW output of the convolution in front; the head reads it through that layer's BatchNorm + ReLU,
// a = T(max(fma(x, scale, shift), 0)) -- the value uz_bn_relu_apply would have stored -- so that the normalised tensor of
// the last decoder block never exists (uz_outconv_fwd_xf)
template <typename T, int KOUT, bool XF = false>
__global__ __launch_bounds__(256) void outconv_fwd_kernel(const T* __restrict__ x, int ldx, int N,
                                                          int HW, int C, const float* __restrict__ w,
                                                          const float* __restrict__ b,
                                                          float* __restrict__ out,
                                                          const float* __restrict__ xf_scale = nullptr,
                                                          const float* __restrict__ xf_shift = nullptr) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int LPP = lanes_per_pixel(C / VEC);  // power of two >= C/VEC, <= 64; lanes >= C/VEC carry zeros
  const int ppb = blockDim.x / LPP;
  const int sub = threadIdx.x % LPP, pl = threadIdx.x / LPP;
  const bool live = sub < C / VEC;
  const int P = N * HW;  // < 2^31 (checked on the host)
  float wr[KOUT][VEC];
#pragma unroll
  for (int k = 0; k < KOUT; ++k)
#pragma unroll
    for (int i = 0; i < VEC; ++i) wr[k][i] = live ? w[k * C + sub * VEC + i] : 0.f;
  float xsc[VEC], xsh[VEC];
  if constexpr (XF) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      xsc[i] = live ? xf_scale[sub * VEC + i] : 0.f;
      xsh[i] = live ? xf_shift[sub * VEC + i] : 0.f;
    }
  }
  for (int p0 = blockIdx.x * ppb; p0 < P; p0 += gridDim.x * ppb) {
    const int p = p0 + pl;
    float v[VEC];
    if (p < P && live) {
      load_f(x + (size_t)p * ldx + sub * VEC, v);
      if constexpr (XF) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] = (float)(T)fmaxf(fmaf(v[i], xsc[i], xsh[i]), 0.f);
      }
    } else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) v[i] = 0.f;
    }
    const int img = (KOUT == 1) ? 0 : p / HW;
    const int hw = p - img * HW;
#pragma unroll
    for (int k = 0; k < KOUT; ++k) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) s = fmaf(v[i], wr[k][i], s);
      for (int o = LPP >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o);
      if (sub == 0 && p < P) out[((size_t)img * KOUT + k) * HW + hw] = s + b[k];
    }
  }
}

// BNRED: x = relu(bn(bn_y)) has no other reader: the two sums of that BatchNorm's backward over this workgroup's pixels
// (of dz = dx * [scale * bn_y + shift > 0], dx as stored) go to bnpart[blockIdx.x][2][C] (uz_outconv_bwd_bnred)
// XNULL (with BNRED): x was never written down (uz_outconv_fwd_xf): the activation is formed from bn_y
template <typename T, int KOUT, bool BNRED = false, bool XNULL = false>
__global__ __launch_bounds__(256) void outconv_bwd_kernel(const T* __restrict__ x, int ldx, int N,
                                                          int HW, int C, const float* __restrict__ w,
                                                          const float* __restrict__ g,
                                                          T* __restrict__ dx, int lddx,
                                                          float* __restrict__ partial, const T* __restrict__ bn_y = nullptr, int ld_bny = 0,
                                                          const float* __restrict__ bn_scale = nullptr, const float* __restrict__ bn_shift = nullptr,
                                                          const float* __restrict__ bn_mean = nullptr, const float* __restrict__ bn_invstd = nullptr,
                                                          float* __restrict__ bnpart = nullptr) {
  // partial[blockIdx.x][k][C + 1]: per-workgroup sums of g*x (C values) and g (1 value); a second
  // kernel adds the rows (float atomics onto the 65 hot addresses serialised: 425 us -> this form)
  constexpr int VEC = ElemTraits<T>::VEC;
  constexpr int UNR = 4;
  __shared__ float red[256 * 9];
  const int LPP = lanes_per_pixel(C / VEC);
  const int ppb = blockDim.x / LPP;
  const int sub = threadIdx.x % LPP, pl = threadIdx.x / LPP;
  const bool live = sub < C / VEC;
  const int P = N * HW;  // < 2^31 (checked on the host)
  float wr[KOUT][VEC], aw[KOUT][VEC], ab[KOUT];
#pragma unroll
  for (int k = 0; k < KOUT; ++k) {
    ab[k] = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      wr[k][i] = live ? w[k * C + sub * VEC + i] : 0.f;
      aw[k][i] = 0.f;
    }
  }
  float bsc[VEC], bsh[VEC], bmu[VEC], bis[VEC], S0[VEC], S1[VEC];
  if constexpr (BNRED) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const int ch = live ? sub * VEC + i : 0;
      bsc[i] = bn_scale[ch];
      bsh[i] = bn_shift[ch];
      bmu[i] = bn_mean[ch];
      bis[i] = bn_invstd[ch];
      S0[i] = S1[i] = 0.f;
    }
  }
  for (int p0 = blockIdx.x * ppb * UNR; p0 < P; p0 += gridDim.x * ppb * UNR) {
    float v[UNR][VEC], gk[UNR][KOUT], yv[UNR][VEC];
    int pp[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {   // all loads first: UNR pixels in flight.  UNCONDITIONAL loads (a lane out of range
      // reads pixel 0 / chunk 0 and is zeroed by a select): a load inside `if (in range)` ends in s_waitcnt vmcnt(0) at the
      // block's end (hipcc 7.2) and the UNR pixels came one memory round trip after the other
      const int p = p0 + u * ppb + pl;
      const bool inp = p < P, ok = inp && live;
      pp[u] = inp ? p : -1;
      const size_t pc = ok ? (size_t)p : 0;
      const int cs = ok ? sub * VEC : 0;
      Vec16<T> xr, yr;
      if constexpr (BNRED) yr = ld16(bn_y + pc * ld_bny + cs);
      if constexpr (!XNULL) xr = ld16(x + pc * ldx + cs);
      const int pg = inp ? p : 0;
      const int img = (KOUT == 1) ? 0 : pg / HW;  // KOUT == 1: NCHW index == pixel index
      const int hw = pg - img * HW;
#pragma unroll
      for (int k = 0; k < KOUT; ++k) {
        const float gq = g[((size_t)img * KOUT + k) * HW + hw];
        gk[u][k] = inp ? gq : 0.f;
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        if constexpr (BNRED) {
          yv[u][i] = (float)yr.v[i];
          // XNULL: the activation was never written down (uz_outconv_fwd_xf): the value the apply pass would have stored
          float xv;
          if constexpr (XNULL) xv = (float)(T)fmaxf(fmaf(yv[u][i], bsc[i], bsh[i]), 0.f);
          else xv = (float)xr.v[i];
          v[u][i] = ok ? xv : 0.f;
        } else {
          v[u][i] = ok ? (float)xr.v[i] : 0.f;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      float d[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) d[i] = 0.f;
#pragma unroll
      for (int k = 0; k < KOUT; ++k) {
        ab[k] += gk[u][k];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          d[i] = fmaf(gk[u][k], wr[k][i], d[i]);
          aw[k][i] = fmaf(gk[u][k], v[u][i], aw[k][i]);
        }
      }
      if (dx != nullptr && pp[u] >= 0 && live) store_f(dx + (size_t)pp[u] * lddx + sub * VEC, d);
      if constexpr (BNRED) {
        if (pp[u] >= 0 && live) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            const float dz = fmaf(yv[u][i], bsc[i], bsh[i]) > 0.f ? (float)(T)d[i] : 0.f;   // the gradient as stored
            S0[i] += dz;
            S1[i] += dz * ((yv[u][i] - bmu[i]) * bis[i]);
          }
        }
      }
    }
  }
  if constexpr (BNRED) {   // block reduction over the pixel lanes, one partial row per workgroup (deterministic)
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < VEC; ++i) red[threadIdx.x * 9 + i] = which ? S1[i] : S0[i];
      __syncthreads();
      if (pl == 0 && live) {
        float t[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) t[i] = 0.f;
        for (int r = 0; r < ppb; ++r)
#pragma unroll
          for (int i = 0; i < VEC; ++i) t[i] += red[(r * LPP + sub) * 9 + i];
#pragma unroll
        for (int i = 0; i < VEC; ++i) bnpart[((size_t)blockIdx.x * 2 + which) * C + sub * VEC + i] = t[i];
      }
    }
  }
  // block reduction over the ppb pixel lanes that share `sub`
#pragma unroll
  for (int k = 0; k < KOUT; ++k) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < VEC; ++i) red[threadIdx.x * 9 + i] = aw[k][i];
    red[threadIdx.x * 9 + 8] = ab[k];
    __syncthreads();
    if (pl == 0) {
      float t[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) t[i] = 0.f;
      for (int r = 0; r < ppb; ++r)
#pragma unroll
        for (int i = 0; i < 9; ++i)
          if (i < VEC || i == 8) t[i] += red[(r * LPP + sub) * 9 + i];
      float* row = partial + ((size_t)blockIdx.x * KOUT + k) * (C + 1);
      if (live) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) row[sub * VEC + i] = t[i];
      }
      if (sub == 0) row[C] = t[8];
    }
  }
}

// dw[k][c] = sum_rows partial[row][k][c]; db[k] = sum_rows partial[row][k][C]
// 1024 threads per 32 elements: thread (e = t & 31, g = t >> 5) adds rows g, g+32, ...
__global__ __launch_bounds__(1024) void outconv_bwd_finalize_kernel(const float* __restrict__ partial, int rows,
                                                                    int Kout, int C, float* __restrict__ dw,
                                                                    float* __restrict__ db) {
  __shared__ double sh[32][33];
  const int el = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int ne = Kout * (C + 1);
  const int e = blockIdx.x * 32 + el;
  double s = 0.0;
  {   // eight rows per trip, unconditional loads (row 0 / element 0 out of range, dropped), same order of additions
    const bool in = e < ne;
    const float* base = partial + (in ? e : 0);
    for (int r = g; r < rows; r += 8 * 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = base[(size_t)(r + u * 32 < rows ? r + u * 32 : 0) * ne];
#pragma unroll
      for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(v[u]));
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (in && r + u * 32 < rows) s += (double)v[u];
    }
  }
  sh[g][el] = s;
  __syncthreads();
#pragma unroll
  for (int st = 16; st > 0; st >>= 1) {   // fixed-order pairwise tree
    if (g < st) sh[g][el] += sh[g + st][el];
    __syncthreads();
  }
  if (g == 0 && e < ne) {
    const double t = sh[0][el];
    const int k = e / (C + 1), c = e - k * (C + 1);
    if (c < C) dw[k * C + c] = (float)t;
    else db[k] = (float)t;
  }
}

#define UZ_KOUT_SWITCH(K, ...)                   \
  switch (K) {                                   \
    case 1: { constexpr int KOUT = 1; __VA_ARGS__; break; } \
    case 2: { constexpr int KOUT = 2; __VA_ARGS__; break; } \
    case 3: { constexpr int KOUT = 3; __VA_ARGS__; break; } \
    case 4: { constexpr int KOUT = 4; __VA_ARGS__; break; } \
    case 5: { constexpr int KOUT = 5; __VA_ARGS__; break; } \
    case 6: { constexpr int KOUT = 6; __VA_ARGS__; break; } \
    case 7: { constexpr int KOUT = 7; __VA_ARGS__; break; } \
    default: { constexpr int KOUT = 8; __VA_ARGS__; break; } \
  }

// out[c] += sum_p x[p*ld + c]
// PARTIAL: one row of per-workgroup sums (out[blockIdx.x][C], finished by colsum_finalize_kernel:
// deterministic, no contended atomics); otherwise atomicAdd onto out[C] (legacy uz_colsum).
template <typename T, bool PARTIAL>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int ld, long long P, int C,
                                                     float* __restrict__ out) {
  constexpr int VEC = ElemTraits<T>::VEC;
  extern __shared__ __attribute__((aligned(16))) float red[];  // [blockDim.y][blockDim.x][VEC]
  const int CC = C / VEC;
  const int cc = blockIdx.y * blockDim.x + threadIdx.x;
  const bool cok = cc < CC;
  const int c0 = (cok ? cc : 0) * VEC;
  float s[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) s[i] = 0.f;
  for (long long p = (long long)blockIdx.x * blockDim.y + threadIdx.y; p < P && cok;
       p += (long long)gridDim.x * blockDim.y) {
    float v[VEC];
    load_f(x + (size_t)p * ld + c0, v);
#pragma unroll
    for (int i = 0; i < VEC; ++i) s[i] += v[i];
  }
  float* mine = red + ((size_t)threadIdx.y * blockDim.x + threadIdx.x) * VEC;
#pragma unroll
  for (int i = 0; i < VEC; ++i) mine[i] = s[i];
  __syncthreads();
  for (int e = threadIdx.y; e < VEC; e += blockDim.y) {
    float t = 0.f;
    for (int r = 0; r < (int)blockDim.y; ++r) t += red[((size_t)r * blockDim.x + threadIdx.x) * VEC + e];
    if (cok) {
      if (PARTIAL) out[(size_t)blockIdx.x * C + c0 + e] = t;
      else atomicAdd(out + c0 + e, t);
    }
  }
}

// per-channel sum and sum of squares of an arbitrary NHWC tensor, as partial rows [blockIdx.x][2][C] in the
// layout uz_bn_finalize() reads (what the convolution epilogues produce for their own outputs): the statistics
// of a BatchNorm whose input is not a convolution output (pre-activation blocks: ResidualConv, common_layers.py:186)
template <typename T>
__global__ __launch_bounds__(256) void colstats_kernel(const T* __restrict__ x, int ld, long long P, int C,
                                                       float* __restrict__ out) {
  constexpr int VEC = ElemTraits<T>::VEC;
  extern __shared__ __attribute__((aligned(16))) float red[];  // [blockDim.y][blockDim.x][2 * VEC]
  const int CC = C / VEC;
  const int cc = blockIdx.y * blockDim.x + threadIdx.x;
  const bool cok = cc < CC;
  const int c0 = (cok ? cc : 0) * VEC;
  float s[VEC], q[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) s[i] = q[i] = 0.f;
  const long long stride = (long long)gridDim.x * blockDim.y;
  for (long long p = (long long)blockIdx.x * blockDim.y + threadIdx.y; p < P && cok; p += 4 * stride) {
    float v[4][VEC];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (p + k * stride < P) {
        load_f(x + (size_t)(p + k * stride) * ld + c0, v[k]);
      } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[k][i] = 0.f;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        s[i] += v[k][i];
        q[i] = fmaf(v[k][i], v[k][i], q[i]);
      }
  }
  float* mine = red + ((size_t)threadIdx.y * blockDim.x + threadIdx.x) * (2 * VEC);
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    mine[i] = s[i];
    mine[VEC + i] = q[i];
  }
  __syncthreads();
  for (int e = threadIdx.y; e < 2 * VEC; e += blockDim.y) {
    float t = 0.f;
    for (int r = 0; r < (int)blockDim.y; ++r) t += red[((size_t)r * blockDim.x + threadIdx.x) * (2 * VEC) + e];
    if (cok) out[((size_t)blockIdx.x * 2 + e / VEC) * C + c0 + (e % VEC)] = t;
  }
}

__global__ __launch_bounds__(1024) void colsum_finalize_kernel(const float* __restrict__ partial, int rows, int C,
                                                               float* __restrict__ out) {
  __shared__ double sh[32][33];
  const int el = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + el;
  double s = 0.0;
  {   // eight rows per trip, unconditional loads, same order of additions
    const bool in = c < C;
    const float* base = partial + (in ? c : 0);
    for (int r = g; r < rows; r += 8 * 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = base[(size_t)(r + u * 32 < rows ? r + u * 32 : 0) * C];
#pragma unroll
      for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(v[u]));
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (in && r + u * 32 < rows) s += (double)v[u];
    }
  }
  sh[g][el] = s;
  __syncthreads();
  if (g == 0 && c < C) {
    double t = 0.0;
    for (int r = 0; r < 32; ++r) t += sh[r][el];
    out[c] = (float)t;
  }
}

// ------------------------------------------------------------------------------------------
// Weight packing and input im2col
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float pack_element(int mode, const float* __restrict__ w, int Co, int Ci,
                                              int Tn, int Kpad, long long idx) {
  switch (mode) {
    case UZ_PACK_CONV_FWD: {  // dst[Co][t*Ci+ci] <- w[co][ci][t]
      const int k = (int)(idx % ((long long)Tn * Ci));
      const int co = (int)(idx / ((long long)Tn * Ci));
      const int t = k / Ci, ci = k - t * Ci;
      return w[((size_t)co * Ci + ci) * Tn + t];
    }
    case UZ_PACK_CONV_DGRAD: {  // dst[Ci][(T-1-t)*Co+co] <- w[co][ci][t]
      const int k = (int)(idx % ((long long)Tn * Co));
      const int ci = (int)(idx / ((long long)Tn * Co));
      const int tf = k / Co, co = k - tf * Co;
      return w[((size_t)co * Ci + ci) * Tn + (Tn - 1 - tf)];
    }
    case UZ_PACK_CONVT_FWD: {  // dst[t*Co+co][ci] <- w[ci][co][t]
      const int ci = (int)(idx % Ci);
      const int r = (int)(idx / Ci);
      const int t = r / Co, co = r - t * Co;
      return w[((size_t)ci * Co + co) * Tn + t];
    }
    case UZ_PACK_CONVT_DGRAD: {  // dst[ci][t*Co+co] <- w[ci][co][t]
      const int k = (int)(idx % ((long long)Tn * Co));
      const int ci = (int)(idx / ((long long)Tn * Co));
      const int t = k / Co, co = k - t * Co;
      return w[((size_t)ci * Co + co) * Tn + t];
    }
    default: {  // UZ_PACK_IM2COL: dst[Co][Kpad], k = t*Ci+ci
      const int k = (int)(idx % Kpad);
      const int co = (int)(idx / Kpad);
      if (k < Tn * Ci) {
        const int t = k / Ci, ci = k - t * Ci;
        return w[((size_t)co * Ci + ci) * Tn + t];
      }
      return 0.f;
    }
  }
}

template <typename T>
__global__ void pack_weights_kernel(int mode, const float* __restrict__ w, int Co, int Ci, int Tn,
                                    int Kpad, T* __restrict__ dst, long long total) {
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x)
    dst[idx] = (T)pack_element(mode, w, Co, Ci, Tn, Kpad, idx);
}

// every weight tensor of a model in ONE launch: blockIdx.y = item, blockIdx.x strides over its elements.
// One-tap items (Linear layers, 1x1 convolutions: all of swin_unet_v2's 27 M parameters, twice per step) are
// plain 2-D matrices and take two fast paths instead of the element-wise gather with its 64-bit divisions:
// a vectorised convert-copy when source and destination have the same orientation, and a 64 x 64 transpose
// through LDS (coalesced both ways) when they do not.
template <typename T>
__global__ __launch_bounds__(256) void pack_weights_batched_kernel(const uz_pack_item* __restrict__ items, int n, long long total) {
  const uz_pack_item it = items[blockIdx.y];
  const long long count = ((int)blockIdx.y + 1 < n ? items[blockIdx.y + 1].begin : total) - it.begin;
  T* __restrict__ dst = static_cast<T*>(it.dst);
  const float* __restrict__ src = it.src;
  if (it.mode == UZ_PACK_VEC_REPEAT) {   // fp32 in, fp32 out: v[Co] T times over
    float* __restrict__ df = static_cast<float*>(it.dst);
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < count; idx += (long long)gridDim.x * blockDim.x)
      df[idx] = src[idx % it.Co];
    return;
  }
  const bool same = it.mode == UZ_PACK_CONV_FWD || it.mode == UZ_PACK_CONVT_DGRAD;
  const bool transposed = it.mode == UZ_PACK_CONV_DGRAD || it.mode == UZ_PACK_CONVT_FWD;
  if (it.T == 1 && same && count == (long long)it.Co * it.Ci && (count & 3) == 0 &&
      (reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 7) == 0) {
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < count / 4;
         q += (long long)gridDim.x * blockDim.x) {
      const float4 v = reinterpret_cast<const float4*>(src)[q];
      dst[4 * q] = (T)v.x;
      dst[4 * q + 1] = (T)v.y;
      dst[4 * q + 2] = (T)v.z;
      dst[4 * q + 3] = (T)v.w;
    }
    return;
  }
  if (it.T == 1 && transposed && count == (long long)it.Co * it.Ci) {
    // source rows R x columns Cc -> destination [Cc][R]
    const int R = it.mode == UZ_PACK_CONV_DGRAD ? it.Co : it.Ci;
    const int Cc = it.mode == UZ_PACK_CONV_DGRAD ? it.Ci : it.Co;
    __shared__ float tile[64][65];
    const int tr = (R + 63) / 64, tc = (Cc + 63) / 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int t = blockIdx.x; t < tr * tc; t += gridDim.x) {
      const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;
      __syncthreads();
      {   // the 16 loads of a thread unconditional and in flight together (element 0 outside the matrix, dropped by the select;
          // the values pass through an empty asm so that the select cannot pull the loads back under a branch): as
          // `inside ? src[...] : 0` each was waited for before the next was issued
        float tv[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const int r = ty + 4 * k;
          tv[k] = src[(r0 + r < R && c0 + tx < Cc) ? (size_t)(r0 + r) * Cc + c0 + tx : 0];
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(tv[k]));
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const int r = ty + 4 * k;
          tile[r][tx] = (r0 + r < R && c0 + tx < Cc) ? tv[k] : 0.f;
        }
      }
      __syncthreads();
      for (int c = ty; c < 64; c += 4)
        if (c0 + c < Cc && r0 + tx < R) dst[(size_t)(c0 + c) * R + r0 + tx] = (T)tile[tx][c];
    }
    return;
  }
  // ConvTranspose2d k2 s2 (T = 4 taps, w[ci][co][t]; unet's four up-convolutions, 2.8 M parameters, both layouts per step): the
  // source is a [Ci][4 Co] matrix with column j = 4 co + t.  The element-wise gather below (64-bit divisions per element) ran
  // them at 0.3 TB/s; these two paths are 32-bit and coalesced on the wide side.
  if (it.T == 4 && it.mode == UZ_PACK_CONVT_DGRAD && count == 4LL * it.Co * it.Ci) {
    // dst[ci][t Co + co]: a row is the source row with (co, t) -> (t, co)
    const int rowlen = 4 * it.Co;
    for (int ci = blockIdx.x; ci < it.Ci; ci += gridDim.x) {
      const float* __restrict__ srow = src + (size_t)ci * rowlen;
      T* __restrict__ drow = dst + (size_t)ci * rowlen;
      for (int k = threadIdx.x; k < rowlen; k += 256) {
        const int t = k / it.Co, co = k - t * it.Co;
        drow[k] = (T)srow[4 * co + t];
      }
    }
    return;
  }
  if (it.T == 4 && it.mode == UZ_PACK_CONVT_FWD && count == 4LL * it.Co * it.Ci) {
    // dst[t Co + co][ci]: the transpose of the source matrix with destination row perm(j) = (j & 3) Co + (j >> 2)
    const int R = it.Ci, Cc = 4 * it.Co;
    __shared__ float tile4[64][65];
    const int tr = (R + 63) / 64, tc = (Cc + 63) / 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int t = blockIdx.x; t < tr * tc; t += gridDim.x) {
      const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;
      __syncthreads();
      {   // (unconditional loads, as in the one-tap transpose above)
        float tv[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const int r = ty + 4 * k;
          tv[k] = src[(r0 + r < R && c0 + tx < Cc) ? (size_t)(r0 + r) * Cc + c0 + tx : 0];
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(tv[k]));
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const int r = ty + 4 * k;
          tile4[r][tx] = (r0 + r < R && c0 + tx < Cc) ? tv[k] : 0.f;
        }
      }
      __syncthreads();
      for (int c = ty; c < 64; c += 4) {
        const int j = c0 + c;
        if (j < Cc && r0 + tx < R) dst[(size_t)((j & 3) * it.Co + (j >> 2)) * R + r0 + tx] = (T)tile4[tx][c];
      }
    }
    return;
  }
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < count;
       idx += (long long)gridDim.x * blockDim.x)
    dst[idx] = (T)pack_element(it.mode, it.src, it.Co, it.Ci, it.T, it.Kpad, idx);
}

// generic: one thread per 16-byte destination chunk
template <typename T>
__global__ void im2col3x3_kernel(const float* __restrict__ x, int N, int C, int H, int W, int Kpad,
                                 T* __restrict__ dst, long long total) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int cpr = Kpad / VEC;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int ch = (int)(idx % cpr);
    const long long p = idx / cpr;
    const int w0 = (int)(p % W);
    const long long q = p / W;
    const int h0 = (int)(q % H);
    const int img = (int)(q / H);
    Vec16<T> v;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const int k = ch * VEC + e;
      float f = 0.f;
      if (k < 9 * C) {
        const int t = k / C, c = k - t * C;
        const int hh = h0 + t / 3 - 1, ww = w0 + t % 3 - 1;
        if ((unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W)
          f = x[(((size_t)img * C + c) * H + hh) * W + ww];
      }
      v.v[e] = (T)f;
    }
    st16(dst + (size_t)p * Kpad + ch * VEC, v);
  }
}

// small compile-time C (the RGB / grey network input): one thread per pixel.  For a fixed
// (tap, channel) consecutive lanes read consecutive w of the NCHW image (coalesced 4-byte loads).  The Kpad-wide
// rows are then exchanged through a wave-private LDS strip so that consecutive LANES store consecutive 16-byte pieces
// (a wave-store = one contiguous span; every lane writing its own 64-byte row in four 16-byte stores at a 64-byte lane
// stride ran at 1.1 TB/s).
template <typename T, int C>
__global__ __launch_bounds__(256) void im2col3x3_smallc_kernel(const float* __restrict__ x, int N, int H,
                                                               int W, int Kpad, T* __restrict__ dst,
                                                               long long npix) {
  constexpr int VEC = ElemTraits<T>::VEC;
  constexpr int K = 9 * C;
  constexpr int MAXCH = 64 / VEC;  // Kpad <= 64
  constexpr int ROWB = MAXCH * 16 + 16;   // LDS row pitch in bytes (+16: the 16-byte pieces of a row group spread over banks)
  __shared__ __attribute__((aligned(16))) char strip[4][64 * ROWB];
  const int nch = Kpad / VEC;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  char* mine = strip[wv];
  // whole waves walk the pixel range together (the exchange is wave-wide): p0 = first pixel of this wave's group of 64
  for (long long p0 = ((long long)blockIdx.x * blockDim.x + (threadIdx.x & ~63)); p0 < npix;
       p0 += (long long)gridDim.x * blockDim.x) {
    const long long p = p0 + lane;
    float row[K];
    if (p < npix) {
      const int w0 = (int)(p % W);
      const long long q = p / W;
      const int h0 = (int)(q % H);
      const int img = (int)(q / H);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int hh = h0 + t / 3 - 1, ww = w0 + t % 3 - 1;
        const bool ok = (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
#pragma unroll
        for (int c = 0; c < C; ++c) row[t * C + c] = ok ? x[(((size_t)img * C + c) * H + hh) * W + ww] : 0.f;
      }
    } else {
#pragma unroll
      for (int k = 0; k < K; ++k) row[k] = 0.f;
    }
#pragma unroll
    for (int ch = 0; ch < MAXCH; ++ch) {
      if (ch < nch) {
        Vec16<T> v;
#pragma unroll
        for (int e = 0; e < VEC; ++e) v.v[e] = (ch * VEC + e < K) ? (T)row[(ch * VEC + e < K) ? ch * VEC + e : 0] : (T)0.f;
        *reinterpret_cast<Vec16<T>*>(mine + lane * ROWB + ch * 16) = v;
      }
    }
    // (one wave: program order + the LDS queue's in-order completion make the writes visible to the reads below)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // piece j of the wave's 64 * nch sixteen-byte pieces: pixel j / nch, chunk j % nch -> global byte offset 16 j
    const int total = 64 * nch;
    for (int j = lane; j < total; j += 64) {
      const int px = j / nch, ch = j - px * nch;
      if (p0 + px < npix) {
        const Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(mine + px * ROWB + ch * 16);
        st16(dst + (size_t)(p0 + px) * Kpad + ch * VEC, v);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next group's writes
  }
}

int grid_for(long long total, int block) {
  long long g = (total + block - 1) / block;
  const long long cap = (long long)UZ_NUM_CU * 8;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}


// thread-block shape for the (channel-chunk x pixel) reductions
void reduce_shape(int CC, long long units, dim3* grid, dim3* block, int wg_per_cu = 2) {
  int bx = 1;
  while (bx < CC && bx < 64) bx <<= 1;
  const int by = 256 / bx;
  const int gy = (CC + bx - 1) / bx;
  long long gx = (units + by - 1) / by;
  // reductions: 2-4 workgroups per CU (every workgroup leaves one partial row for the finalize kernel);
  // the write-back pass of the BN backward has no rows to keep few and runs 8 per CU
  long long cap = (long long)UZ_NUM_CU * wg_per_cu / gy;
  if (cap < 1) cap = 1;
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  *grid = dim3((unsigned)gx, (unsigned)gy);
  *block = dim3(bx, by);
}

}  // namespace

extern "C" int uz_bn_finalize(const float* stats_partial, int grid_m, int C, double count,
                              const float* gamma, const float* beta, float eps, float momentum,
                              float* running_mean, float* running_var, float* scale, float* shift,
                              float* mean, float* invstd, void* stream) {
  UZ_REQUIRE(stats_partial && gamma && beta && scale && shift && mean && invstd, "uz_bn_finalize: null");
  UZ_REQUIRE(grid_m > 0 && C > 0 && count > 0, "uz_bn_finalize: bad shape");
  UZ_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "uz_bn_finalize: running stats");
  if (grid_m >= 128 && C <= 512)
    hipLaunchKernelGGL(bn_finalize_kernel<8>, dim3(uz_cdiv(C, 8)), dim3(1024), 0, (hipStream_t)stream,
                       stats_partial, grid_m, C, count, gamma, beta, eps, momentum, running_mean,
                       running_var, scale, shift, mean, invstd);
  else
    hipLaunchKernelGGL(bn_finalize_kernel<32>, dim3(uz_cdiv(C, 32)), dim3(1024), 0, (hipStream_t)stream,
                       stats_partial, grid_m, C, count, gamma, beta, eps, momentum, running_mean,
                       running_var, scale, shift, mean, invstd);
  UZ_LAUNCH_CHECK("uz_bn_finalize");
  return UZ_OK;
}

extern "C" int uz_bn_eval_scale(int C, const float* gamma, const float* beta, const float* running_mean,
                                const float* running_var, float eps, float* scale, float* shift,
                                void* stream) {
  UZ_REQUIRE(C > 0 && gamma && beta && running_mean && running_var && scale && shift, "uz_bn_eval_scale: null");
  hipLaunchKernelGGL(bn_eval_scale_kernel, dim3(uz_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, C,
                     gamma, beta, running_mean, running_var, eps, scale, shift);
  UZ_LAUNCH_CHECK("uz_bn_eval_scale");
  return UZ_OK;
}

template <typename T>
static int bn_relu_apply_t(const void* y, int ldy, const float* scale, const float* shift, int N, int H,
                           int W, int C, void* act, int lda, void* pooled, int ldp, const void* res, int ldr,
                           int pool_ceil, hipStream_t s) {
  constexpr int VEC = ElemTraits<T>::VEC;
  if (pooled != nullptr) {
    const long long total = (long long)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / VEC);
    hipLaunchKernelGGL((bn_relu_apply_kernel<T, true>), dim3(grid_for(total, 256)), dim3(256), 0, s,
                       (const T*)y, ldy, scale, shift, N, H, W, C, (T*)act, lda, (T*)pooled, ldp, (const T*)res, ldr, pool_ceil, BnFin{});
  } else {
    const long long total = (long long)N * H * W * (C / VEC);
    hipLaunchKernelGGL((bn_relu_apply_kernel<T, false>), dim3(grid_for(total, 256)), dim3(256), 0, s,
                       (const T*)y, ldy, scale, shift, N, H, W, C, (T*)act, lda, (T*)nullptr, 0, (const T*)res, ldr, pool_ceil & 6, BnFin{});
  }
  UZ_LAUNCH_CHECK("uz_bn_relu_apply");
  return UZ_OK;
}

extern "C" int uz_bn_relu_add_apply(int dtype, const void* y, int ldy, const float* scale, const float* shift,
                                    int N, int H, int W, int C, const void* res, int ldr, void* act, int lda,
                                    void* pooled, int ldp, int pool_ceil, void* stream);

extern "C" int uz_bn_relu_apply(int dtype, const void* y, int ldy, const float* scale, const float* shift,
                                int N, int H, int W, int C, void* act, int lda, void* pooled, int ldp,
                                void* stream) {
  return uz_bn_relu_add_apply(dtype, y, ldy, scale, shift, N, H, W, C, nullptr, 0, act, lda, pooled, ldp, 0, stream);
}

extern "C" int uz_bn_relu_add_apply(int dtype, const void* y, int ldy, const float* scale, const float* shift,
                                    int N, int H, int W, int C, const void* res, int ldr, void* act, int lda,
                                    void* pooled, int ldp, int pool_ceil, void* stream) {
  UZ_REQUIRE(dtype == UZ_F32 || dtype == UZ_BF16, "uz_bn_relu_apply: bad dtype");
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(y && scale && shift && act, "uz_bn_relu_apply: null pointer");
  UZ_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % vec == 0, "uz_bn_relu_apply: C=%d must be a multiple of %d", C, vec);
  UZ_REQUIRE(ldy % vec == 0 && lda % vec == 0 && ldy >= C && lda >= C, "uz_bn_relu_apply: bad ld");
  if (res != nullptr) UZ_REQUIRE(ldr % vec == 0 && ldr >= C, "uz_bn_relu_apply: bad ldr");
  if (pooled != nullptr) {
    UZ_REQUIRE((pool_ceil & 1) || (H >= 2 && W >= 2), "uz_bn_relu_apply: floor-mode pool of a %dx%d map is empty", H, W);
    UZ_REQUIRE(ldp % vec == 0 && ldp >= C, "uz_bn_relu_apply: bad ldp");
  }
  hipStream_t s = (hipStream_t)stream;
  return dtype == UZ_BF16 ? bn_relu_apply_t<bf16_t>(y, ldy, scale, shift, N, H, W, C, act, lda, pooled, ldp, res, ldr, pool_ceil, s)
                          : bn_relu_apply_t<float>(y, ldy, scale, shift, N, H, W, C, act, lda, pooled, ldp, res, ldr, pool_ceil, s);
}

static inline int fin_ew(int rows, int C) { return (rows >= 128 && C <= 512) ? 8 : 32; }   // as uz_bn_finalize chooses

template <typename T>
static int bn_relu_apply_fin_t(const void* y, int ldy, const BnFin& fin, int N, int H, int W, int C, void* act, int lda,
                               void* pooled, int ldp, const void* res, int ldr, int pool_ceil, hipStream_t s, bool* done) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const bool pool = pooled != nullptr;
  const long long total = (pool ? (long long)N * ((H + 1) / 2) * ((W + 1) / 2) : (long long)N * H * W) * (C / VEC);
  const int grid = grid_for(total, 256);
  *done = grid >= (C + fin.ew - 1) / fin.ew && 2 * C <= 2 * FIN_SH_DOUBLES;   // the table of the waiting workgroups: 2 C floats
  if (!*done) return UZ_OK;
  if (pool)
    hipLaunchKernelGGL((bn_relu_apply_kernel<T, true, true>), dim3(grid), dim3(256), 0, s, (const T*)y, ldy, fin.vec, fin.vec + C, N,
                       H, W, C, (T*)act, lda, (T*)pooled, ldp, (const T*)res, ldr, pool_ceil, fin);
  else
    hipLaunchKernelGGL((bn_relu_apply_kernel<T, false, true>), dim3(grid), dim3(256), 0, s, (const T*)y, ldy, fin.vec, fin.vec + C,
                       N, H, W, C, (T*)act, lda, (T*)nullptr, 0, (const T*)res, ldr, pool_ceil & 6, fin);
  UZ_LAUNCH_CHECK("uz_bn_relu_add_apply_fin");
  return UZ_OK;
}

extern "C" int uz_bn_relu_add_apply_fin(int dtype, const void* y, int ldy, const float* stats_partial, int rows, double count,
                                        const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                                        float* running_var, float* vec, int* flag, int N, int H, int W, int C, const void* res,
                                        int ldr, void* act, int lda, void* pooled, int ldp, int pool_ceil, void* stream) {
  UZ_REQUIRE(dtype == UZ_F32 || dtype == UZ_BF16, "uz_bn_relu_add_apply_fin: bad dtype");
  const int vecw = dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(y && act && stats_partial && gamma && beta && vec && flag, "uz_bn_relu_add_apply_fin: null pointer");
  UZ_REQUIRE(rows > 0 && count > 0, "uz_bn_relu_add_apply_fin: bad statistics shape");
  UZ_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "uz_bn_relu_add_apply_fin: running stats");
  UZ_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % vecw == 0, "uz_bn_relu_add_apply_fin: C=%d must be a multiple of %d", C, vecw);
  UZ_REQUIRE(ldy % vecw == 0 && lda % vecw == 0 && ldy >= C && lda >= C, "uz_bn_relu_add_apply_fin: bad ld");
  if (res != nullptr) UZ_REQUIRE(ldr % vecw == 0 && ldr >= C, "uz_bn_relu_add_apply_fin: bad ldr");
  if (pooled != nullptr) {
    UZ_REQUIRE((pool_ceil & 1) || (H >= 2 && W >= 2), "uz_bn_relu_add_apply_fin: floor-mode pool of a %dx%d map is empty", H, W);
    UZ_REQUIRE(ldp % vecw == 0 && ldp >= C, "uz_bn_relu_add_apply_fin: bad ldp");
  }
  BnFin fin{};
  fin.part = stats_partial; fin.rows = rows; fin.ew = fin_ew(rows, C); fin.count = count;
  fin.gamma = gamma; fin.beta = beta; fin.eps = eps; fin.momentum = momentum;
  fin.running_mean = running_mean; fin.running_var = running_var; fin.vec = vec; fin.flag = flag;
  hipStream_t s = (hipStream_t)stream;
  bool done = false;
  const int rc = dtype == UZ_BF16
      ? bn_relu_apply_fin_t<bf16_t>(y, ldy, fin, N, H, W, C, act, lda, pooled, ldp, res, ldr, pool_ceil, s, &done)
      : bn_relu_apply_fin_t<float>(y, ldy, fin, N, H, W, C, act, lda, pooled, ldp, res, ldr, pool_ceil, s, &done);
  if (rc != UZ_OK || done) return rc;
  // a grid smaller than the finalize (a map of a few pixels with many channels): the two launches
  const int rc2 = uz_bn_finalize(stats_partial, rows, C, count, gamma, beta, eps, momentum, running_mean, running_var, vec,
                                 vec + C, vec + 2 * C, vec + 3 * C, stream);
  if (rc2 != UZ_OK) return rc2;
  return uz_bn_relu_add_apply(dtype, y, ldy, vec, vec + C, N, H, W, C, res, ldr, act, lda, pooled, ldp, pool_ceil, stream);
}

static int bnbwd_check(const uz_bnbwd_desc* d, const void* g0, const void* g1, const void* gp) {
  UZ_REQUIRE(d != nullptr, "uz_bn_relu_bwd: null descriptor");
  UZ_REQUIRE(d->dtype == UZ_F32 || d->dtype == UZ_BF16, "uz_bn_relu_bwd: bad dtype");
  const int vec = d->dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->C % vec == 0, "uz_bn_relu_bwd: bad shape");
  UZ_REQUIRE(d->ldy % vec == 0 && d->ldy >= d->C, "uz_bn_relu_bwd: bad ldy");
  if (g0) UZ_REQUIRE(d->ldg0 % vec == 0 && d->ldg0 >= d->C, "uz_bn_relu_bwd: bad ldg0");
  if (g1) UZ_REQUIRE(d->ldg1 % vec == 0 && d->ldg1 >= d->C, "uz_bn_relu_bwd: bad ldg1");
  if (gp) {
    UZ_REQUIRE(d->ldgp % vec == 0 && d->ldgp >= d->C, "uz_bn_relu_bwd: bad ldgp");
  }
  UZ_REQUIRE(g0 || g1 || gp, "uz_bn_relu_bwd: no incoming gradient");
  return UZ_OK;
}

static void bnbwd_shape(const uz_bnbwd_desc* d, bool pool, dim3* grid, dim3* block, int pass = 1) {
  const int vec = d->dtype == UZ_BF16 ? 8 : 4;
  const long long units = pool ? (long long)d->N * ((d->H + 1) / 2) * ((d->W + 1) / 2) : (long long)d->N * d->H * d->W;
  const int f = uz_tune_flags() & 6;   // ablation build: workgroups per CU of the reduce pass (2: 8, 4: 6)
  reduce_shape(d->C / vec, units, grid, block, pass == 2 ? 8 : (f == 2 ? 8 : f == 4 ? 6 : 4));
}

template <typename T, int PASS, bool FIN = false>
static int bnbwd_launch(const uz_bnbwd_desc* d, const BnBwdArgs& a, bool pool, hipStream_t s) {
  constexpr int VEC = ElemTraits<T>::VEC;
  dim3 grid, block;
  bnbwd_shape(d, pool, &grid, &block, PASS);
  const size_t shm = PASS == 1 ? (size_t)256 * 2 * VEC * sizeof(float) : 0;
  if constexpr (FIN) {
    UZ_REQUIRE((long long)grid.x * grid.y >= (a.C + a.fin.ew - 1) / a.fin.ew && block.x * block.y == 256,
               "uz_bn_relu_bwd_apply_fin: the grid is smaller than the finalize (ask uz_bn_relu_bwd_apply_fin_supported)");
  }
  if (pool) {
    hipLaunchKernelGGL((bn_relu_bwd_kernel<T, true, PASS, FIN>), grid, block, shm, s, a);
  } else {
    hipLaunchKernelGGL((bn_relu_bwd_kernel<T, false, PASS, FIN>), grid, block, shm, s, a);
  }
  UZ_LAUNCH_CHECK("uz_bn_relu_bwd");
  return UZ_OK;
}

static BnBwdArgs bnbwd_args(const uz_bnbwd_desc* d, const void* y, const float* scale, const float* shift,
                            const float* mean, const float* invstd, const void* g0, const void* g1,
                            const void* gp) {
  BnBwdArgs a;
  a.y = y;
  a.g0 = g0;
  a.g1 = g1;
  a.gp = gp;
  a.dy = nullptr;
  a.scale = scale;
  a.shift = shift;
  a.mean = mean;
  a.invstd = invstd;
  a.sums = nullptr;
  a.partials = nullptr;
  a.dgamma = nullptr;
  a.dbeta = nullptr;
  a.inv_count = 0.0;
  a.N = d->N;
  a.H = d->H;
  a.W = d->W;
  a.C = d->C;
  a.ldy = d->ldy;
  a.ldg0 = d->ldg0;
  a.ldg1 = d->ldg1;
  a.ldgp = d->ldgp;
  a.pool_ceil = d->pool_ceil;
  a.lddy = d->lddy;
  a.fin = BnFin{};
  return a;
}

static void bnbwd_finalize(const float* partial, int rows, int C, double* sums, float* dgamma, float* dbeta,
                           hipStream_t s) {
  if (rows >= 128 && C <= 512)
    hipLaunchKernelGGL(bn_bwd_finalize_kernel<8>, dim3(uz_cdiv(C, 8)), dim3(1024), 0, s, partial, rows, C, sums, dgamma, dbeta);
  else
    hipLaunchKernelGGL(bn_bwd_finalize_kernel<32>, dim3(uz_cdiv(C, 32)), dim3(1024), 0, s, partial, rows, C, sums, dgamma, dbeta);
}

extern "C" long long uz_bn_relu_bwd_workspace_bytes(const uz_bnbwd_desc* d, int has_pool_grad) {
  UZ_REQUIRE(d != nullptr && d->C > 0 && d->N > 0 && d->H > 0 && d->W > 0, "uz_bn_relu_bwd_workspace_bytes: bad descriptor");
  UZ_REQUIRE(d->dtype == UZ_F32 || d->dtype == UZ_BF16, "uz_bn_relu_bwd_workspace_bytes: bad dtype");
  dim3 grid, block;
  bnbwd_shape(d, has_pool_grad != 0, &grid, &block);
  return (long long)grid.x * 2 * d->C * (long long)sizeof(float);
}

extern "C" int uz_bn_relu_bwd_reduce(const uz_bnbwd_desc* d, const void* y, const float* scale,
                                     const float* shift, const float* mean, const float* invstd,
                                     const void* g0, const void* g1, const void* gpool, void* workspace,
                                     double* sums, float* dgamma, float* dbeta, void* stream) {
  const int rc = bnbwd_check(d, g0, g1, gpool);
  if (rc != UZ_OK) return rc;
  UZ_REQUIRE(y && scale && shift && mean && invstd && sums && workspace, "uz_bn_relu_bwd_reduce: null pointer");
  UZ_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "uz_bn_relu_bwd_reduce: dgamma/dbeta");
  BnBwdArgs a = bnbwd_args(d, y, scale, shift, mean, invstd, g0, g1, gpool);
  a.partials = static_cast<float*>(workspace);
  hipStream_t s = (hipStream_t)stream;
  const int rc2 = d->dtype == UZ_BF16 ? bnbwd_launch<bf16_t, 1>(d, a, gpool != nullptr, s)
                                      : bnbwd_launch<float, 1>(d, a, gpool != nullptr, s);
  if (rc2 != UZ_OK) return rc2;
  dim3 grid, block;
  bnbwd_shape(d, gpool != nullptr, &grid, &block);
  bnbwd_finalize(static_cast<const float*>(workspace), (int)grid.x, d->C, sums, dgamma, dbeta, s);
  UZ_LAUNCH_CHECK("uz_bn_relu_bwd_reduce(finalize)");
  return UZ_OK;
}

extern "C" int uz_bn_bwd_finalize(const float* partial, int rows, int C, double* sums, float* dgamma, float* dbeta,
                                  void* stream) {
  UZ_REQUIRE(partial && sums && rows > 0 && C > 0, "uz_bn_bwd_finalize: bad arguments");
  UZ_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "uz_bn_bwd_finalize: dgamma/dbeta");
  bnbwd_finalize(partial, rows, C, sums, dgamma, dbeta, (hipStream_t)stream);
  UZ_LAUNCH_CHECK("uz_bn_bwd_finalize");
  return UZ_OK;
}

extern "C" int uz_bn_relu_bwd_apply(const uz_bnbwd_desc* d, const void* y, const float* scale,
                                    const float* shift, const float* mean, const float* invstd,
                                    const void* g0, const void* g1, const void* gpool, const double* sums,
                                    double count, void* dy, void* stream) {
  const int rc = bnbwd_check(d, g0, g1, gpool);
  if (rc != UZ_OK) return rc;
  UZ_REQUIRE(y && scale && shift && mean && invstd && sums && dy, "uz_bn_relu_bwd_apply: null pointer");
  UZ_REQUIRE(count > 0, "uz_bn_relu_bwd_apply: count");
  const int vec = d->dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(d->lddy % vec == 0 && d->lddy >= d->C, "uz_bn_relu_bwd_apply: bad lddy");
  BnBwdArgs a = bnbwd_args(d, y, scale, shift, mean, invstd, g0, g1, gpool);
  a.sums = const_cast<double*>(sums);
  a.dy = dy;
  a.inv_count = 1.0 / count;
  hipStream_t s = (hipStream_t)stream;
  return d->dtype == UZ_BF16 ? bnbwd_launch<bf16_t, 2>(d, a, gpool != nullptr, s)
                             : bnbwd_launch<float, 2>(d, a, gpool != nullptr, s);
}

// the first pass alone: partial rows [rows][2][C] into the workspace, rows = uz_bn_relu_bwd_workspace_bytes / (8 C)
extern "C" int uz_bn_relu_bwd_reduce_rows(const uz_bnbwd_desc* d, const void* y, const float* scale, const float* shift,
                                          const float* mean, const float* invstd, const void* g0, const void* g1,
                                          const void* gpool, void* workspace, void* stream) {
  const int rc = bnbwd_check(d, g0, g1, gpool);
  if (rc != UZ_OK) return rc;
  UZ_REQUIRE(y && scale && shift && mean && invstd && workspace, "uz_bn_relu_bwd_reduce_rows: null pointer");
  BnBwdArgs a = bnbwd_args(d, y, scale, shift, mean, invstd, g0, g1, gpool);
  a.partials = static_cast<float*>(workspace);
  hipStream_t s = (hipStream_t)stream;
  return d->dtype == UZ_BF16 ? bnbwd_launch<bf16_t, 1>(d, a, gpool != nullptr, s) : bnbwd_launch<float, 1>(d, a, gpool != nullptr, s);
}

extern "C" int uz_bn_relu_bwd_apply_fin(const uz_bnbwd_desc* d, const void* y, const float* scale, const float* shift,
                                        const float* mean, const float* invstd, const void* g0, const void* g1,
                                        const void* gpool, const float* partial, int rows, double* sums, float* dgamma,
                                        float* dbeta, int* flag, double count, void* dy, void* stream) {
  const int rc = bnbwd_check(d, g0, g1, gpool);
  if (rc != UZ_OK) return rc;
  UZ_REQUIRE(y && scale && shift && mean && invstd && sums && dy && partial && flag, "uz_bn_relu_bwd_apply_fin: null pointer");
  UZ_REQUIRE(count > 0 && rows > 0, "uz_bn_relu_bwd_apply_fin: count / rows");
  UZ_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "uz_bn_relu_bwd_apply_fin: dgamma/dbeta");
  const int vec = d->dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(d->lddy % vec == 0 && d->lddy >= d->C, "uz_bn_relu_bwd_apply_fin: bad lddy");
  BnBwdArgs a = bnbwd_args(d, y, scale, shift, mean, invstd, g0, g1, gpool);
  a.sums = sums;
  a.dy = dy;
  a.inv_count = 1.0 / count;
  a.fin.part = partial; a.fin.rows = rows; a.fin.ew = fin_ew(rows, d->C);
  a.fin.sums = sums; a.fin.dgamma = dgamma; a.fin.dbeta = dbeta; a.fin.flag = flag;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid, block;
  bnbwd_shape(d, gpool != nullptr, &grid, &block, 2);
  if ((long long)grid.x * grid.y < (d->C + a.fin.ew - 1) / a.fin.ew) {   // grid smaller than the finalize: the two launches
    bnbwd_finalize(partial, rows, d->C, sums, dgamma, dbeta, s);
    UZ_LAUNCH_CHECK("uz_bn_relu_bwd_apply_fin(finalize)");
    a.fin = BnFin{};
    return d->dtype == UZ_BF16 ? bnbwd_launch<bf16_t, 2>(d, a, gpool != nullptr, s) : bnbwd_launch<float, 2>(d, a, gpool != nullptr, s);
  }
  return d->dtype == UZ_BF16 ? bnbwd_launch<bf16_t, 2, true>(d, a, gpool != nullptr, s)
                             : bnbwd_launch<float, 2, true>(d, a, gpool != nullptr, s);
}

extern "C" int uz_outconv_fwd(int dtype, const void* x, int ldx, int N, int HW, int C, const float* w,
                              const float* b, int Kout, float* out_nchw, void* stream) {
  UZ_REQUIRE(dtype == UZ_F32 || dtype == UZ_BF16, "uz_outconv_fwd: bad dtype");
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(x && w && b && out_nchw, "uz_outconv_fwd: null pointer");
  UZ_REQUIRE(Kout >= 1 && Kout <= OUTCONV_MAXK, "uz_outconv_fwd: Kout=%d (max %d)", Kout, OUTCONV_MAXK);
  UZ_REQUIRE(C % vec == 0 && C / vec <= 64, "uz_outconv_fwd: C=%d unsupported", C);
  UZ_REQUIRE(ldx % vec == 0 && ldx >= C && N > 0 && HW > 0 && (long long)N * HW < (1LL << 31),
             "uz_outconv_fwd: bad shape");
  const int ppb = 256 / lanes_per_pixel(C / vec);
  const int grid = grid_for(((long long)N * HW + ppb - 1) / ppb, 1);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UZ_BF16) {
    UZ_KOUT_SWITCH(Kout, hipLaunchKernelGGL((outconv_fwd_kernel<bf16_t, KOUT>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x, ldx, N, HW, C, w, b, out_nchw, (const float*)nullptr, (const float*)nullptr))
  } else {
    UZ_KOUT_SWITCH(Kout, hipLaunchKernelGGL((outconv_fwd_kernel<float, KOUT>), dim3(grid), dim3(256), 0, s, (const float*)x, ldx, N, HW, C, w, b, out_nchw, (const float*)nullptr, (const float*)nullptr))
  }
  UZ_LAUNCH_CHECK("uz_outconv_fwd");
  return UZ_OK;
}

extern "C" int uz_outconv_fwd_xf(int dtype, const void* y, int ldy, int N, int HW, int C, const float* scale,
                                 const float* shift, const float* w, const float* b, int Kout, float* out_nchw,
                                 void* stream) {
  UZ_REQUIRE(dtype == UZ_BF16, "uz_outconv_fwd_xf: bf16 only");
  UZ_REQUIRE(y && scale && shift && w && b && out_nchw, "uz_outconv_fwd_xf: null pointer");
  UZ_REQUIRE(Kout >= 1 && Kout <= OUTCONV_MAXK, "uz_outconv_fwd_xf: Kout=%d (max %d)", Kout, OUTCONV_MAXK);
  UZ_REQUIRE(C % 8 == 0 && C / 8 <= 64, "uz_outconv_fwd_xf: C=%d unsupported", C);
  UZ_REQUIRE(ldy % 8 == 0 && ldy >= C && N > 0 && HW > 0 && (long long)N * HW < (1LL << 31), "uz_outconv_fwd_xf: bad shape");
  const int ppb = 256 / lanes_per_pixel(C / 8);
  const int grid = grid_for(((long long)N * HW + ppb - 1) / ppb, 1);
  hipStream_t s = (hipStream_t)stream;
  UZ_KOUT_SWITCH(Kout, hipLaunchKernelGGL((outconv_fwd_kernel<bf16_t, KOUT, true>), dim3(grid), dim3(256), 0, s, (const bf16_t*)y, ldy, N, HW, C, w, b, out_nchw, scale, shift))
  UZ_LAUNCH_CHECK("uz_outconv_fwd_xf");
  return UZ_OK;
}

static int outconv_bwd_grid(int dtype, int N, int HW, int C) {
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  const int ppb = 256 / lanes_per_pixel(C / vec);
  long long g = ((long long)N * HW + (long long)ppb * 4 - 1) / ((long long)ppb * 4);
  if (g > UZ_NUM_CU * 4) g = UZ_NUM_CU * 4;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" long long uz_outconv_bwd_workspace_bytes(int dtype, int N, int HW, int C, int Kout) {
  UZ_REQUIRE(dtype == UZ_F32 || dtype == UZ_BF16, "uz_outconv_bwd_workspace_bytes: bad dtype");
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(N > 0 && HW > 0 && Kout >= 1 && Kout <= OUTCONV_MAXK && C % vec == 0 && C / vec <= 64,
             "uz_outconv_bwd_workspace_bytes: bad shape");
  return (long long)outconv_bwd_grid(dtype, N, HW, C) * Kout * (C + 1) * (long long)sizeof(float);
}

extern "C" int uz_outconv_bwd_rows(int dtype, int N, int HW, int C) {
  UZ_REQUIRE(dtype == UZ_F32 || dtype == UZ_BF16, "uz_outconv_bwd_rows: bad dtype");
  return outconv_bwd_grid(dtype, N, HW, C);
}

extern "C" int uz_outconv_bwd_bnred(int dtype, const void* x, int ldx, int N, int HW, int C, const float* w,
                                    int Kout, const float* g_nchw, void* dx, int lddx, float* dw, float* db,
                                    void* workspace, const void* bn_y, int ld_bny, const float* scale,
                                    const float* shift, const float* mean, const float* invstd, float* bn_partial,
                                    void* stream) {
  UZ_REQUIRE(dtype == UZ_BF16, "uz_outconv_bwd_bnred: bf16 only");
  // x == NULL: the activation was never written down (uz_outconv_fwd_xf); the kernel forms it from bn_y
  UZ_REQUIRE(w && g_nchw && dw && db && workspace && dx && bn_y && scale && shift && mean && invstd && bn_partial,
             "uz_outconv_bwd_bnred: null pointer");
  UZ_REQUIRE(Kout >= 1 && Kout <= OUTCONV_MAXK, "uz_outconv_bwd_bnred: Kout=%d", Kout);
  UZ_REQUIRE(C % 8 == 0 && C / 8 <= 64, "uz_outconv_bwd_bnred: C=%d unsupported", C);
  UZ_REQUIRE((x == nullptr || (ldx % 8 == 0 && ldx >= C)) && lddx % 8 == 0 && lddx >= C && ld_bny % 8 == 0 && ld_bny >= C && N > 0 &&
                 HW > 0 && (long long)N * HW < (1LL << 31), "uz_outconv_bwd_bnred: bad shape");
  const int g = outconv_bwd_grid(dtype, N, HW, C);
  float* part = static_cast<float*>(workspace);
  hipStream_t s = (hipStream_t)stream;
  if (x != nullptr) {
    UZ_KOUT_SWITCH(Kout, hipLaunchKernelGGL((outconv_bwd_kernel<bf16_t, KOUT, true>), dim3((unsigned)g), dim3(256), 0, s,
                                            (const bf16_t*)x, ldx, N, HW, C, w, g_nchw, (bf16_t*)dx, lddx, part,
                                            (const bf16_t*)bn_y, ld_bny, scale, shift, mean, invstd, bn_partial))
  } else {
    UZ_KOUT_SWITCH(Kout, hipLaunchKernelGGL((outconv_bwd_kernel<bf16_t, KOUT, true, true>), dim3((unsigned)g), dim3(256), 0, s,
                                            (const bf16_t*)nullptr, 0, N, HW, C, w, g_nchw, (bf16_t*)dx, lddx, part,
                                            (const bf16_t*)bn_y, ld_bny, scale, shift, mean, invstd, bn_partial))
  }
  UZ_LAUNCH_CHECK("uz_outconv_bwd_bnred");
  const int ne = Kout * (C + 1);
  hipLaunchKernelGGL(outconv_bwd_finalize_kernel, dim3(uz_cdiv(ne, 32)), dim3(1024), 0, s, part, g, Kout, C, dw, db);
  UZ_LAUNCH_CHECK("uz_outconv_bwd_bnred(finalize)");
  return UZ_OK;
}

extern "C" int uz_outconv_bwd(int dtype, const void* x, int ldx, int N, int HW, int C, const float* w,
                              int Kout, const float* g_nchw, void* dx, int lddx, float* dw, float* db,
                              void* workspace, void* stream) {
  UZ_REQUIRE(dtype == UZ_F32 || dtype == UZ_BF16, "uz_outconv_bwd: bad dtype");
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(x && w && g_nchw && dw && db && workspace, "uz_outconv_bwd: null pointer");
  UZ_REQUIRE(Kout >= 1 && Kout <= OUTCONV_MAXK, "uz_outconv_bwd: Kout=%d", Kout);
  UZ_REQUIRE(C % vec == 0 && C / vec <= 64, "uz_outconv_bwd: C=%d unsupported", C);
  UZ_REQUIRE(ldx % vec == 0 && ldx >= C && N > 0 && HW > 0 && (long long)N * HW < (1LL << 31),
             "uz_outconv_bwd: bad shape");
  if (dx) UZ_REQUIRE(lddx % vec == 0 && lddx >= C, "uz_outconv_bwd: bad lddx");
  const int g = outconv_bwd_grid(dtype, N, HW, C);
  float* part = static_cast<float*>(workspace);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UZ_BF16) {
    UZ_KOUT_SWITCH(Kout, hipLaunchKernelGGL((outconv_bwd_kernel<bf16_t, KOUT>), dim3((unsigned)g), dim3(256), 0, s, (const bf16_t*)x, ldx, N, HW, C, w, g_nchw, (bf16_t*)dx, lddx, part, (const bf16_t*)nullptr, 0, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)nullptr))
  } else {
    UZ_KOUT_SWITCH(Kout, hipLaunchKernelGGL((outconv_bwd_kernel<float, KOUT>), dim3((unsigned)g), dim3(256), 0, s, (const float*)x, ldx, N, HW, C, w, g_nchw, (float*)dx, lddx, part, (const float*)nullptr, 0, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)nullptr))
  }
  UZ_LAUNCH_CHECK("uz_outconv_bwd");
  const int ne = Kout * (C + 1);
  hipLaunchKernelGGL(outconv_bwd_finalize_kernel, dim3(uz_cdiv(ne, 32)), dim3(1024), 0, s, part, g, Kout, C, dw, db);
  UZ_LAUNCH_CHECK("uz_outconv_bwd(finalize)");
  return UZ_OK;
}

extern "C" int uz_colsum(int dtype, const void* x, int ld, int P, int C, float* out, void* stream) {
  UZ_REQUIRE(dtype == UZ_F32 || dtype == UZ_BF16, "uz_colsum: bad dtype");
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(x && out && P > 0 && C > 0 && C % vec == 0 && ld % vec == 0 && ld >= C, "uz_colsum: bad args");
  dim3 grid, block;
  reduce_shape(C / vec, P, &grid, &block);
  const size_t shm = (size_t)256 * vec * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UZ_BF16)
    hipLaunchKernelGGL((colsum_kernel<bf16_t, false>), grid, block, shm, s, (const bf16_t*)x, ld, (long long)P, C, out);
  else
    hipLaunchKernelGGL((colsum_kernel<float, false>), grid, block, shm, s, (const float*)x, ld, (long long)P, C, out);
  UZ_LAUNCH_CHECK("uz_colsum");
  return UZ_OK;
}

extern "C" long long uz_colsum_workspace_bytes(int dtype, int P, int C) {
  UZ_REQUIRE(dtype == UZ_F32 || dtype == UZ_BF16, "uz_colsum_workspace_bytes: bad dtype");
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(P > 0 && C > 0 && C % vec == 0, "uz_colsum_workspace_bytes: bad shape");
  dim3 grid, block;
  reduce_shape(C / vec, P, &grid, &block);
  return (long long)grid.x * C * (long long)sizeof(float);
}

extern "C" int uz_colsum_ws(int dtype, const void* x, int ld, int P, int C, float* out, void* workspace,
                            void* stream) {
  UZ_REQUIRE(dtype == UZ_F32 || dtype == UZ_BF16, "uz_colsum_ws: bad dtype");
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(x && out && workspace && P > 0 && C > 0 && C % vec == 0 && ld % vec == 0 && ld >= C, "uz_colsum_ws: bad args");
  dim3 grid, block;
  reduce_shape(C / vec, P, &grid, &block);
  const size_t shm = (size_t)256 * vec * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  float* part = static_cast<float*>(workspace);
  if (dtype == UZ_BF16)
    hipLaunchKernelGGL((colsum_kernel<bf16_t, true>), grid, block, shm, s, (const bf16_t*)x, ld, (long long)P, C, part);
  else
    hipLaunchKernelGGL((colsum_kernel<float, true>), grid, block, shm, s, (const float*)x, ld, (long long)P, C, part);
  UZ_LAUNCH_CHECK("uz_colsum_ws");
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3(uz_cdiv(C, 32)), dim3(1024), 0, s, part, (int)grid.x, C, out);
  UZ_LAUNCH_CHECK("uz_colsum_ws(finalize)");
  return UZ_OK;
}

extern "C" int uz_colstats_rows(int dtype, int P, int C) {
  UZ_REQUIRE(dtype == UZ_F32 || dtype == UZ_BF16, "uz_colstats_rows: bad dtype");
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(P > 0 && C > 0 && C % vec == 0, "uz_colstats_rows: bad shape");
  dim3 grid, block;
  reduce_shape(C / vec, P, &grid, &block, 4);
  return (int)grid.x;
}

extern "C" int uz_colstats(int dtype, const void* x, int ld, int P, int C, float* partial, void* stream) {
  UZ_REQUIRE(dtype == UZ_F32 || dtype == UZ_BF16, "uz_colstats: bad dtype");
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(x && partial && P > 0 && C > 0 && C % vec == 0 && ld % vec == 0 && ld >= C, "uz_colstats: bad args");
  dim3 grid, block;
  reduce_shape(C / vec, P, &grid, &block, 4);
  const size_t shm = (size_t)256 * 2 * vec * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UZ_BF16)
    hipLaunchKernelGGL((colstats_kernel<bf16_t>), grid, block, shm, s, (const bf16_t*)x, ld, (long long)P, C, partial);
  else
    hipLaunchKernelGGL((colstats_kernel<float>), grid, block, shm, s, (const float*)x, ld, (long long)P, C, partial);
  UZ_LAUNCH_CHECK("uz_colstats");
  return UZ_OK;
}

extern "C" int uz_pack_weights(int dtype, int mode, const float* w, int Co, int Ci, int T, int Kpad,
                               void* dst, void* stream) {
  UZ_REQUIRE(dtype == UZ_F32 || dtype == UZ_BF16, "uz_pack_weights: bad dtype");
  UZ_REQUIRE(w && dst && Co > 0 && Ci > 0 && T > 0, "uz_pack_weights: bad args");
  UZ_REQUIRE(mode >= UZ_PACK_CONV_FWD && mode <= UZ_PACK_IM2COL, "uz_pack_weights: bad mode %d", mode);
  long long total = (long long)Co * Ci * T;
  if (mode == UZ_PACK_IM2COL) {
    UZ_REQUIRE(Kpad >= T * Ci, "uz_pack_weights: Kpad too small");
    total = (long long)Co * Kpad;
  }
  hipStream_t s = (hipStream_t)stream;
  const int grid = grid_for(total, 256);
  if (dtype == UZ_BF16)
    hipLaunchKernelGGL((pack_weights_kernel<bf16_t>), dim3(grid), dim3(256), 0, s, mode, w, Co, Ci, T, Kpad, (bf16_t*)dst, total);
  else
    hipLaunchKernelGGL((pack_weights_kernel<float>), dim3(grid), dim3(256), 0, s, mode, w, Co, Ci, T, Kpad, (float*)dst, total);
  UZ_LAUNCH_CHECK("uz_pack_weights");
  return UZ_OK;
}

// 3x3 conv weights, both kernel layouts from ONE coalesced read: a workgroup stages a
// [32 co][32 ci][9] block of the OIHW tensor in LDS and writes the forward layout (ci fastest) and
// the input-gradient layout (co fastest, taps flipped) from it.
template <typename T>
__global__ __launch_bounds__(256) void pack_conv3x3_tiled_kernel(const uz_pack3x3_item* __restrict__ items,
                                                                 int n, int total_tiles) {
  // [tap][co][ci] with rows of 33 and tap planes of 32 * 33 + 4 floats: the staging writes walk tap fastest, and a plane
  // stride that is a multiple of 32 banks put the nine taps of one (co, ci) on ONE bank (9-way conflict on every write)
  constexpr int TP = 32 * 33 + 4;
  __shared__ float tile[9 * TP];
  for (int gt = blockIdx.x; gt < total_tiles; gt += gridDim.x) {
    int lo = 0, hi = n - 1;  // item that owns global tile gt (tile_begin is a prefix sum)
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (items[mid].tile_begin <= gt) lo = mid; else hi = mid - 1;
    }
    const uz_pack3x3_item it = items[lo];
    const int tl = gt - it.tile_begin;
    const int tiles_ci = it.Ci / 32;
    T* __restrict__ df = static_cast<T*>(it.dst_fwd);
    T* __restrict__ dd = static_cast<T*>(it.dst_dgrad);
    const int co0 = (tl / tiles_ci) * 32, ci0 = (tl % tiles_ci) * 32;
    __syncthreads();
    // all 36 loads of a thread are requested before the first LDS write: as a rolled loop every pass waited for its own
    // memory round trip (36 in a row: the kernel ran at 102 us for 124 MB)
    float stage[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) {
      const int e = threadIdx.x + 256 * i;
      const int co_l = e / 288, r = e - co_l * 288;
      stage[i] = it.src[((size_t)(co0 + co_l) * it.Ci + ci0) * 9 + r];
    }
#pragma unroll
    for (int i = 0; i < 36; ++i) {
      const int e = threadIdx.x + 256 * i;
      const int co_l = e / 288, r = e - co_l * 288;
      const int ci_l = r / 9, tap = r - ci_l * 9;
      tile[tap * TP + co_l * 33 + ci_l] = stage[i];
    }
    __syncthreads();
    // sixteen-byte stores (bf16: 8 values, fp32: 4): piece = (row, part) with row = (y, tap), 32 / VEC parts per row.
    // (Two-byte stores, one value per lane, ran this kernel at 2.4 TB/s.)
    constexpr int VEC = ElemTraits<T>::VEC, PPR = 32 / VEC;
    for (int e = threadIdx.x; e < 9 * 32 * PPR; e += 256) {
      const int part = e % PPR, row = e / PPR;
      const int y = row & 31, tap = row >> 5;
      if (df != nullptr) {  // y = co, the part's values run over ci
        Vec16<T> v;
#pragma unroll
        for (int k = 0; k < VEC; ++k) v.v[k] = (T)tile[tap * TP + y * 33 + part * VEC + k];
        st16(df + (size_t)(co0 + y) * 9 * it.Ci + tap * it.Ci + ci0 + part * VEC, v);
      }
      if (dd != nullptr) {  // y = ci, the part's values run over co
        Vec16<T> v;
#pragma unroll
        for (int k = 0; k < VEC; ++k) v.v[k] = (T)tile[tap * TP + (part * VEC + k) * 33 + y];
        st16(dd + (size_t)(ci0 + y) * 9 * it.Co + (8 - tap) * it.Co + co0 + part * VEC, v);
      }
    }
  }
}

extern "C" int uz_pack_conv3x3_batched(int dtype, const uz_pack3x3_item* items_device, int n_items,
                                       int total_tiles, void* stream) {
  UZ_REQUIRE(dtype == UZ_F32 || dtype == UZ_BF16, "uz_pack_conv3x3_batched: bad dtype");
  UZ_REQUIRE(items_device && n_items > 0 && total_tiles > 0, "uz_pack_conv3x3_batched: bad args");
  const dim3 grid(total_tiles < 4096 ? total_tiles : 4096);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UZ_BF16)
    hipLaunchKernelGGL((pack_conv3x3_tiled_kernel<bf16_t>), grid, dim3(256), 0, s, items_device, n_items, total_tiles);
  else
    hipLaunchKernelGGL((pack_conv3x3_tiled_kernel<float>), grid, dim3(256), 0, s, items_device, n_items, total_tiles);
  UZ_LAUNCH_CHECK("uz_pack_conv3x3_batched");
  return UZ_OK;
}

extern "C" int uz_pack_weights_batched(int dtype, const uz_pack_item* items_device, int n_items,
                                       long long total_elements, void* stream) {
  UZ_REQUIRE(dtype == UZ_F32 || dtype == UZ_BF16, "uz_pack_weights_batched: bad dtype");
  UZ_REQUIRE(items_device && n_items > 0 && total_elements > 0, "uz_pack_weights_batched: bad args");
  hipStream_t s = (hipStream_t)stream;
  UZ_REQUIRE(n_items <= 65535, "uz_pack_weights_batched: too many items");
  // workgroups per item: the host does not see the items' sizes (the table lives in device memory).  Few items (unet: 12, the
  // largest a 1024 x 512 ConvTranspose of 2 M elements) want the chip per item: 128 -> 512 took the launch from 44 to 18 us;
  // many items (swin_unet_v2: 77 Linear weights) pay for the empty workgroups instead: 40 -> 54 us at 512.
  const dim3 grid(n_items <= 16 ? 512 : (n_items <= 40 ? 256 : 128), n_items);
  if (dtype == UZ_BF16)
    hipLaunchKernelGGL((pack_weights_batched_kernel<bf16_t>), grid, dim3(256), 0, s, items_device, n_items, total_elements);
  else
    hipLaunchKernelGGL((pack_weights_batched_kernel<float>), grid, dim3(256), 0, s, items_device, n_items, total_elements);
  UZ_LAUNCH_CHECK("uz_pack_weights_batched");
  return UZ_OK;
}

extern "C" int uz_im2col3x3_nchw(int dtype, const float* x_nchw, int N, int C, int H, int W, int Kpad,
                                 void* dst, void* stream) {
  UZ_REQUIRE(dtype == UZ_F32 || dtype == UZ_BF16, "uz_im2col3x3_nchw: bad dtype");
  UZ_REQUIRE(x_nchw && dst && N > 0 && C > 0 && H > 0 && W > 0 && Kpad >= 9 * C, "uz_im2col3x3_nchw: bad args");
  const int vec_ = dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(Kpad % vec_ == 0, "uz_im2col3x3_nchw: Kpad must be a multiple of %d", vec_);
  hipStream_t s = (hipStream_t)stream;
  const long long npix = (long long)N * H * W;
  if (C <= 4 && Kpad <= 64) {
    const int grid = grid_for(npix, 256);
#define UZ_IM2COL_C(TT, CC) \
  hipLaunchKernelGGL((im2col3x3_smallc_kernel<TT, CC>), dim3(grid), dim3(256), 0, s, x_nchw, N, H, W, Kpad, (TT*)dst, npix)
    if (dtype == UZ_BF16) {
      if (C == 1) UZ_IM2COL_C(bf16_t, 1); else if (C == 2) UZ_IM2COL_C(bf16_t, 2);
      else if (C == 3) UZ_IM2COL_C(bf16_t, 3); else UZ_IM2COL_C(bf16_t, 4);
    } else {
      if (C == 1) UZ_IM2COL_C(float, 1); else if (C == 2) UZ_IM2COL_C(float, 2);
      else if (C == 3) UZ_IM2COL_C(float, 3); else UZ_IM2COL_C(float, 4);
    }
#undef UZ_IM2COL_C
  } else {
    const long long total = npix * (Kpad / vec_);
    const int grid = grid_for(total, 256);
    if (dtype == UZ_BF16)
      hipLaunchKernelGGL((im2col3x3_kernel<bf16_t>), dim3(grid), dim3(256), 0, s, x_nchw, N, C, H, W, Kpad, (bf16_t*)dst, total);
    else
      hipLaunchKernelGGL((im2col3x3_kernel<float>), dim3(grid), dim3(256), 0, s, x_nchw, N, C, H, W, Kpad, (float*)dst, total);
  }
  UZ_LAUNCH_CHECK("uz_im2col3x3_nchw");
  return UZ_OK;
}

// ---- CUs held back from the persistent grids (process-wide: set once, before the plans are queried) --------------------
#include <atomic>
static std::atomic<int> g_cu_reserve{0};
int uz_num_cu() { return UZ_NUM_CU_HW - g_cu_reserve.load(std::memory_order_relaxed); }
extern "C" int uz_set_cu_reserve(int n) {
  UZ_REQUIRE(n >= 0 && n <= UZ_NUM_CU_HW / 2, "uz_set_cu_reserve: %d not in 0 .. %d", n, UZ_NUM_CU_HW / 2);
  g_cu_reserve.store(n, std::memory_order_relaxed);
  return UZ_OK;
}
extern "C" int uz_get_cu_reserve(void) { return g_cu_reserve.load(std::memory_order_relaxed); }

// ---- error string -------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void uz_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* uz_last_error_string(void) { return g_err; }

// ---- measurement hook (see uz_common.h) ------------------------------------------------------------------------------
static thread_local hipEvent_t g_prof_e0 = nullptr, g_prof_e1 = nullptr;
static thread_local int g_prof_launches = -1;   // -1: not armed
bool uz_prof_take(hipEvent_t* e0, hipEvent_t* e1) {
  if (g_prof_launches < 0) return false;
  *e0 = g_prof_launches == 0 ? g_prof_e0 : nullptr;
  *e1 = g_prof_e1;
  ++g_prof_launches;
  return true;
}
extern "C" int uz_profile_arm(void* start_event, void* stop_event) {
  UZ_REQUIRE(start_event && stop_event, "uz_profile_arm: null event");
  g_prof_e0 = (hipEvent_t)start_event;
  g_prof_e1 = (hipEvent_t)stop_event;
  g_prof_launches = 0;
  return UZ_OK;
}
extern "C" int uz_profile_disarm(void) {
  const int n = g_prof_launches;
  g_prof_launches = -1;
  return n < 0 ? 0 : n;   // kernel launches recorded since uz_profile_arm
}
extern "C" int uz_abi_version(void) { return UZ_ABI_VERSION; }
extern "C" int uz_build_ablate(void) {
#ifdef UZ_ABLATE
  return 1;
#else
  return 0;
#endif
}

// ------------------------------------------------------------------------------------------
// clip_grad_norm_ + AdamW over flat fp32 buffers (reference step tail, training_loop.py:119-121:
// torch.nn.utils.clip_grad_norm_(max_norm=1.0); optimizer.step() with torch.optim.AdamW).
// Three launches: per-workgroup sums of squares -> one workgroup derives the clip coefficient,
// advances the step counter and the bias corrections -> one streaming pass over p, g, m, v
// (28 bytes per parameter; the separate torch calls move 48).
// ------------------------------------------------------------------------------------------
namespace {

constexpr int ADAMW_ROWS = 1024;

__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, long long n,
                                                             double* __restrict__ partial) {
  __shared__ double red[256];
  double s = 0.0;
  const long long n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const float4 v = g4[i];
    s += (double)(v.x * v.x + v.y * v.y) + (double)(v.z * v.z + v.w * v.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float v = g[(n4 << 2) + threadIdx.x];
    s += (double)v * v;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// scal[0] = clip coefficient, scal[1] = 1 - beta1^t, scal[2] = 1 - beta2^t, scal[3] = total gradient norm
__global__ __launch_bounds__(256) void adamw_prepare_kernel(const double* __restrict__ partial, int rows, float max_norm,
                                                            float beta1, float beta2, float* __restrict__ step,
                                                            float* __restrict__ scal) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < rows; i += 256) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(red[0]);
    float coef = 1.f;
    if (max_norm > 0.f) {
      coef = max_norm / (norm + 1e-6f);   // torch: clamp(max_norm / (total_norm + 1e-6), max=1)
      if (coef > 1.f) coef = 1.f;
    }
    const float t = step[0] + 1.f;
    step[0] = t;
    scal[0] = coef;
    scal[1] = 1.f - powf(beta1, t);
    scal[2] = 1.f - powf(beta2, t);
    scal[3] = norm;
  }
}

__global__ __launch_bounds__(256) void adamw_apply_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v, long long n,
                                                          float lr, float beta1, float beta2, float eps, float wd,
                                                          const float* __restrict__ scal) {
  const float coef = scal[0], bc1 = scal[1], bc2 = scal[2];
  const float step_size = lr / bc1, rs2 = rsqrtf(bc2), decay = 1.f - lr * wd;
  auto upd = [&](float& pv, float gv, float& mv, float& vv) {
    gv *= coef;
    pv *= decay;                                   // decoupled weight decay
    mv = beta1 * mv + (1.f - beta1) * gv;          // torch: exp_avg.lerp_(grad, 1 - beta1)
    vv = beta2 * vv + (1.f - beta2) * gv * gv;
    pv -= step_size * mv / (sqrtf(vv) * rs2 + eps);
  };
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 pv = reinterpret_cast<float4*>(p)[i], mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    upd(pv.x, gv.x, mv.x, vv.x);
    upd(pv.y, gv.y, mv.y, vv.y);
    upd(pv.z, gv.z, mv.z, vv.z);
    upd(pv.w, gv.w, mv.w, vv.w);
    reinterpret_cast<float4*>(p)[i] = pv;
    reinterpret_cast<float4*>(m)[i] = mv;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    upd(p[i], g[i], m[i], v[i]);
  }
}

}  // namespace

extern "C" long long uz_clip_adamw_workspace_bytes(void) { return (long long)ADAMW_ROWS * 8 + 64; }

extern "C" int uz_clip_adamw(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, float max_norm, float* step,
                             void* workspace, void* stream) {
  UZ_REQUIRE(p && g && m && v && step && workspace && n > 0, "uz_clip_adamw: bad args");
  UZ_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)workspace) & 15) == 0,
             "uz_clip_adamw: buffers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  double* partial = static_cast<double*>(workspace);
  float* scal = reinterpret_cast<float*>(partial + ADAMW_ROWS);
  long long blocks = (n / 4 + 255) / 256;
  int rows = (int)(blocks < ADAMW_ROWS ? (blocks < 1 ? 1 : blocks) : ADAMW_ROWS);
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(rows), dim3(256), 0, s, g, n, partial);
  UZ_LAUNCH_CHECK("uz_clip_adamw(norm)");
  hipLaunchKernelGGL(adamw_prepare_kernel, dim3(1), dim3(256), 0, s, partial, rows, max_norm, beta1, beta2, step, scal);
  UZ_LAUNCH_CHECK("uz_clip_adamw(prepare)");
  const int grid = (int)(blocks < UZ_NUM_CU * 8 ? (blocks < 1 ? 1 : blocks) : UZ_NUM_CU * 8);
  hipLaunchKernelGGL(adamw_apply_kernel, dim3(grid), dim3(256), 0, s, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, scal);
  UZ_LAUNCH_CHECK("uz_clip_adamw(apply)");
  return UZ_OK;
}
