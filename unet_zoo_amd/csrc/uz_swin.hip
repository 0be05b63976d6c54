// Swin-UNet V2 specific kernels for gfx950 (MI355X)  (reference: unet_zoo/models/swin_unet_v2.py):
//   * patch extraction for PatchEmbed's Conv2d(k = s = patch)                           (:548-556)
//   * LayerNorm forward / backward over the channel dimension of a token tensor [P][C], with the
//     reference's token permutations folded into the addressing — PatchMerging's 2x2 gather+concat
//     (:315-332), PatchExpand / FinalPatchExpand_X4's 'b h w (p1 p2 c) -> b (h p1) (w p2) c'
//     (:352-362, :375-387) — and the block tail `shortcut + drop_path(norm1(.))` (:264-267) fused in
//   * window attention core, forward / backward                                          (:127-159):
//     cosine attention with learned per-entry temperature tau (clipped at 0.01), additive continuous
//     position bias, shifted-window mask computed from the region ids (:214-236), softmax, @v.
//     window_partition / torch.roll / window_reverse (:30-56, :246-262) are index arithmetic here:
//     a (window, head) workgroup reads q, k, v of its tokens from the [P][3C] qkv tensor and writes
//     the head's output back to the SAME token rows.
// The Linear layers around these run on the LDS-DMA GEMM (uz_gemm_dma.hip) and the weight-gradient
// kernels.  Everything here is bandwidth / latency bound: the attention core is 0.5 MFLOP per
// (window, head) and 3 % of the model's flops, so it is plain fp32 VALU code, one thread per query
// (forward, first backward phase) or per key (second backward phase), flash-style: the forward keeps
// only the row log-sum-exp, the backward recomputes P.
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

// ---------------------------------------------------------------------------------------------
// patches: out[p = (b, i, j)][k = (kh*ps + kw)*C + c] = x[b][c][i*ps + kh][j*ps + kw], zero for k >= ps*ps*C
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ x, int N, int C, int H, int W, int ps,
                                                       int Kpad, T* __restrict__ out) {
  const int Ho = H / ps, Wo = W / ps, K = ps * ps * C;
  const long long total = (long long)N * Ho * Wo * Kpad;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(idx % Kpad);
    const long long p = idx / Kpad;
    float v = 0.f;
    if (k < K) {
      const int c = k % C, tap = k / C, kh = tap / ps, kw = tap - kh * ps;
      const int j = (int)(p % Wo);
      const long long t = p / Wo;
      const int i = (int)(t % Ho), b = (int)(t / Ho);
      v = x[(((size_t)b * C + c) * H + i * ps + kh) * W + j * ps + kw];
    }
    out[idx] = (T)v;
  }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm.  Output token t (grid Ho x Wo, C channels); its input row is assembled by `mode`:
//   0 plain     : x[t][c]
//   1 merge 2x2 : Ho = H/2; channel segment s = c / (C/4) comes from input token (2i + (s&1), 2j + (s>>1)),
//                 channels c - s*C/4 (torch.cat([x0, x1, x2, x3], -1) of PatchMerging)
//   2 expand r  : Ho = H*r; output token (h*r + p1, w*r + p2) reads input token (h, w), channels
//                 (p1*r + p2)*C + c
// y = [res +] [sb[image] *] (xhat * gamma + beta); mean and rstd per token are kept for the backward.
// One wave per token, lanes stride over the 16-byte channel chunks (at most MAXIT per lane).
// ---------------------------------------------------------------------------------------------
struct FastDiv {
  unsigned m;
  int s;
};

struct LnArgs {
  const void* x;
  void* y;            // fwd: output; bwd: unused
  const void* res;    // fwd: optional residual (same layout as y)
  const void* g;      // bwd: gradient of y
  void* dx;           // bwd: gradient of x (mapped like x)
  const float* gamma;
  const float* beta;
  const float* sb;    // optional per-image factor of the normalised branch (stochastic depth)
  float* stats;       // [P_out][2] mean, rstd
  float* partial;     // bwd: [gridDim.x][2][C] sums of g*xhat (dgamma) and g (dbeta)
  int N, Ho, Wo, C, ldx, ldy, ldr, ldg, lddx, mode, r;
  float eps;
  int act;                // 1: GELU applied to the result (ACT instantiations)
  FastDiv fWo, fHo, fr;   // token index -> (image, row, column) and the expand sub-position without v_rcp sequences
  int Hin, Win;           // grid of x: (Ho, Wo) plain, (2 Ho, 2 Wo) merge, (Ho / r, Wo / r) expand
};

constexpr int LN_MAXIT = 8;     // 16-byte chunks per lane: C <= 64 * 8 * 4 = 2048 in fp32
constexpr int LN_MAXC = 2048;   // MixFFN_skip of MISSFormer's 512-channel stage: LayerNorm(4 * 512)

// Unsigned division by a launch constant (n < 2^31): q = umulhi(n, m) >> s with m = ceil(2^(32+s) / d),
// exact for every 31-bit n (Granlund & Montgomery); d = 1 is m = 0.  A runtime integer division costs ~25
// VALU instructions; the token -> (image, row, column[, sub-position]) decomposition needs four of them per
// token and lane and dominated the instruction count of these bandwidth kernels.
__device__ __forceinline__ int fdiv(int n, const FastDiv f) {
  return f.m == 0 ? n : (int)(__umulhi((unsigned)n, f.m) >> f.s);
}

// where token t of the normalised map reads x: pixel `pix` of x's grid, first channel `coff`
struct TokPos {
  int img, pix, coff;
};
__device__ __forceinline__ TokPos ln_tok(const LnArgs& a, int t) {
  TokPos p;
  const int tt = fdiv(t, a.fWo), ow = t - tt * a.Wo;
  p.img = fdiv(tt, a.fHo);
  const int oh = tt - p.img * a.Ho;
  p.coff = 0;
  if (a.mode == 0) {
    p.pix = t;
  } else if (a.mode == 1) {
    p.pix = (p.img * a.Hin + 2 * oh) * a.Win + 2 * ow;
  } else {
    const int h = fdiv(oh, a.fr), p1 = oh - h * a.r, w = fdiv(ow, a.fr), p2 = ow - w * a.r;
    p.pix = (p.img * a.Hin + h) * a.Win + w;
    p.coff = (p1 * a.r + p2) * a.C;
  }
  return p;
}
// per-lane constants of channel chunk c0: merge mode takes segment s = c0 / (C/4) from the pixel at
// (+ (s & 1) rows, + (s >> 1) columns), channels c0 - s C/4
__device__ __forceinline__ void ln_chunk(const LnArgs& a, int c0, int* dpix, int* cch) {
  *dpix = 0;
  *cch = c0;
  if (a.mode == 1) {
    const int Cq = a.C >> 2, sg = c0 / Cq;
    *dpix = (sg & 1) * a.Win + (sg >> 1);
    *cch = c0 - sg * Cq;
  }
}
__device__ __forceinline__ size_t ln_off(const TokPos& p, int dpix, int cch, int ld) {
  return (size_t)(p.pix + dpix) * ld + p.coff + cch;
}

// sum over the LPT consecutive lanes that share a token (LPT a power of two <= 64)
__device__ __forceinline__ float group_sum(float v, int lpt) {
  for (int o = lpt >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// LPT = min(64, pow2 >= C/VEC) lanes per token, 64/LPT tokens per wave, 4 waves per workgroup: a
// 96-channel bf16 token (12 chunks) occupies 16 lanes, not a whole wave.  A lane carries MAXIT chunks of
// each of U tokens; all their loads are issued before the first reduction and stay packed (16 bytes = 4
// registers) until they are used, so that a wave keeps U * MAXIT * 16 (x2 with a residual / in the
// backward) bytes per lane in flight at a register count that still admits 4+ waves per SIMD: with one
// token per pass and the register budget of MAXIT = 6 the kernel ran at 1 - 1.5 TB/s, bound by latency.
template <typename T> __device__ __forceinline__ void unpack_f(const uint4& r, float* f);
template <> __device__ __forceinline__ void unpack_f<float>(const uint4& r, float* f) {
  f[0] = __uint_as_float(r.x); f[1] = __uint_as_float(r.y); f[2] = __uint_as_float(r.z); f[3] = __uint_as_float(r.w);
}
template <> __device__ __forceinline__ void unpack_f<bf16_t>(const uint4& r, float* f) {
  const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f[2 * k] = __uint_as_float(w[k] << 16);
    f[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
  }
}
// keeps a packed load packed: the optimiser otherwise converts every loaded chunk to fp32 as soon as it lands
// (twice the registers), which costs the occupancy that the loads in flight were meant to buy
__device__ __forceinline__ void pin(uint4& r) { asm volatile("" : "+v"(r.x), "+v"(r.y), "+v"(r.z), "+v"(r.w)); }

// GELU (erf form, nn.GELU(): MixFFN_skip's act(norm1(.)), missformer.py:206) and its derivative.  erf by
// Abramowitz-Stegun 7.1.26 (|error| < 1.5e-7, below fp32 resolution of the products it enters): one v_exp, one
// v_rcp and five FMAs; its e^(-z^2/2) is also the density the derivative needs.  libdevice's erff costs ~3x that
// and made the fused LayerNorm kernels ALU-bound.
__device__ __forceinline__ void ln_gelu_parts(float z, float* cdf, float* pdf) {
  const float x = fabsf(z) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * x);
  const float e = __expf(-x * x);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float er = copysignf(1.f - poly * e, z);
  *cdf = 0.5f * (1.f + er);
  *pdf = 0.3989422804014327f * e;
}
__device__ __forceinline__ float ln_gelu(float z) {
  float c, p;
  ln_gelu_parts(z, &c, &p);
  return z * c;
}
__device__ __forceinline__ float ln_dgelu(float z) {
  float c, p;
  ln_gelu_parts(z, &c, &p);
  return c + z * p;
}

template <typename T, bool BWD, int MAXIT, int U, bool ACT>
__global__ __launch_bounds__(256) void layernorm_kernel(const LnArgs a, int lpt) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane & (lpt - 1), grp = lane / lpt, tpw = 64 / lpt;
  const int CC = a.C / VEC;
  const int P = a.N * a.Ho * a.Wo;
  const T* __restrict__ x = static_cast<const T*>(a.x);
  const T* __restrict__ second = static_cast<const T*>(BWD ? a.g : a.res);   // g, or the optional residual
  const int ld2 = BWD ? a.ldg : a.ldr;
  float gam[MAXIT][VEC], bet[MAXIT][VEC], ag[MAXIT][VEC], ab[MAXIT][VEC];
  int dpix[MAXIT], cch[MAXIT];
  // Every load of this kernel is UNCONDITIONAL: an out-of-range lane reads a valid dummy address and its value is replaced
  // by a select.  Loads inside `if (in range)` blocks made hipcc 7.2 close every block with s_waitcnt vmcnt(0): the three
  // chunks of a token (and the 48 gamma / beta words before them) arrived one memory round trip after the other, which
  // is what a launch on a 1.5 MB tensor spent its 10 us on (tools/ln_bench.py under rocprofv3: see DESIGN 3b).
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int cc = sub + lpt * it;
    ln_chunk(a, cc * VEC, &dpix[it], &cch[it]);
    const bool in = cc < CC;
    const int c0 = in ? cc * VEC : 0;
#pragma unroll
    for (int e = 0; e < VEC; e += 4) {
      const float4 gq = *reinterpret_cast<const float4*>(a.gamma + c0 + e);
      const float4 bq = (!BWD || ACT) ? *reinterpret_cast<const float4*>(a.beta + c0 + e) : make_float4(0.f, 0.f, 0.f, 0.f);
      gam[it][e] = in ? gq.x : 0.f, gam[it][e + 1] = in ? gq.y : 0.f, gam[it][e + 2] = in ? gq.z : 0.f, gam[it][e + 3] = in ? gq.w : 0.f;
      bet[it][e] = in ? bq.x : 0.f, bet[it][e + 1] = in ? bq.y : 0.f, bet[it][e + 2] = in ? bq.z : 0.f, bet[it][e + 3] = in ? bq.w : 0.f;
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) ag[it][e] = ab[it][e] = 0.f;
  }
  const float invC = 1.f / (float)a.C;
  const int tpb = 4 * tpw * U;  // tokens per workgroup pass
  for (int t0 = blockIdx.x * tpb; t0 < P; t0 += gridDim.x * tpb) {
    int t[U];
    TokPos pos[U];
    bool tok[U];    // whole lane groups go idle together; shuffles below stay inside a group
    uint4 xr[U][MAXIT], sr[U][MAXIT];
    float mean[U], rstd[U], fsc[U];
    const bool has2 = BWD || second != nullptr;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      t[u] = t0 + (u * 4 + wave) * tpw + grp;
      tok[u] = t[u] < P;
      const int tc = tok[u] ? t[u] : 0;
      pos[u] = ln_tok(a, tc);
#pragma unroll
      for (int it = 0; it < MAXIT; ++it) {
        const int cc = sub + lpt * it;
        const bool ok = cc < CC && tok[u], ok2 = ok && has2;
        const T* px = ok ? x + ln_off(pos[u], dpix[it], cch[it], a.ldx) : x;
        const T* ps = ok2 ? second + (size_t)tc * ld2 + cc * VEC : x;
        const uint4 rx = *reinterpret_cast<const uint4*>(px), rs = *reinterpret_cast<const uint4*>(ps);
        xr[u][it].x = ok ? rx.x : 0u, xr[u][it].y = ok ? rx.y : 0u, xr[u][it].z = ok ? rx.z : 0u, xr[u][it].w = ok ? rx.w : 0u;
        sr[u][it].x = ok2 ? rs.x : 0u, sr[u][it].y = ok2 ? rs.y : 0u, sr[u][it].z = ok2 ? rs.z : 0u, sr[u][it].w = ok2 ? rs.w : 0u;
      }
      if constexpr (BWD) {
        const float2 ms = *reinterpret_cast<const float2*>(a.stats + (size_t)tc * 2);
        mean[u] = tok[u] ? ms.x : 0.f;
        rstd[u] = tok[u] ? ms.y : 0.f;
      }
      // the per-image factor (stochastic depth) with the operands, not after the reductions
      const float* pf = a.sb != nullptr ? a.sb + pos[u].img : a.gamma;
      const float fv = *pf;
      fsc[u] = a.sb != nullptr ? fv : 1.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int it = 0; it < MAXIT; ++it) {
        pin(xr[u][it]);
        pin(sr[u][it]);
      }
    if constexpr (!BWD) {
      T* __restrict__ y = static_cast<T*>(a.y);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        __builtin_amdgcn_sched_barrier(0);   // one token at a time: the packed loads stay packed until here
        float v[MAXIT][VEC];
        float s = 0.f;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
          unpack_f<T>(xr[u][it], v[it]);
#pragma unroll
          for (int e = 0; e < VEC; ++e) s += v[it][e];
        }
        const float mu = group_sum(s, lpt) * invC;
        float q = 0.f;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it)
          if (sub + lpt * it < CC) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
              const float d = v[it][e] - mu;
              q += d * d;
            }
          }
        const float rs = rsqrtf(group_sum(q, lpt) * invC + a.eps);
        const float f = fsc[u];
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
          const int cc = sub + lpt * it;
          if (cc < CC && tok[u]) {
            float o[VEC], rv[VEC];
            unpack_f<T>(sr[u][it], rv);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
              o[e] = f * ((v[it][e] - mu) * rs * gam[it][e] + bet[it][e]);
              if (second != nullptr) o[e] += rv[e];
              if constexpr (ACT) o[e] = ln_gelu(o[e]);
            }
            store_f(y + (size_t)t[u] * a.ldy + cc * VEC, o);
          }
        }
        // (after the outputs: the block's join waits for the stores in front of it)
        if (sub == 0 && tok[u]) *reinterpret_cast<float2*>(a.stats + (size_t)t[u] * 2) = make_float2(mu, rs);
      }
    } else {
      T* __restrict__ dx = static_cast<T*>(a.dx);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        __builtin_amdgcn_sched_barrier(0);   // one token at a time: the packed loads stay packed until here
        const float f = fsc[u];
        float xh[MAXIT][VEC], gv[MAXIT][VEC];
        float s1 = 0.f, s2 = 0.f;  // sum of g*gamma, sum of g*gamma*xhat
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
          unpack_f<T>(xr[u][it], xh[it]);
          unpack_f<T>(sr[u][it], gv[it]);
          if (sub + lpt * it < CC && tok[u]) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
              gv[it][e] *= f;
              xh[it][e] = (xh[it][e] - mean[u]) * rstd[u];
              if constexpr (ACT) gv[it][e] *= ln_dgelu(xh[it][e] * gam[it][e] + bet[it][e]);
              ag[it][e] += gv[it][e] * xh[it][e];
              ab[it][e] += gv[it][e];
              const float gg = gv[it][e] * gam[it][e];
              s1 += gg;
              s2 += gg * xh[it][e];
            }
          }
        }
        s1 = group_sum(s1, lpt) * invC;
        s2 = group_sum(s2, lpt) * invC;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
          const int cc = sub + lpt * it;
          if (cc < CC && tok[u]) {
            float o[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = rstd[u] * (gv[it][e] * gam[it][e] - s1 - xh[it][e] * s2);
            store_f(dx + ln_off(pos[u], dpix[it], cch[it], a.lddx), o);
          }
        }
      }
    }
  }
  if constexpr (BWD) {
    // one partial row per workgroup: the 4 * tpw lane groups are summed through LDS in a fixed order
    extern __shared__ float red[];  // [4 * tpw][2][C]
    const int gidx = wave * tpw + grp, ngrp = 4 * tpw;
    float* mine = red + (size_t)gidx * 2 * a.C;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int cc = sub + lpt * it;
      if (cc < CC) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          mine[cc * VEC + e] = ag[it][e];
          mine[a.C + cc * VEC + e] = ab[it][e];
        }
      }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * a.C; c += 256) {
      float t = 0.f;
      for (int k = 0; k < ngrp; ++k) t += red[(size_t)k * 2 * a.C + c];
      a.partial[(size_t)blockIdx.x * 2 * a.C + c] = t;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm + 1x1 head: logits[img][k][oh][ow] = b[k] + sum_c w[k][c] LN(x)[token][c], the tail of
// swin_unet_v2 (FinalPatchExpand_X4's norm, :385, followed by the 1x1 `output` convolution, :690 / :753).
// At B=16 256x256 the normalised tensor is 201 MB: written by the LayerNorm, read by the head, written
// again as its gradient and read back by the LayerNorm backward.  Fused, the forward reads x once and the
// backward reads x and writes dx.  With wg[k][c] = w[k][c] gamma[c]:
//   logit_k = rstd * sum_c wg_kc (x_c - mean) + (sum_c w_kc beta_c + b_k)
//   d x     = rstd * (gg - mean_c(gg) - xhat mean_c(gg xhat)),   gg_c = sum_k dlogit_k wg_kc
// and all four parameter gradients follow from S_kc = sum_t dlogit_tk xhat_tc and D_k = sum_t dlogit_tk:
//   d gamma_c = sum_k w_kc S_kc   d beta_c = sum_k w_kc D_k   d w_kc = gamma_c S_kc + beta_c D_k   d b_k = D_k
// so a lane carries wg and S only (not gamma, beta, w and four accumulators).  Partial rows [K*C + K] per
// workgroup, finished by ln_head_finalize_kernel.  KT = 1 with three chunks per lane, or up to 4 classes with one.
// ---------------------------------------------------------------------------------------------
struct LnHeadArgs {
  LnArgs ln;            // x, dx, gamma, beta, stats, partial, N, Ho, Wo, C, ldx, lddx, mode, r, eps
  const float* w;       // [K][C]
  const float* b;       // [K] or null
  float* logits;        // fwd out  (N, K, Ho, Wo)
  const float* dlogits; // bwd in   (N, K, Ho, Wo)
  int K;
};

template <typename T, bool BWD, int KT, int MAXIT, int U>
__global__ __launch_bounds__(256) void ln_head_kernel(const LnHeadArgs h, int lpt) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const LnArgs& a = h.ln;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane & (lpt - 1), grp = lane / lpt, tpw = 64 / lpt;
  const int CC = a.C / VEC, K = h.K;
  const int P = a.N * a.Ho * a.Wo, HW = a.Ho * a.Wo;
  const T* __restrict__ x = static_cast<const T*>(a.x);
  float wg[KT][MAXIT][VEC], S[KT][MAXIT][VEC], D[KT], cst[KT], swg[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    float c0 = 0.f, c1 = 0.f;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int cc = sub + lpt * it;
      const bool in = cc < CC && k < K;
      // unconditional 16-byte loads (chunk 0 of class 0 for a lane out of range, then a select): as `in ? w[...] : 0` these
      // were 3 x 8 x 3 dword loads, each waited for before the next was issued (see layernorm_kernel)
      const int co = in ? cc * VEC : 0;
      const size_t wo = in ? (size_t)k * a.C + cc * VEC : 0;
#pragma unroll
      for (int e = 0; e < VEC; e += 4) {
        const float4 wq = *reinterpret_cast<const float4*>(h.w + wo + e);
        const float4 gq = *reinterpret_cast<const float4*>(a.gamma + co + e);
        const float4 bq = *reinterpret_cast<const float4*>(a.beta + co + e);
        const float wv[4] = {wq.x, wq.y, wq.z, wq.w}, gv[4] = {gq.x, gq.y, gq.z, gq.w}, bv[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          wg[k][it][e + q] = in ? wv[q] * gv[q] : 0.f;
          S[k][it][e + q] = 0.f;
          c0 += in ? wv[q] * bv[q] : 0.f;
          c1 += wg[k][it][e + q];
        }
      }
    }
    const float bk = h.b != nullptr ? h.b[k < K ? k : 0] : 0.f;
    cst[k] = group_sum(c0, lpt) + ((h.b != nullptr && k < K) ? bk : 0.f);
    swg[k] = group_sum(c1, lpt);
    D[k] = 0.f;
  }
  (void)swg;
  const float invC = 1.f / (float)a.C;
  const int tpb = 4 * tpw * U;
  for (int t0 = blockIdx.x * tpb; t0 < P; t0 += gridDim.x * tpb) {
    int t[U], rem[U];   // rem: the token's pixel inside its image (logits are NCHW planes)
    TokPos pos[U];
    bool tok[U];
    uint4 xr[U][MAXIT];
    float dl[U][KT], mean[U], rstd[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      t[u] = t0 + (u * 4 + wave) * tpw + grp;
      tok[u] = t[u] < P;
      const int tc = tok[u] ? t[u] : 0;
      pos[u] = ln_tok(a, tc);
      rem[u] = tc - pos[u].img * HW;
      // unconditional loads (a dummy address for lanes out of range, then a select): see layernorm_kernel
#pragma unroll
      for (int it = 0; it < MAXIT; ++it) {
        const int cc = sub + lpt * it;
        const bool ok = cc < CC && tok[u];
        const uint4 rx = *reinterpret_cast<const uint4*>(ok ? x + ln_off(pos[u], 0, cc * VEC, a.ldx) : x);
        xr[u][it].x = ok ? rx.x : 0u, xr[u][it].y = ok ? rx.y : 0u, xr[u][it].z = ok ? rx.z : 0u, xr[u][it].w = ok ? rx.w : 0u;
      }
      if constexpr (BWD) {
        const float2 ms = *reinterpret_cast<const float2*>(a.stats + (size_t)tc * 2);
        mean[u] = tok[u] ? ms.x : 0.f;
        rstd[u] = tok[u] ? ms.y : 0.f;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          const bool okk = tok[u] && k < K;
          const float dv = h.dlogits[okk ? ((size_t)pos[u].img * K + k) * HW + rem[u] : 0];
          dl[u][k] = okk ? dv : 0.f;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int it = 0; it < MAXIT; ++it) pin(xr[u][it]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      __builtin_amdgcn_sched_barrier(0);
      float v[MAXIT][VEC];
#pragma unroll
      for (int it = 0; it < MAXIT; ++it) unpack_f<T>(xr[u][it], v[it]);
      if constexpr (!BWD) {
        float s = 0.f;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it)
#pragma unroll
          for (int e = 0; e < VEC; ++e) s += v[it][e];
        const float mu = group_sum(s, lpt) * invC;
        float q = 0.f, lk[KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) lk[k] = 0.f;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it)
          if (sub + lpt * it < CC) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
              const float d = v[it][e] - mu;
              q = fmaf(d, d, q);
#pragma unroll
              for (int k = 0; k < KT; ++k) lk[k] = fmaf(d, wg[k][it][e], lk[k]);
            }
          }
        const float rs = rsqrtf(group_sum(q, lpt) * invC + a.eps);
        if (sub == 0 && tok[u]) {
          a.stats[(size_t)t[u] * 2] = mu;
          a.stats[(size_t)t[u] * 2 + 1] = rs;
        }
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          const float tot = group_sum(lk[k], lpt);
          if (sub == 0 && tok[u] && k < K) h.logits[((size_t)pos[u].img * K + k) * HW + rem[u]] = fmaf(rs, tot, cst[k]);
        }
      } else {
        T* __restrict__ dx = static_cast<T*>(a.dx);
        float gg[MAXIT][VEC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
          const bool in = sub + lpt * it < CC && tok[u];
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const float xh = in ? (v[it][e] - mean[u]) * rstd[u] : 0.f;
            v[it][e] = xh;
            float g = 0.f;
#pragma unroll
            for (int k = 0; k < KT; ++k) {
              g = fmaf(dl[u][k], wg[k][it][e], g);
              S[k][it][e] = fmaf(dl[u][k], xh, S[k][it][e]);
            }
            gg[it][e] = g;
            s1 += g;
            s2 = fmaf(g, xh, s2);
          }
        }
#pragma unroll
        for (int k = 0; k < KT; ++k) D[k] += dl[u][k];
        s1 = group_sum(s1, lpt) * invC;
        s2 = group_sum(s2, lpt) * invC;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
          const int cc = sub + lpt * it;
          if (cc < CC && tok[u]) {
            float o[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = rstd[u] * (gg[it][e] - s1 - v[it][e] * s2);
            store_f(dx + ln_off(pos[u], 0, cc * VEC, a.lddx), o);
          }
        }
      }
    }
  }
  if constexpr (BWD) {
    // one partial row per workgroup: [S (K*C) | D (K)]
    extern __shared__ float red[];  // [4 * tpw][K * C + K]
    const int gidx = wave * tpw + grp, ngrp = 4 * tpw;
    const int n = K * a.C + K;
    float* mine = red + (size_t)gidx * n;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int cc = sub + lpt * it;
      if (cc < CC) {
#pragma unroll
        for (int e = 0; e < VEC; ++e)
#pragma unroll
          for (int k = 0; k < KT; ++k)
            if (k < K) mine[k * a.C + cc * VEC + e] = S[k][it][e];
      }
    }
    if (sub == 0) {
#pragma unroll
      for (int k = 0; k < KT; ++k)
        if (k < K) mine[K * a.C + k] = D[k];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < n; c += 256) {
      float t = 0.f;
      for (int k = 0; k < ngrp; ++k) t += red[(size_t)k * n + c];
      a.partial[(size_t)blockIdx.x * n + c] = t;
    }
  }
}

// grid (C / 8), 1024 threads = 8 channels x 128 row groups (the rows are many -- one per workgroup of the main
// kernel -- and the columns few); sums the rows (in double, fixed order) and forms the four gradients from S
// and D (see the head comment)
__global__ __launch_bounds__(1024) void ln_head_finalize_kernel(const float* __restrict__ partial, int rows, int C, int K,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const float* __restrict__ w, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, float* __restrict__ dw,
                                                                float* __restrict__ db) {
  __shared__ double sh[128][9];
  __shared__ double sD[4];
  const int el = threadIdx.x & 7, g = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + el, n = K * C + K;
  double Skc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int k = 0; k <= K; ++k) {   // k == K: the D columns (el < K)
    const int col = k < K ? k * C + c : K * C + el;
    const bool in = k < K ? c < C : el < K;
    double s = 0.0;
    // eight rows per trip, unconditional loads (row 0 / column 0 out of range, dropped below), added in the same order as one row
    // per trip: this loop was rows / 128 dependent round trips per pass
    for (int r = g; r < rows; r += 8 * 128) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(r + u * 128 < rows ? r + u * 128 : 0) * n + (in ? col : 0)];
#pragma unroll
      for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(v[u]));
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (in && r + u * 128 < rows) s += (double)v[u];
    }
    __syncthreads();
    sh[g][el] = s;
    __syncthreads();
    if (g == 0) {
      double t = 0.0;
      for (int r = 0; r < 128; ++r) t += sh[r][el];
      if (k < K) Skc[k] = t;
      else if (el < K) sD[el] = t;
    }
  }
  __syncthreads();
  if (g == 0 && c < C) {
    double dg = 0.0, dbt = 0.0;
    for (int k = 0; k < K; ++k) {
      const double wv = (double)w[(size_t)k * C + c];
      dg += wv * Skc[k];
      dbt += wv * sD[k];
      dw[(size_t)k * C + c] = (float)((double)gamma[c] * Skc[k] + (double)beta[c] * sD[k]);
    }
    dgamma[c] = (float)dg;
    dbeta[c] = (float)dbt;
  }
  if (blockIdx.x == 0 && g == 0 && el < K && db != nullptr) db[el] = (float)sD[el];
}

// ---------------------------------------------------------------------------------------------
// Window attention.
// ---------------------------------------------------------------------------------------------
struct AttnArgs {
  const void* qkv;    // [P][3C]: per token [3][heads][32]
  void* out;          // [P][C]   (bwd: the forward output, read)
  float* lse;         // [B*nW][heads][N] row log-sum-exp
  const float* tau;   // [heads][Nt][Nt] (Nt = window_size^2 of the parameter, N <= Nt used)
  const float* bias;  // [heads][N][N]
  const void* dout;   // bwd: gradient of out [P][C]
  void* dqkv;         // bwd: gradient of qkv [P][3C]
  float* partial;     // bwd: [gridDim.x][2][heads][N][N] sums of dS (dbias) and d(tau)
  int B, H, W, C, heads, ws, shift, Nt;
  int ldq, ldo, lddo, lddq;
  float scale;
  int flags;          // ablation build only (UZ_KFLAGS)
};

constexpr int AD = 32;       // head dimension (embed_dim 96 / 3 heads, doubled together: always 32)
constexpr int AN = 64;       // max tokens per window (window_size <= 8)
constexpr int ARS = AD + 4;  // LDS row stride of the [token][32] tiles: rows stay 16-byte aligned, so a row
                             // (read by all lanes at once = broadcast) costs 8 ds_read_b128, not 32 ds_read_b32
constexpr int ANS = AN + 1;  // LDS row stride of the [N][N] matrices
constexpr int AJ = AN / 4;   // keys (forward, backward phase A) per wave: the four waves split the other index

__device__ __forceinline__ void lds_row(const float* row, float* f) {  // 32 floats, 16-byte aligned
#pragma unroll
  for (int c = 0; c < AD / 4; ++c) {
    const float4 v = reinterpret_cast<const float4*>(row)[c];
    f[4 * c] = v.x;
    f[4 * c + 1] = v.y;
    f[4 * c + 2] = v.z;
    f[4 * c + 3] = v.w;
  }
}
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }  // v_rcp_f32, 1 ulp

// 32-wide fp32 vector helpers written on float pairs so that they compile to v_pk_fma_f32 / v_pk_mul_f32
// (two fp32 operations per lane and instruction): the attention kernels are VALU-bound
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ __forceinline__ float dot32(const float* a, const float* b) {
  f32x2 acc = {0.f, 0.f};
#pragma unroll
  for (int e = 0; e < AD; e += 2) {
    const f32x2 x = {a[e], a[e + 1]}, y = {b[e], b[e + 1]};
    acc = __builtin_elementwise_fma(x, y, acc);
  }
  return acc.x + acc.y;
}
__device__ __forceinline__ void axpy32(float w, const float* x, float* y) {  // y += w * x
  const f32x2 ws = {w, w};
#pragma unroll
  for (int e = 0; e < AD; e += 2) {
    const f32x2 xv = {x[e], x[e + 1]}, yv = {y[e], y[e + 1]};
    const f32x2 r = __builtin_elementwise_fma(ws, xv, yv);
    y[e] = r.x;
    y[e + 1] = r.y;
  }
}
__device__ __forceinline__ void scale_axpy32(float c, float w, const float* x, float* y) {  // y = c * y + w * x
  const f32x2 cs = {c, c}, ws = {w, w};
#pragma unroll
  for (int e = 0; e < AD; e += 2) {
    const f32x2 xv = {x[e], x[e + 1]}, yv = {y[e], y[e + 1]};
    const f32x2 r = __builtin_elementwise_fma(ws, xv, cs * yv);
    y[e] = r.x;
    y[e + 1] = r.y;
  }
}

struct WinTok {
  int tok;   // row of the token tensor
  int cnt;   // region id of the shifted-window mask
};
__device__ __forceinline__ WinTok win_token(const AttnArgs& a, int win, int i) {
  const int nwx = a.W / a.ws, nwy = a.H / a.ws, nW = nwx * nwy;
  const int b = win / nW, wi = win - b * nW, wy = wi / nwx, wx = wi - wy * nwx;
  const int iy = i / a.ws, ix = i - iy * a.ws;
  const int hs = wy * a.ws + iy, wsx = wx * a.ws + ix;  // coordinates in the rolled image
  int h = hs + a.shift, w = wsx + a.shift;
  if (h >= a.H) h -= a.H;
  if (w >= a.W) w -= a.W;
  WinTok t;
  t.tok = (b * a.H + h) * a.W + w;
  const int hid = hs < a.H - a.ws ? 0 : (hs < a.H - a.shift ? 1 : 2);
  const int wid = wsx < a.W - a.ws ? 0 : (wsx < a.W - a.shift ? 1 : 2);
  t.cnt = a.shift > 0 ? hid * 3 + wid : 0;
  return t;
}

template <typename T> __device__ __forceinline__ void load_head(const T* p, float* f) {  // 32 values
  constexpr int VEC = ElemTraits<T>::VEC;
#pragma unroll
  for (int c = 0; c < AD / VEC; ++c) load_f(p + c * VEC, f + c * VEC);
}
template <typename T> __device__ __forceinline__ void store_head(T* p, const float* f) {
  constexpr int VEC = ElemTraits<T>::VEC;
#pragma unroll
  for (int c = 0; c < AD / VEC; ++c) store_f(p + c * VEC, f + c * VEC);
}

template <typename T> __device__ __forceinline__ void store8(T* p, const float* f) {  // 8 consecutive values
  constexpr int VEC = ElemTraits<T>::VEC;
#pragma unroll
  for (int c = 0; c < 8 / VEC; ++c) store_f(p + c * VEC, f + c * VEC);
}

// Forward: one 256-thread workgroup (one wave per SIMD) per (window, head); lane = query i, the four
// waves split the key range, each with its own running (max, sum, output); the partial states are
// merged through LDS in a fixed order.
template <typename T>
__global__ __launch_bounds__(256) void winattn_fwd_kernel(const AttnArgs a) {
  constexpr int PS = AD + 3;  // partial row: 32 outputs, max, sum (+1 pad)
  __shared__ float sK[AN * ARS], sV[AN * ARS], sKn[AN], sPart[4 * AN * PS];
  __shared__ int sCnt[AN];
  const int tid = threadIdx.x, w = tid >> 6, i = tid & 63, h = blockIdx.y;
  const int N = a.ws * a.ws;
  const int jc = (N + 3) >> 2, lo = w * jc, hi = min(N, lo + jc);
  const int nWin = a.B * (a.H / a.ws) * (a.W / a.ws);
  const T* __restrict__ qkv = static_cast<const T*>(a.qkv);
  T* __restrict__ out = static_cast<T*>(a.out);
  // 1/clip(tau) and bias of this thread's (query, key range): fixed for the head, kept in registers
  float ti[AJ], bi[AJ];
#pragma unroll
  for (int jj = 0; jj < AJ; ++jj) {
    const int j = lo + jj;
    const bool ok = i < N && j < hi;
    ti[jj] = ok ? 1.f / fmaxf(a.tau[((size_t)h * a.Nt + i) * a.Nt + j], 0.01f) : 0.f;
    bi[jj] = ok ? a.bias[((size_t)h * N + i) * N + j] : 0.f;
  }
  for (int win = blockIdx.x; win < nWin; win += gridDim.x) {
    __syncthreads();  // previous window's readers are done
    float q[AD];
    float qn = 0.f;
    WinTok me = {0, 0};
    if (i < N) {
      me = win_token(a, win, i);
      const T* row = qkv + (size_t)me.tok * a.ldq + h * AD;
      load_head(row, q);
#pragma unroll
      for (int e = 0; e < AD; ++e) {
        q[e] *= a.scale;
        qn += q[e] * q[e];
      }
      qn = sqrtf(qn);
      if (w == 0) {
        float kv[AD];
        load_head(row + a.C, kv);
        float kn = 0.f;
#pragma unroll
        for (int e = 0; e < AD; ++e) {
          kn += kv[e] * kv[e];
          sK[i * ARS + e] = kv[e];
        }
        sKn[i] = sqrtf(kn);
        sCnt[i] = me.cnt;
      } else if (w == 1) {
        float kv[AD];
        load_head(row + 2 * a.C, kv);
#pragma unroll
        for (int e = 0; e < AD; ++e) sV[i * ARS + e] = kv[e];
      }
    }
    __syncthreads();
    {
      float m = -INFINITY, l = 0.f, o[AD];
#pragma unroll
      for (int e = 0; e < AD; ++e) o[e] = 0.f;
      if (i < N) {
#pragma unroll
        for (int jj = 0; jj < AJ; ++jj) {
          const int j = lo + jj;
          if (j < hi) {
            float row[AD];
            lds_row(sK + j * ARS, row);
            const float u = dot32(q, row);
            float s = u * rcp(fmaxf(qn * sKn[j], 1e-6f)) * ti[jj] + bi[jj];
            if (sCnt[j] != me.cnt) s -= 100.f;
            const float mn = fmaxf(m, s);
            const float corr = __expf(m - mn), p = __expf(s - mn);
            l = l * corr + p;
            lds_row(sV + j * ARS, row);
            scale_axpy32(corr, p, row, o);
            m = mn;
          }
        }
      }
      float* pr = sPart + (w * AN + i) * PS;
#pragma unroll
      for (int e = 0; e < AD; ++e) pr[e] = o[e];
      pr[AD] = m;
      pr[AD + 1] = l;
    }
    __syncthreads();
    if (i < N) {  // merge the four partial states; wave w writes output components [8w, 8w + 8)
      float m = -INFINITY;
#pragma unroll
      for (int k = 0; k < 4; ++k) m = fmaxf(m, sPart[(k * AN + i) * PS + AD]);
      float l = 0.f, o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float* pr = sPart + (k * AN + i) * PS;
        const float f = pr[AD + 1] > 0.f ? __expf(pr[AD] - m) : 0.f;  // a wave with an empty key range has l = 0
        l = fmaf(pr[AD + 1], f, l);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = fmaf(pr[8 * w + e], f, o[e]);
      }
      const float inv = 1.f / l;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] *= inv;
      store8(out + (size_t)me.tok * a.ldo + h * AD + 8 * w, o);
      if (w == 0) a.lse[((size_t)win * a.heads + h) * N + i] = m + __logf(l);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// bf16 forward on the matrix cores.  One WAVE per (window, head) (four independent waves per workgroup):
//   S^T = K Q^T   v_mfma_f32_32x32x16_bf16, K rows / Q rows straight from global memory as the A / B
//                 fragments (16 bytes of one token's head slice per lane) -> accumulator column = query
//                 (lane & 31), rows = keys: a lane owns, for ITS query, 16 keys per 32-key tile, so the
//                 softmax runs over registers (+ one xor-32 shuffle), no LDS round trip;
//   O^T = V^T P^T the exponentials are packed to bf16 in registers and are directly the B fragment (the
//                 accumulator's key order 32kt + 16s + 8(e>>2) + 4(lane>>5) + (e&3) is used as the K order
//                 of both operands); V^T comes from a per-wave LDS tile [32 d][64 keys].
// tau / bias of the lane's (query, key) pairs are fixed for the head and live in registers.
// ---------------------------------------------------------------------------------------------
constexpr int VTS = 72;  // V^T row stride in bf16 elements (144 B: 8-byte aligned rows, spreads banks)

__global__ __launch_bounds__(256) void winattn_fwd_mfma_kernel(const AttnArgs a) {
  __shared__ __attribute__((aligned(16))) bf16_t sVT[4][AD * VTS];
  __shared__ float sKn[4][AN];
  __shared__ int sCnt[4][AN];
  __shared__ float sTab[2][AN * ANS];   // this head's 1/clip(tau) and bias, staged once per workgroup (coalesced)
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, l31 = lane & 31, lh = lane >> 5, h = blockIdx.y;
  const int N = a.ws * a.ws;
  const int nWin = a.B * (a.H / a.ws) * (a.W / a.ws);
  const bf16_t* __restrict__ qkv = static_cast<const bf16_t*>(a.qkv);
  bf16_t* __restrict__ out = static_cast<bf16_t*>(a.out);
  bf16_t* vt = sVT[w];
  for (int e = tid; e < N * N; e += 256) {
    const int r = e / N, c = e - r * N;
    sTab[0][r * ANS + c] = 1.f / fmaxf(a.tau[((size_t)h * a.Nt + r) * a.Nt + c], 0.01f);
    sTab[1][r * ANS + c] = a.bias[((size_t)h * N + r) * N + c];
  }
  __syncthreads();
  // tables of this lane's (query 32 qt + l31, key 32 kt + row(r)) pairs, in registers
  float ti[2][2][16], bi[2][2][16];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = 32 * qt + l31, j = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const bool ok = i < N && j < N;
        ti[qt][kt][r] = ok ? sTab[0][i * ANS + j] : 0.f;
        bi[qt][kt][r] = ok ? sTab[1][i * ANS + j] : 0.f;
      }
  for (int win = blockIdx.x * 4 + w; win < nWin; win += gridDim.x * 4) {
    // token t = 32 x + l31 (x = 0, 1) in its query role (B fragment column) and key role (A fragment row)
    WinTok tk[2];
    bf16x8 qf[2][2], kf[2][2];
    float qn[2], kn[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const int t = 32 * x + l31;
      float q2 = 0.f, k2 = 0.f;
      if (t < N) {
        tk[x] = win_token(a, win, t);
        const bf16_t* row = qkv + (size_t)tk[x].tok * a.ldq + h * AD + 8 * lh;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          qf[x][ks] = *reinterpret_cast<const bf16x8*>(row + 16 * ks);
          kf[x][ks] = *reinterpret_cast<const bf16x8*>(row + a.C + 16 * ks);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float qv = (float)qf[x][ks][e], kv = (float)kf[x][ks][e];
            q2 = fmaf(qv, qv, q2);
            k2 = fmaf(kv, kv, k2);
          }
        }
      } else {
        tk[x].tok = 0;
        tk[x].cnt = -1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int e = 0; e < 8; ++e) qf[x][ks][e] = kf[x][ks][e] = (bf16_t)0.f;
      }
      q2 += __shfl_xor(q2, 32);
      k2 += __shfl_xor(k2, 32);
      qn[x] = a.scale * sqrtf(q2);
      kn[x] = sqrtf(k2);
      if (lh == 0) {
        sKn[w][t] = kn[x];
        sCnt[w][t] = tk[x].cnt;
      }
    }
    {  // V^T tile: lane = token (32 lh + l31 == lane), its 32 values scattered down the column
      const int t = lane;
      const WinTok me = tk[lh];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        bf16x8 v;
        if (t < N) v = *reinterpret_cast<const bf16x8*>(qkv + (size_t)me.tok * a.ldq + 2 * a.C + h * AD + 8 * c);
        else
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (bf16_t)0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) vt[(8 * c + e) * VTS + t] = v[e];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's LDS writes are visible to its reads
    // S^T tiles
    f32x16 st[2][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st[kt][qt][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          st[kt][qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kt][ks], qf[qt][ks], st[kt][qt], 0, 0, 0);
      }
    f32x16 ot[2];
    float lsum[2], mrow[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      // scores of this lane's query against its 32 keys, then the row maximum
      float m = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * lh;
          float sv = st[kt][qt][r] * a.scale * rcp(fmaxf(qn[qt] * sKn[w][j], 1e-6f)) * ti[qt][kt][r] + bi[qt][kt][r];
          if (sCnt[w][j] != tk[qt].cnt) sv -= 100.f;
          if (j >= N) sv = -INFINITY;
          st[kt][qt][r] = sv;
          m = fmaxf(m, sv);
        }
      m = fmaxf(m, __shfl_xor(m, 32));
      if (m == -INFINITY) m = 0.f;  // padded query column
      float l = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[qt][r] = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          bf16x8 pf;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float p = __expf(st[kt][qt][8 * s2 + e] - m);
            const bf16_t pb = (bf16_t)p;
            l += (float)pb;   // normalise by what is actually multiplied
            pf[e] = pb;
          }
          // A fragment = V^T rows (d = l31), keys 32 kt + 16 s2 + 4 lh + {0..3} and + 8
          const bf16_t* vr = vt + l31 * VTS + 32 * kt + 16 * s2 + 4 * lh;
          const bf16x4 lo4 = *reinterpret_cast<const bf16x4*>(vr), hi4 = *reinterpret_cast<const bf16x4*>(vr + 8);
          const bf16x8 vf = __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
          ot[qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, ot[qt], 0, 0, 0);
        }
      l += __shfl_xor(l, 32);
      lsum[qt] = l;
      mrow[qt] = m;
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int i = 32 * qt + l31;
      if (i < N) {
        const float inv = 1.f / lsum[qt];
        bf16_t* orow = out + (size_t)tk[qt].tok * a.ldo + h * AD + 4 * lh;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          bf16x4 o4;
#pragma unroll
          for (int e = 0; e < 4; ++e) o4[e] = (bf16_t)(ot[qt][4 * q4 + e] * inv);
          *reinterpret_cast<bf16x4*>(orow + 8 * q4) = o4;
        }
        if (lh == 0) a.lse[((size_t)win * a.heads + h) * N + i] = mrow[qt] + __logf(lsum[qt]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // LDS reads done before the next window overwrites the tiles
  }
}

// ---------------------------------------------------------------------------------------------
// bf16 backward on the matrix cores.
// Pass 1 (accumulator column = query i, rows = keys): U^T = K Q^T and dP^T = V dO^T by MFMA, then per
// element P, dS, d(bias) / d(tau) sums (registers, kept over the wave's windows) and W1 = dS/(tau den);
// dQ^T = K^T W1^T with the bf16-packed W1 registers as the B fragment.  Pass 2 (column = key j, rows =
// queries): U = Q K^T, dP = dO V^T, the same element math, dV^T = dO^T P and dK^T = Q^T W1.  The
// transposed operands (K^T, dO^T, Q^T: [32 d][64 tokens]) are per-wave LDS tiles.
// (A first version with one wave per (window, head) needed 128 running-sum registers per lane, spilled and
// was slower than the VALU kernel; see the block decomposition inside the kernel.)
// ---------------------------------------------------------------------------------------------
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for vmcnt(0): every global load in flight (the
// next window's prefetch) and every global store (the window's results, acknowledged ~1 us after issue) -- four such drains
// per window were more than half of the attention kernels' wave time (SQ_WAIT_ANY / SQ_WAVE_CYCLES = 0.55).  The kernels
// below exchange data between waves through LDS only.
// Contract of every call site (round-4 review): (1) nothing a wave wrote to GLOBAL memory is read by another wave of the
// workgroup afterwards -- results leave through each wave's own stores, the prefetch loads land in registers of the wave that
// issued them -- so no vmcnt wait is owed; (2) every call sits in workgroup-uniform control flow (the window loop's trip count
// and the pass structure depend on blockIdx and kernel arguments only), as s_barrier requires; `asm volatile` with a memory
// clobber is not moved across other memory accesses or into a branch by the compiler.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ bf16x8 pack8(const float* f) {
  bf16x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (bf16_t)f[e];
  return v;
}
// A fragment of a transposed tile XT[d][token]: row d = l31, the 8 K-slots = tokens 32 t + 16 s + 4 lh + {0..3}, + 8
__device__ __forceinline__ bf16x8 tfrag(const bf16_t* xt, int l31, int lh, int t, int s2) {
  const bf16_t* p = xt + l31 * VTS + 32 * t + 16 * s2 + 4 * lh;
  const bf16x4 lo4 = *reinterpret_cast<const bf16x4*>(p), hi4 = *reinterpret_cast<const bf16x4*>(p + 8);
  return __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
}

__global__ __launch_bounds__(256, 2) void winattn_bwd_mfma_kernel(const AttnArgs a) {
  // One workgroup per (window, head); wave w owns the 32 x 32 block (query tile qt = w >> 1, key tile kt = w & 1)
  // of the score matrix, so a lane carries 16 + 16 running sums instead of 128 and the element code exists once.
  // The element pass runs with column = query (everything per query is the lane's own); it leaves P and
  // W = dC / (|q||k|) of its block in LDS as bf16 [key][query], which is the B operand of the dk / dv products,
  // so there is no second element pass.  The projection terms of the two normalisations come from the products
  // themselves: with A_i = sum_j W_ij k_j, sum_j dC_ij c_ij = q_i . A_i (likewise for k), a 32-term dot in
  // the epilogue instead of six operations per score element.  The reference clamps the norm product at 1e-6
  // (swin_unet_v2.py:137-139): a clamped pair keeps u / 1e-6 and has NO projection term.  Round 5 follows that to the
  // letter: a wave whose queries could reach the clamp against this key tile (|scale q| * min |k| < 1e-6, a test of one
  // multiply per window) takes a slow path that sums W_ij u_ij over its clamped pairs -- per query in a register, per key
  // by shuffles into sCorrK -- and the epilogues subtract those sums from q_i . A_i / k_j . B_j.  (Rounds 1-4 kept the
  // projection term for 0 < |q||k| <= 1e-6 and were exact for zero rows only.)
  // Partial dq (over kt) and dk / dv (over qt) of the two waves that share a tile meet in LDS.
  __shared__ __attribute__((aligned(16))) bf16_t sKT[AD * VTS], sGT[AD * VTS], sQT[AD * VTS];
  __shared__ __attribute__((aligned(16))) bf16_t sPW[2 * AN * VTS];   // P, W [key][query]; later the dk / dv hand-over
  __shared__ float sRedQ[2][64 * 17];   // dq hand-over of the kt = 1 waves, [lane][16] (+1 pad)
  __shared__ __attribute__((aligned(16))) float sRk[AN];   // 1 / |k_j|
  __shared__ __attribute__((aligned(16))) int sCnt[AN];
  __shared__ float sRkMax[2];        // per key tile: max 1 / |k_j| over its non-zero keys
  __shared__ float sCorrK[2][AN];    // [query tile][key]: sum over the tile's CLAMPED pairs of W_ij u_ij (zero in the fast path)
  // 1 / clip(tau) (negated where the clip is active) and the bias of this head in LANE ORDER: a lane's 16 score elements are
  // the same (query, key) pairs in every window, [table][g4][thread] holds its four values of key group g4 as one 16-byte
  // read that is conflict-free across the wave (the [query][key] table it replaces cost 32 ds_read_b32 per window, each
  // waited for where it was used)
  __shared__ float4 sTabL[2][4][256];
  static_assert(2 * 2 * 64 * 17 * sizeof(float) <= sizeof(bf16_t) * 2 * AN * VTS, "dk / dv hand-over must fit in sPW");
  bf16_t* sP = sPW;
  bf16_t* sW = sPW + AN * VTS;
  float* sRedK = reinterpret_cast<float*>(sPW);   // [2 key tiles][2 (dk, dv)][64 * 17]
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, l31 = lane & 31, lh = lane >> 5, h = blockIdx.y;
  const int qt = w >> 1, kt = w & 1;
  const int N = a.ws * a.ws;
  const int nWin = a.B * (a.H / a.ws) * (a.W / a.ws);
  const bool masked = a.shift > 0;
  const bf16_t* __restrict__ qkv = static_cast<const bf16_t*>(a.qkv);
  const bf16_t* __restrict__ out = static_cast<const bf16_t*>(a.out);
  const bf16_t* __restrict__ dout = static_cast<const bf16_t*>(a.dout);
  bf16_t* __restrict__ dqkv = static_cast<bf16_t*>(a.dqkv);
  const int iq = 32 * qt + l31, jk = 32 * kt + l31;   // this lane's query (column role) / key (column role)
  // running sums over this workgroup's windows of dS (-> d bias) and dS * c (-> d tau) for
  // (query 32 qt + l31, key 32 kt + 4 lh + (r & 3) + 8 (r >> 2))
  f32x2 accb2[8], acct2[8];   // element pairs (r, r + 1)
#pragma unroll
  for (int r = 0; r < 8; ++r) accb2[r] = acct2[r] = (f32x2){0.f, 0.f};
  // the two tables: coalesced rows from global memory into a [query][key] staging tile (the P / W area), then each lane
  // gathers its own 16 entries into the lane-order table
  float* stage = reinterpret_cast<float*>(sPW);   // [AN][ANS] floats
  static_assert(AN * ANS * sizeof(float) <= sizeof(bf16_t) * 2 * AN * VTS, "the staging tile must fit in sPW");
  {
    // all 32 loads of a thread in flight at once (as a loop of load -> divide -> store they were 16 serialized memory round
    // trips, 8 us of a kernel that spends 4 us per window)
    float tv[16], bv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int e = tid + 256 * i, r = e >> 6, c = e & 63;
      const bool in = r < N && c < N && !(UZ_KFLAGS(a) & 0x10000);
      // (unconditional loads -- entry (0, 0) for a lane outside the window -- then a select: under `in ? ... :` every load
      // was waited for before the next was issued, the "16 serialized round trips" again)
      tv[i] = a.tau[in ? ((size_t)h * a.Nt + r) * a.Nt + c : 0];
      bv[i] = a.bias[in ? ((size_t)h * N + r) * N + c : 0];
    }
    // (the loaded values are made opaque before the selects: a value that is only used when `in` holds is otherwise turned back
    // into a load under a branch)
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(tv[i]), "+v"(bv[i]));
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int e = tid + 256 * i, r = e >> 6, c = e & 63;
      const bool in = r < N && c < N && !(UZ_KFLAGS(a) & 0x10000);
      tv[i] = in ? tv[i] : 1.f;
      bv[i] = in ? bv[i] : ((r < N && c < N) ? 0.f : -1e30f);   // padding: exp() = 0, no test
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int e = tid + 256 * i, r = e >> 6, c = e & 63;
        const float inv = 1.f / fmaxf(tv[i], 0.01f);
        stage[r * ANS + c] = t == 0 ? (tv[i] >= 0.01f ? inv : -inv) : bv[i];   // negated where the clip is active
      }
      __syncthreads();
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float* sp = stage + iq * ANS + 32 * kt + 8 * g4 + 4 * lh;
        sTabL[t][g4][tid] = make_float4(sp[0], sp[1], sp[2], sp[3]);
      }
      __syncthreads();
    }
  }

  // A window's q, dO, O, k, v fragments are fetched one window ahead: the loads are issued right after the
  // score products have consumed the current ones and land during the element pass.
  WinTok ntq = {0, -1}, ntk = {0, -1};
  bf16x8 nq[2], ng[2], no[2], nk[2], nv[2];
  float nlse = 0.f;
  // (The forward kernel's fetch is unconditional, DESIGN 3h.  Here that form measured SLOWER -- 52.8 -> 58.9 us on the 64 x 64
  // token map -- the loop's top then waits with the previous window's dq / dk / dv stores behind the prefetched loads; the
  // conditional form waits for its loads right here, before those stores are issued, and the second resident workgroup of the
  // CU covers the round trip.)
  auto fetch = [&](int win) {
    ntq = {0, -1};
    ntk = {0, -1};
    nlse = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) nq[ks][e] = ng[ks][e] = no[ks][e] = nk[ks][e] = nv[ks][e] = (bf16_t)0.f;
    if (iq < N) {
      ntq = win_token(a, win, iq);
      const bf16_t* row = qkv + (size_t)ntq.tok * a.ldq + h * AD + 8 * lh;
      const bf16_t* grow = dout + (size_t)ntq.tok * a.lddo + h * AD + 8 * lh;
      const bf16_t* orow = out + (size_t)ntq.tok * a.ldo + h * AD + 8 * lh;
      nlse = a.lse[((size_t)win * a.heads + h) * N + iq];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        nq[ks] = *reinterpret_cast<const bf16x8*>(row + 16 * ks);
        ng[ks] = *reinterpret_cast<const bf16x8*>(grow + 16 * ks);
        no[ks] = *reinterpret_cast<const bf16x8*>(orow + 16 * ks);
      }
    }
    if (jk < N) {
      ntk = win_token(a, win, jk);
      const bf16_t* row = qkv + (size_t)ntk.tok * a.ldq + a.C + h * AD + 8 * lh;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        nk[ks] = *reinterpret_cast<const bf16x8*>(row + 16 * ks);
        nv[ks] = *reinterpret_cast<const bf16x8*>(row + a.C + 16 * ks);
      }
    }
  };
  if ((int)blockIdx.x < nWin) fetch(blockIdx.x);

  for (int win = blockIdx.x; win < nWin; win += gridDim.x) {
    if (UZ_KFLAGS(a) & 0x40000) break;
    lds_barrier();   // previous window: every reader of the tiles / hand-over areas is done (and sTab has landed)
    const WinTok tq = ntq, tkk = ntk;
    bf16x8 qf[2], gf[2], kf[2], vf[2];
    float rq = 0.f, Di = 0.f;
    const float lse = nlse;
    {
      float q2 = 0.f, k2 = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        qf[ks] = nq[ks];
        gf[ks] = ng[ks];
        kf[ks] = nk[ks];
        vf[ks] = nv[ks];
        // |q|^2, |k|^2 and dO . O on bf16 pairs (v_dot2c_f32_bf16: fp32 products and sums), 12 instructions instead of ~130
        const bf16x2* qp = reinterpret_cast<const bf16x2*>(&qf[ks]);
        const bf16x2* kp = reinterpret_cast<const bf16x2*>(&kf[ks]);
        const bf16x2* gp = reinterpret_cast<const bf16x2*>(&gf[ks]);
        const bf16x2* op = reinterpret_cast<const bf16x2*>(&no[ks]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          q2 = __builtin_amdgcn_fdot2_f32_bf16(qp[e], qp[e], q2, false);
          k2 = __builtin_amdgcn_fdot2_f32_bf16(kp[e], kp[e], k2, false);
          Di = __builtin_amdgcn_fdot2_f32_bf16(gp[e], op[e], Di, false);
        }
      }
      q2 += __shfl_xor(q2, 32);
      Di += __shfl_xor(Di, 32);
      k2 += __shfl_xor(k2, 32);
      rq = rcp(a.scale * sqrtf(q2));   // inf for a zero row: the product below is clamped
      if (kt == 0) {   // the two waves of a query tile hold the same q / dO: one of them publishes
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int d = 16 * ks + 8 * lh + e;
            sGT[d * VTS + iq] = gf[ks][e];
            sQT[d * VTS + iq] = qf[ks][e];
          }
      }
      if (lh == 0) sCorrK[qt][jk] = 0.f;   // this wave's own region; filled by its slow path only
      if (qt == 0) {
        if (lh == 0) {
          sRk[jk] = rcp(sqrtf(k2));
          sCnt[jk] = tkk.cnt;
        }
        float rkm = k2 > 0.f ? rcp(sqrtf(k2)) : 0.f;   // zero keys (padding, dead rows) cannot contribute: u = 0
#pragma unroll
        for (int m = 1; m < 32; m <<= 1) rkm = fmaxf(rkm, __shfl_xor(rkm, m));
        if (lane == 0) sRkMax[kt] = rkm;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int e = 0; e < 8; ++e) sKT[(16 * ks + 8 * lh + e) * VTS + jk] = kf[ks][e];
      }
    }
    lds_barrier();

    // ---------------- element pass: column = query iq, rows = keys of tile kt
    f32x16 dq, dk, dv;
    float corrQ_keep = 0.f;
    {
      f32x16 ut, dt;
#pragma unroll
      for (int r = 0; r < 16; ++r) ut[r] = dt[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        ut = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], ut, 0, 0, 0);
        dt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[ks], gf[ks], dt, 0, 0, 0);
      }
      if (win + (int)gridDim.x < nWin) fetch(win + gridDim.x);
      float w1[16];
      // could any pair of this wave fall under the clamp?  1 / (|scale q_i| |k_j|) > 1e6 for the smallest non-zero |k| of the tile
      float corrQ = 0.f;
      const bool slow = __ballot(rq < INFINITY && rq * sRkMax[kt] > 1e6f) != 0;
      bf16_t* pcol = sP + (32 * kt + 4 * lh) * VTS + iq;
      bf16_t* wcol = sW + (32 * kt + 4 * lh) * VTS + iq;
      // per key group of four: 1 / |k|, the two table entries (and, under the shifted-window mask, the region ids) in one
      // 16-byte read each -- own data of the lane, so the 16 elements are independent chains; the arithmetic runs on float
      // pairs (v_pk_mul / v_pk_fma / v_pk_add_f32: two elements per instruction)
      const f32x2 sc2 = {a.scale, a.scale}, rq2 = {rq, rq}, nlse2 = {-lse, -lse}, nDi2 = {-Di, -Di};
      const f32x2 l2e2 = {1.44269504088896341f, 1.44269504088896341f};
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float4 v = *reinterpret_cast<const float4*>(&sRk[32 * kt + 8 * g4 + 4 * lh]);
        const float4 t = sTabL[0][g4][tid], b = sTabL[1][g4][tid];
        float4 pn = make_float4(0.f, 0.f, 0.f, 0.f);
        if (masked) {
          const int4 c4 = *reinterpret_cast<const int4*>(&sCnt[32 * kt + 8 * g4 + 4 * lh]);
          pn.x = c4.x != tq.cnt ? -100.f : 0.f;
          pn.y = c4.y != tq.cnt ? -100.f : 0.f;
          pn.z = c4.z != tq.cnt ? -100.f : 0.f;
          pn.w = c4.w != tq.cnt ? -100.f : 0.f;
        }
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
          const int r = 4 * g4 + 2 * hp, jr = 2 * hp + 8 * g4;   // jr: key row inside the tile, less 4 * lh
          const f32x2 rk2 = hp ? (f32x2){v.z, v.w} : (f32x2){v.x, v.y};
          const f32x2 ti2 = hp ? (f32x2){fabsf(t.z), fabsf(t.w)} : (f32x2){fabsf(t.x), fabsf(t.y)};
          const f32x2 bi2 = hp ? (f32x2){b.z, b.w} : (f32x2){b.x, b.y};
          const f32x2 pn2 = hp ? (f32x2){pn.z, pn.w} : (f32x2){pn.x, pn.y};
          const f32x2 ut2 = {ut[r], ut[r + 1]}, dt2 = {dt[r], dt[r + 1]};
          f32x2 rden = rq2 * rk2;                                  // 1 / max(|scale q||k|, 1e-6)
          const bool cx = slow && rden.x > 1e6f, cy = slow && rden.y > 1e6f;
          rden.x = fminf(rden.x, 1e6f);
          rden.y = fminf(rden.y, 1e6f);
          const f32x2 c = (ut2 * sc2) * rden;
          f32x2 sv = __builtin_elementwise_fma(c, ti2, bi2);
          if (masked) sv += pn2;
          const f32x2 ea = (sv + nlse2) * l2e2;
          const f32x2 pp = {__builtin_amdgcn_exp2f(ea.x), __builtin_amdgcn_exp2f(ea.y)};
          const f32x2 ds = pp * (dt2 + nDi2);
          accb2[r >> 1] += ds;
          acct2[r >> 1] = __builtin_elementwise_fma(ds, c, acct2[r >> 1]);
          const f32x2 ww = (ds * ti2) * rden;
          if (slow) {   // wave-uniform; W_ij u_ij of the clamped pairs: per query here, per key across the 32 query lanes
            float ex = cx ? ww.x * ut2.x : 0.f, ey = cy ? ww.y * ut2.y : 0.f;
            corrQ += ex + ey;
#pragma unroll
            for (int m = 1; m < 32; m <<= 1) {
              ex += __shfl_xor(ex, m);
              ey += __shfl_xor(ey, m);
            }
            if (l31 == 0) {
              sCorrK[qt][32 * kt + 4 * lh + jr] = ex;
              sCorrK[qt][32 * kt + 4 * lh + jr + 1] = ey;
            }
          }
          w1[r] = ww.x;
          w1[r + 1] = ww.y;
          pcol[jr * VTS] = (bf16_t)pp.x;
          pcol[(jr + 1) * VTS] = (bf16_t)pp.y;
          wcol[jr * VTS] = (bf16_t)ww.x;
          wcol[(jr + 1) * VTS] = (bf16_t)ww.y;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[r] = dk[r] = dv[r] = 0.f;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
        dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag(sKT, l31, lh, kt, s2), pack8(w1 + 8 * s2), dq, 0, 0, 0);
      // dv, dk partials of (key tile kt) over (query tile qt): the B operand is this wave's own P / W block
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag(sGT, l31, lh, qt, s2), tfrag(sP + 32 * kt * VTS, l31, lh, qt, s2), dv, 0, 0, 0);
        dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag(sQT, l31, lh, qt, s2), tfrag(sW + 32 * kt * VTS, l31, lh, qt, s2), dk, 0, 0, 0);
      }
      if (kt == 1) {   // hand the dq partial to the kt = 0 wave of this query tile
        float* red = &sRedQ[qt][lane * 17];
#pragma unroll
        for (int r = 0; r < 16; ++r) red[r] = dq[r];
        red[16] = corrQ;
      }
      corrQ_keep = corrQ;
    }
    lds_barrier();   // dq partials visible; every wave is done with P / W
    if (kt == 0 && iq < N) {
      // dq_i = scale * (A_i - (q_i . A_i) / |q_i|^2 q_i), A_i = sum_j W_ij k_j
      const float* r1 = &sRedQ[qt][lane * 17];
      bf16_t* drow = dqkv + (size_t)tq.tok * a.lddq + h * AD + 4 * lh;
      float qv[16], dot = 0.f, q2 = 0.f;
      // q_i from the transposed LDS tile, not from global memory again: a load here waits behind the next window's prefetch
      // and in front of this window's stores (vmcnt is in order), a full memory round trip per window and epilogue
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * q4 + e;
          qv[r] = (float)sQT[(4 * lh + 8 * q4 + e) * VTS + iq];
          dq[r] += r1[r];
          dot = fmaf(qv[r], dq[r], dot);
          q2 = fmaf(qv[r], qv[r], q2);
        }
      }
      dot += __shfl_xor(dot, 32);
      q2 += __shfl_xor(q2, 32);
      float cq = corrQ_keep + r1[16];          // the clamped pairs of both key tiles ...
      cq += __shfl_xor(cq, 32);                // ... and both halves of the rows: they have no projection term
      const float pr = (dot - cq) * rcp(fmaxf(q2, 1e-30f));
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        bf16x4 o4;
#pragma unroll
        for (int e = 0; e < 4; ++e) o4[e] = (bf16_t)(a.scale * (dq[4 * q4 + e] - pr * qv[4 * q4 + e]));
        *reinterpret_cast<bf16x4*>(drow + 8 * q4) = o4;
      }
    }
    if (qt == 1) {     // hand the dk / dv partials to the qt = 0 wave of this key tile
      float* red = sRedK + (kt * 2 + 0) * 64 * 17 + lane * 17;
      float* red2 = sRedK + (kt * 2 + 1) * 64 * 17 + lane * 17;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        red[r] = dk[r];
        red2[r] = dv[r];
      }
    }
    lds_barrier();
    if (qt == 0 && jk < N) {
      // dk_j = B_j - (k_j . B_j) / |k_j|^2 k_j, B_j = scale * sum_i W_ij q_i
      const float* k1 = sRedK + (kt * 2 + 0) * 64 * 17 + lane * 17;
      const float* v1 = sRedK + (kt * 2 + 1) * 64 * 17 + lane * 17;
      bf16_t* drow = dqkv + (size_t)tkk.tok * a.lddq + a.C + h * AD + 4 * lh;
      float kv[16], dot = 0.f, k2 = 0.f;
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * q4 + e;
          kv[r] = (float)sKT[(4 * lh + 8 * q4 + e) * VTS + jk];
          dk[r] = a.scale * (dk[r] + k1[r]);
          dot = fmaf(kv[r], dk[r], dot);
          k2 = fmaf(kv[r], kv[r], k2);
        }
      }
      dot += __shfl_xor(dot, 32);
      k2 += __shfl_xor(k2, 32);
      dot -= a.scale * (sCorrK[0][jk] + sCorrK[1][jk]);   // clamped pairs (both query tiles): no projection term
      const float pr = dot * rcp(fmaxf(k2, 1e-30f));
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        bf16x4 o4, o5;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o4[e] = (bf16_t)(dk[4 * q4 + e] - pr * kv[4 * q4 + e]);
          o5[e] = (bf16_t)(dv[4 * q4 + e] + v1[4 * q4 + e]);
        }
        *reinterpret_cast<bf16x4*>(drow + 8 * q4) = o4;
        *reinterpret_cast<bf16x4*>(drow + a.C + 8 * q4) = o5;
      }
    }
  }
  // The sums leave through the staging tile as whole rows (a lane's own 16 elements are 4-byte pieces of 32 different rows:
  // written directly they were 2048 partial-sector writes per wave and table)
  float* part = a.partial + ((size_t)blockIdx.x * 2 * a.heads + h) * N * N;   // [row][2][heads][N][N]
  const size_t tau_off = (size_t)a.heads * N * N;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    __syncthreads();   // the last window's readers of sPW (t = 0) / the row stores of t = 0 are done
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float tis = reinterpret_cast<const float*>(&sTabL[0][r >> 2][tid])[r & 3];
      // d/dtau of c / clip(tau, 0.01): -sum(dS c) / tau^2 where the clip is not active, else 0
      stage[iq * ANS + j] = t == 0 ? accb2[r >> 1][r & 1] : (tis > 0.f ? -acct2[r >> 1][r & 1] * tis * tis : 0.f);
    }
    __syncthreads();
    for (int r = w; r < N; r += 4)
      if (lane < N && !(UZ_KFLAGS(a) & 0x20000)) part[(t ? tau_off : 0) + r * N + lane] = stage[r * ANS + lane];
  }
}

// bf16 forward, block-per-wave form (the decomposition of winattn_bwd_mfma_kernel): one workgroup per
// (window, head), wave w owns the 32 x 32 score block (query tile qt = w >> 1, key tile kt = w & 1) with
// column = query.  The two waves of a query tile exchange their row maxima, then their row sums and partial
// P V products, through LDS.  ~100 VGPRs and 48 KB of LDS: three workgroups per CU where the one-wave-per-unit
// kernel above (325 VGPRs) fits one.
__global__ __launch_bounds__(256) void winattn_fwd_mfma2_kernel(const AttnArgs a) {
  __shared__ __attribute__((aligned(16))) bf16_t sVT[AD * VTS];
  __shared__ float sRedO[2][64 * 17];   // P V hand-over of the kt = 1 waves, [lane][16] (+1 pad)
  __shared__ float sM[2][AN], sL[2][AN];
  __shared__ __attribute__((aligned(16))) float sRk[AN];
  __shared__ __attribute__((aligned(16))) int sCnt[AN];
  // 1 / clip(tau) and bias of this head (padding: bias = -1e30) in lane order, as in winattn_bwd_mfma_kernel: [table][key group
  // of four][thread] -> the lane's own four values in one conflict-free 16-byte read
  __shared__ float4 sTabL[2][4][256];
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, l31 = lane & 31, lh = lane >> 5, h = blockIdx.y;
  const int qt = w >> 1, kt = w & 1;
  const int N = a.ws * a.ws;
  const int nWin = a.B * (a.H / a.ws) * (a.W / a.ws);
  const bool masked = a.shift > 0;
  const bf16_t* __restrict__ qkv = static_cast<const bf16_t*>(a.qkv);
  bf16_t* __restrict__ out = static_cast<bf16_t*>(a.out);
  const int iq = 32 * qt + l31, jk = 32 * kt + l31;   // this lane's query (column role) / key (column role)
  {
    // coalesced rows from global memory (all 32 loads of a thread in flight at once) into a [query][key] staging tile -- the
    // P V hand-over area, not yet in use --, then each lane gathers its own 16 entries
    float* stage = &sRedO[0][0];   // 32 query rows at a time
    static_assert(32 * ANS <= 2 * 64 * 17, "half the staging tile must fit in sRedO");
    float tv[16], bv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int e = tid + 256 * i, r = e >> 6, c = e & 63;
      const bool in = r < N && c < N;
      // unconditional loads (entry (0, 0) for a lane outside the window, then a select): as `in ? table[...] : pad` every
      // load was waited for before the next was issued
      tv[i] = a.tau[in ? ((size_t)h * a.Nt + r) * a.Nt + c : 0];
      bv[i] = a.bias[in ? ((size_t)h * N + r) * N + c : 0];
    }
    // (the loaded values are made opaque before the selects: a value that is only used when `in` holds is otherwise turned back
    // into a load under a branch)
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(tv[i]), "+v"(bv[i]));
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int e = tid + 256 * i, r = e >> 6, c = e & 63;
      const bool in = r < N && c < N;
      tv[i] = in ? tv[i] : 1.f;
      bv[i] = in ? bv[i] : -1e30f;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int i = 8 * half; i < 8 * half + 8; ++i) {   // rows 32 half .. 32 half + 31
          const int e = tid + 256 * i, r = (e >> 6) - 32 * half, c = e & 63;
          stage[r * ANS + c] = t == 0 ? 1.f / fmaxf(tv[i], 0.01f) : bv[i];
        }
        __syncthreads();
        if (qt == half) {
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const float* sp = stage + l31 * ANS + 32 * kt + 8 * g4 + 4 * lh;
            sTabL[t][g4][tid] = make_float4(sp[0], sp[1], sp[2], sp[3]);
          }
        }
        __syncthreads();
      }
    }
  }

  WinTok ntq = {0, -1}, ntk = {0, -1};
  bf16x8 nq[2], nk[2], nv[2];
  // The next window's operands, requested while this one is computed.  UNCONDITIONAL loads: a lane beyond the window (7 x 7
  // windows) reads token 0 of its window and is zeroed by a select -- as loads under `if (iq < N)` each group was closed by
  // s_waitcnt vmcnt(0) (hipcc 7.2) and the prefetch waited for its own data in the middle of the current window.
  auto fetch = [&](int win) {
    const bool qin = iq < N, kin = jk < N;
    const WinTok tq0 = win_token(a, win, qin ? iq : 0), tk0 = win_token(a, win, kin ? jk : 0);
    const bf16_t* rowq = qkv + (size_t)tq0.tok * a.ldq + h * AD + 8 * lh;
    const bf16_t* rowk = qkv + (size_t)tk0.tok * a.ldq + a.C + h * AD + 8 * lh;
    // the RAW loaded registers are kept; lanes beyond the window are zeroed where the registers are consumed, one window
    // later (a select here would wait for the data at once)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      nq[ks] = *reinterpret_cast<const bf16x8*>(rowq + 16 * ks);
      nk[ks] = *reinterpret_cast<const bf16x8*>(rowk + 16 * ks);
      nv[ks] = *reinterpret_cast<const bf16x8*>(rowk + a.C + 16 * ks);
    }
    ntq = qin ? tq0 : WinTok{0, -1};
    ntk = kin ? tk0 : WinTok{0, -1};
  };
  const bool qin = iq < N, kin = jk < N;
  bf16x8 zero8;
#pragma unroll
  for (int e = 0; e < 8; ++e) zero8[e] = (bf16_t)0.f;
  if ((int)blockIdx.x < nWin) fetch(blockIdx.x);

  for (int win = blockIdx.x; win < nWin; win += gridDim.x) {
    lds_barrier();   // previous window's readers are done (and sTab has landed)
    const WinTok tq = ntq, tkk = ntk;
    bf16x8 qf[2], kf[2];
    float rq;
    {
      float q2 = 0.f, k2 = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        qf[ks] = qin ? nq[ks] : zero8;
        kf[ks] = kin ? nk[ks] : zero8;
        const bf16x2* qp = reinterpret_cast<const bf16x2*>(&qf[ks]);
        const bf16x2* kp = reinterpret_cast<const bf16x2*>(&kf[ks]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {   // v_dot2c_f32_bf16
          q2 = __builtin_amdgcn_fdot2_f32_bf16(qp[e], qp[e], q2, false);
          k2 = __builtin_amdgcn_fdot2_f32_bf16(kp[e], kp[e], k2, false);
        }
      }
      q2 += __shfl_xor(q2, 32);
      k2 += __shfl_xor(k2, 32);
      rq = rcp(a.scale * sqrtf(q2));   // inf for a zero row: the product below is clamped
      if (qt == 0) {
        if (lh == 0) {
          sRk[jk] = rcp(sqrtf(k2));
          sCnt[jk] = tkk.cnt;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int e = 0; e < 8; ++e) sVT[(16 * ks + 8 * lh + e) * VTS + jk] = kin ? nv[ks][e] : (bf16_t)0.f;
      }
    }
    lds_barrier();
    f32x16 ut;
#pragma unroll
    for (int r = 0; r < 16; ++r) ut[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) ut = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], ut, 0, 0, 0);
    {   // (unconditionally: the last window of a workgroup fetches itself again rather than putting the loads under a branch)
      const int nxt = win + (int)gridDim.x;
      fetch(nxt < nWin ? nxt : win);
    }
    float sv[16], mx = -3.0e38f;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {   // key group of four: one 16-byte read per operand, 16 independent element chains
      const float4 kq = *reinterpret_cast<const float4*>(&sRk[32 * kt + 8 * g4 + 4 * lh]);
      const float4 t = sTabL[0][g4][tid], b = sTabL[1][g4][tid];
      const float rk4[4] = {kq.x, kq.y, kq.z, kq.w}, ti4[4] = {t.x, t.y, t.z, t.w}, bi4[4] = {b.x, b.y, b.z, b.w};
      float pen4[4] = {0.f, 0.f, 0.f, 0.f};
      if (masked) {
        const int4 c4 = *reinterpret_cast<const int4*>(&sCnt[32 * kt + 8 * g4 + 4 * lh]);
        pen4[0] = c4.x != tq.cnt ? -100.f : 0.f;
        pen4[1] = c4.y != tq.cnt ? -100.f : 0.f;
        pen4[2] = c4.z != tq.cnt ? -100.f : 0.f;
        pen4[3] = c4.w != tq.cnt ? -100.f : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g4 + e;
        const float rden = fminf(rq * rk4[e], 1e6f);      // 1 / max(|scale q||k|, 1e-6)
        const float v = fmaf(ut[r] * a.scale * rden, ti4[e], bi4[e]) + pen4[e];
        sv[r] = v;
        mx = fmaxf(mx, v);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (lh == 0) sM[kt][iq] = mx;
    lds_barrier();
    const float m = fmaxf(sM[0][iq], sM[1][iq]);
    float ls = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sv[r] = __expf(sv[r] - m);
      ls += sv[r];
    }
    ls += __shfl_xor(ls, 32);
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
      o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag(sVT, l31, lh, kt, s2), pack8(sv + 8 * s2), o, 0, 0, 0);
    if (kt == 1) {
      float* red = &sRedO[qt][lane * 17];
#pragma unroll
      for (int r = 0; r < 16; ++r) red[r] = o[r];
      if (lh == 0) sL[qt][l31] = ls;
    }
    lds_barrier();
    if (kt == 0 && iq < N) {
      const float* r1 = &sRedO[qt][lane * 17];
      const float l = ls + sL[qt][l31];
      const float inv = rcp(l);
      bf16_t* orow = out + (size_t)tq.tok * a.ldo + h * AD + 4 * lh;
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        bf16x4 o4;
#pragma unroll
        for (int e = 0; e < 4; ++e) o4[e] = (bf16_t)((o[4 * q4 + e] + r1[4 * q4 + e]) * inv);
        *reinterpret_cast<bf16x4*>(orow + 8 * q4) = o4;
      }
      if (lh == 0) a.lse[((size_t)win * a.heads + h) * N + iq] = m + __logf(l);
    }
  }
}

// Backward: one 256-thread workgroup (one wave per SIMD) per (window, head).  Phase A: lane = query i, the
// four waves split the key range; phase B: lane = key j, the waves split the query range; per-wave
// partial sums of dq / dk / dv meet in LDS and are added in a fixed order.  dS-derived sums for d(bias)
// and d(tau) accumulate over the workgroup's windows in LDS (wave w owns its key columns).
template <typename T>
__global__ __launch_bounds__(256) void winattn_bwd_kernel(const AttnArgs a) {
  constexpr int TILE = AN * ARS, MAT = AN * ANS;
  __shared__ float smem[2 * TILE + 2 * MAT + 4 * TILE + 2 * MAT + 3 * AN];
  __shared__ int sCnt[AN];
  float* const sK = smem;                 // phase A: K rows; phase B: Q rows
  float* const sV = smem + TILE;          // phase A: V rows; phase B: dO rows
  float* const sP = smem + 2 * TILE;
  float* const sDC = sP + MAT;
  float* const sRA = sDC + MAT;           // [4][AN][ARS]: per-wave partial (dq vector part, scalar part)
  float* const sDB = sRA + 4 * TILE;
  float* const sDT = sDB + MAT;
  float* const sKn = sDT + MAT;
  float* const sQn = sKn + AN;
  float* const sPB = smem;                // [4][AN][ANS] partial (dk, dv, scalar), aliases sK .. sRA after phase B
  static_assert(4 * MAT <= 2 * TILE + 2 * MAT + 4 * TILE, "phase B partials must fit the aliased region");
  const int tid = threadIdx.x, w = tid >> 6, i = tid & 63, h = blockIdx.y;
  const int N = a.ws * a.ws;
  const int jc = (N + 3) >> 2, lo = w * jc, hi = min(N, lo + jc);
  const int nWin = a.B * (a.H / a.ws) * (a.W / a.ws);
  const T* __restrict__ qkv = static_cast<const T*>(a.qkv);
  const T* __restrict__ out = static_cast<const T*>(a.out);
  const T* __restrict__ dout = static_cast<const T*>(a.dout);
  T* __restrict__ dqkv = static_cast<T*>(a.dqkv);
  for (int e = tid; e < MAT; e += 256) sDB[e] = sDT[e] = 0.f;
  for (int win = blockIdx.x; win < nWin; win += gridDim.x) {
    __syncthreads();
    float q[AD], kk[AD], go[AD];
    float qn = 0.f, kn = 0.f, Di = 0.f, rqn = 0.f, rkn = 0.f;
    WinTok me = {0, 0};
    if (i < N) {
      me = win_token(a, win, i);
      const T* row = qkv + (size_t)me.tok * a.ldq + h * AD;
      float t[AD];
      load_head(row, q);
      load_head(row + a.C, kk);
      load_head(dout + (size_t)me.tok * a.lddo + h * AD, go);
      load_head(out + (size_t)me.tok * a.ldo + h * AD, t);
#pragma unroll
      for (int e = 0; e < AD; ++e) {
        q[e] *= a.scale;
        qn += q[e] * q[e];
        kn += kk[e] * kk[e];
        Di = fmaf(go[e], t[e], Di);
      }
      qn = sqrtf(qn);
      kn = sqrtf(kn);
      rqn = rcp(qn);
      rkn = rcp(kn);
      if (w == 0) {
        load_head(row + 2 * a.C, t);
#pragma unroll
        for (int e = 0; e < AD; ++e) {
          sK[i * ARS + e] = kk[e];
          sV[i * ARS + e] = t[e];
        }
        sKn[i] = kn;
        sQn[i] = qn;
        sCnt[i] = me.cnt;
      }
    }
    __syncthreads();
    {  // ---- phase A: query i, keys [lo, hi)
      float av[AD], bs = 0.f;
#pragma unroll
      for (int e = 0; e < AD; ++e) av[e] = 0.f;
      if (i < N) {
        const float lse = a.lse[((size_t)win * a.heads + h) * N + i];
        for (int j = lo; j < hi; ++j) {
          float krow[AD], vrow[AD];
          lds_row(sK + j * ARS, krow);
          lds_row(sV + j * ARS, vrow);
          const float u = dot32(q, krow), dp = dot32(go, vrow);
          const float nn = qn * sKn[j];
          const bool clamped = nn <= 1e-6f;
          const float den = clamped ? 1e-6f : nn;
          const float tv = a.tau[((size_t)h * a.Nt + i) * a.Nt + j];
          const float ti = rcp(fmaxf(tv, 0.01f));
          const float rden = rcp(den);
          const float c = u * rden;
          float s = c * ti + a.bias[((size_t)h * N + i) * N + j];
          if (sCnt[j] != me.cnt) s -= 100.f;
          const float p = __expf(s - lse);
          const float ds = p * (dp - Di);
          sP[i * ANS + j] = p;
          sDB[i * ANS + j] += ds;
          if (tv >= 0.01f) sDT[i * ANS + j] -= ds * c * ti * ti;
          const float dc = ds * ti;
          sDC[i * ANS + j] = dc;
          const float w1 = dc * rden;
          axpy32(w1, krow, av);
          if (!clamped) bs += dc * u * sKn[j] * rden * rden * rqn;  // d(den)/d(qs_i) = kn_j * qs_i / n_i
        }
      }
      float* ra = sRA + (w * AN + i) * ARS;
#pragma unroll
      for (int e = 0; e < AD; ++e) ra[e] = av[e];
      ra[AD] = bs;
    }
    __syncthreads();
    if (i < N) {  // dq: wave w finishes components [8w, 8w + 8) of query i; wave 0 re-stages Q and dO
      float tot[8], bt = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) tot[e] = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float* ra = sRA + (k * AN + i) * ARS;
        bt += ra[AD];
#pragma unroll
        for (int e = 0; e < 8; ++e) tot[e] += ra[8 * w + e];
      }
      float dq[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float qe = 0.f;  // q[8 w + e] without dynamic register indexing
#pragma unroll
        for (int k = 0; k < 4; ++k) qe = (w == k) ? q[8 * k + e] : qe;
        dq[e] = a.scale * (tot[e] - bt * qe);
      }
      store8(dqkv + (size_t)me.tok * a.lddq + h * AD + 8 * w, dq);
      if (w == 0) {
#pragma unroll
        for (int e = 0; e < AD; ++e) {
          sK[i * ARS + e] = q[e];
          sV[i * ARS + e] = go[e];
        }
      }
    }
    __syncthreads();
    float dk[AD], dv[AD], bsk = 0.f;
#pragma unroll
    for (int e = 0; e < AD; ++e) dk[e] = dv[e] = 0.f;
    if (i < N) {  // ---- phase B: key j = i, queries [lo, hi)
      const int j = i;
      for (int r = lo; r < hi; ++r) {
        const float p = sP[r * ANS + j], dc = sDC[r * ANS + j];
        float qrow[AD], grow[AD];
        lds_row(sK + r * ARS, qrow);
        lds_row(sV + r * ARS, grow);
        const float u = dot32(qrow, kk);
        axpy32(p, grow, dv);
        const float nn = sQn[r] * kn;
        const bool clamped = nn <= 1e-6f;
        const float den = clamped ? 1e-6f : nn;
        const float rden = rcp(den);
        const float w1 = dc * rden;
        axpy32(w1, qrow, dk);
        if (!clamped) bsk += dc * u * sQn[r] * rden * rden * rkn;
      }
    }
    __syncthreads();  // everyone is done with sP / sDC / the Q, dO tiles: the partials may overwrite them
    {
      float* pb = sPB + (w * AN + i) * ANS;
#pragma unroll
      for (int e = 0; e < AD; ++e) {
        pb[e] = dk[e];
        pb[AD + e] = dv[e];
      }
      pb[2 * AD] = bsk;
    }
    __syncthreads();
    if (i < N) {  // waves 0, 1: halves of dk; waves 2, 3: halves of dv
      float tot[16], bt = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) tot[e] = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float* pb = sPB + (k * AN + i) * ANS;
        bt += pb[2 * AD];
#pragma unroll
        for (int e = 0; e < 16; ++e) tot[e] += pb[16 * w + e];
      }
      if (w < 2) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float ke = (w == 0) ? kk[e] : kk[16 + e];
          tot[e] -= bt * ke;
        }
      }
      T* row = dqkv + (size_t)me.tok * a.lddq + h * AD + (w < 2 ? a.C + 16 * w : 2 * a.C + 16 * (w - 2));
      store8(row, tot);
      store8(row + 8, tot + 8);
    }
  }
  __syncthreads();
  float* part = a.partial + ((size_t)blockIdx.x * 2 * a.heads + h) * N * N;   // [row][2][heads][N][N]
  const size_t tau_off = (size_t)a.heads * N * N;
  for (int e = tid; e < N * N; e += 256) {
    const int r = e / N, c = e - r * N;
    part[e] = sDB[r * ANS + c];
    part[tau_off + e] = sDT[r * ANS + c];
  }
}
// (Measured and rejected: keeping the tau / bias values and the d(bias) / d(tau) sums of a lane's 16 keys in
// registers with the key loop fully unrolled — 15 % slower, the unrolled body no longer fits the
// instruction cache; recomputing P in phase B to halve LDS and double the occupancy — 26 % slower.)

// ---------------------------------------------------------------------------------------------
// Continuous position bias: bias[h][r] = b2[h] + sum_k w2[h][k] relu(w1[k][0] x0(r) + w1[k][1] x1(r) + b1[k])
// over the R = N*N log-spaced offsets (get_continuous_relative_position_bias, swin_unet_v2.py:121-125 with
// Mlp_Relu :58-72).  A function of parameters only; R <= 4096, hidden = 256, heads <= 32.
// ---------------------------------------------------------------------------------------------
constexpr int CPB_MAXH = 32;
constexpr int CPB_MAXHID = 512;

// grid (R / 256, heads): one thread per (offset r, head h); fc1 and this head's fc2 row sit in LDS
__global__ __launch_bounds__(256) void cpb_fwd_kernel(const float* __restrict__ idx, const float* __restrict__ w1,
                                                      const float* __restrict__ b1, const float* __restrict__ w2,
                                                      const float* __restrict__ b2, int R, int hidden, int heads,
                                                      float* __restrict__ bias) {
  __shared__ float4 sW[CPB_MAXHID];  // (w1[k][0], w1[k][1], b1[k], w2[h][k])
  const int h = blockIdx.y;
  for (int k = threadIdx.x; k < hidden; k += 256)
    sW[k] = make_float4(w1[2 * k], w1[2 * k + 1], b1[k], w2[h * hidden + k]);
  __syncthreads();
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const float x0 = idx[2 * r], x1 = idx[2 * r + 1];
  float acc = b2[h];
#pragma unroll 8
  for (int k = 0; k < hidden; ++k) {
    const float4 wv = sW[k];
    acc = fmaf(wv.w, fmaxf(fmaf(wv.x, x0, fmaf(wv.y, x1, wv.z)), 0.f), acc);
  }
  bias[(size_t)h * R + r] = acc;
}

// one 1024-thread workgroup per hidden unit (the kernel is bound by the latency of the few G / idx loads each
// thread issues, so the rows are spread over 16 waves).  Sums over the R rows in a fixed order: per-thread
// strided sums, xor-shuffle within a wave, then the sixteen waves in order.
constexpr int CPB_KB = 1;
constexpr int CPB_NW = 16;

__global__ __launch_bounds__(1024) void cpb_bwd_kernel(const float* __restrict__ idx, const float* __restrict__ w1,
                                                      const float* __restrict__ b1, const float* __restrict__ w2,
                                                      const float* __restrict__ G, int R, int hidden, int heads,
                                                      float* __restrict__ dw1, float* __restrict__ db1,
                                                      float* __restrict__ dw2, float* __restrict__ db2) {
  __shared__ float red[CPB_NW][CPB_KB * (CPB_MAXH + 3) + CPB_MAXH];
  const int k0 = blockIdx.x * CPB_KB, t = threadIdx.x, lane = t & 63, wv = t >> 6;
  float wa[CPB_KB], wb[CPB_KB], bk[CPB_KB];
#pragma unroll
  for (int q = 0; q < CPB_KB; ++q) {
    const int k = min(k0 + q, hidden - 1);
    wa[q] = w1[2 * k];
    wb[q] = w1[2 * k + 1];
    bk[q] = b1[k];
  }
  float a2[CPB_KB][CPB_MAXH], g2[CPB_MAXH], a10[CPB_KB], a11[CPB_KB], ab[CPB_KB];
#pragma unroll
  for (int q = 0; q < CPB_KB; ++q) {
    a10[q] = a11[q] = ab[q] = 0.f;
#pragma unroll
    for (int h = 0; h < CPB_MAXH; ++h) a2[q][h] = 0.f;
  }
#pragma unroll
  for (int h = 0; h < CPB_MAXH; ++h) g2[h] = 0.f;
  for (int r = t; r < R; r += 64 * CPB_NW) {
    const float x0 = idx[2 * r], x1 = idx[2 * r + 1];
    float pre[CPB_KB], hv[CPB_KB], gs[CPB_KB];
#pragma unroll
    for (int q = 0; q < CPB_KB; ++q) {
      pre[q] = fmaf(wa[q], x0, fmaf(wb[q], x1, bk[q]));
      hv[q] = fmaxf(pre[q], 0.f);
      gs[q] = 0.f;
    }
#pragma unroll
    for (int h = 0; h < CPB_MAXH; ++h)
      if (h < heads) {
        const float g = G[(size_t)h * R + r];
        g2[h] += g;
#pragma unroll
        for (int q = 0; q < CPB_KB; ++q) {
          gs[q] = fmaf(g, w2[h * hidden + min(k0 + q, hidden - 1)], gs[q]);
          a2[q][h] = fmaf(g, hv[q], a2[q][h]);
        }
      }
#pragma unroll
    for (int q = 0; q < CPB_KB; ++q) {
      const float dl = pre[q] > 0.f ? gs[q] : 0.f;
      a10[q] = fmaf(dl, x0, a10[q]);
      a11[q] = fmaf(dl, x1, a11[q]);
      ab[q] += dl;
    }
  }
  auto wsum = [](float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
  };
  constexpr int PERK = CPB_MAXH + 3;
#pragma unroll
  for (int q = 0; q < CPB_KB; ++q) {
    const float s0 = wsum(a10[q]), s1 = wsum(a11[q]), s2 = wsum(ab[q]);
    if (lane == 0) {
      red[wv][q * PERK] = s0;
      red[wv][q * PERK + 1] = s1;
      red[wv][q * PERK + 2] = s2;
    }
#pragma unroll
    for (int h = 0; h < CPB_MAXH; ++h)
      if (h < heads) {
        const float v = wsum(a2[q][h]);
        if (lane == 0) red[wv][q * PERK + 3 + h] = v;
      }
  }
  if (blockIdx.x == 0) {
#pragma unroll
    for (int h = 0; h < CPB_MAXH; ++h)
      if (h < heads) {
        const float v = wsum(g2[h]);
        if (lane == 0) red[wv][CPB_KB * PERK + h] = v;
      }
  }
  __syncthreads();
  if (t < CPB_KB * PERK + CPB_MAXH) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < CPB_NW; ++k) v += red[k][t];
    if (t < CPB_KB * PERK) {
      const int q = t / PERK, e = t - q * PERK, k = k0 + q;
      if (k < hidden) {
        if (e == 0) dw1[2 * k] = v;
        else if (e == 1) dw1[2 * k + 1] = v;
        else if (e == 2) db1[k] = v;
        else if (e - 3 < heads) dw2[(e - 3) * hidden + k] = v;
      }
    } else if (blockIdx.x == 0 && t - CPB_KB * PERK < heads) {
      db2[t - CPB_KB * PERK] = v;
    }
  }
}

// ---- all position-bias MLPs of a model in one launch ------------------------------------------------
// The MLPs are tiny (R <= 4096 offsets, 256 hidden units, <= 32 heads) and a function of parameters only, so
// a model's 14 of them are evaluated together at the start of the forward and differentiated together at the
// end of the backward: 2 + 1 launches instead of 28, and the backward reads each G once (cpb_bwd_kernel above
// re-reads it per hidden unit: 22 us per module, bound by L2).
constexpr int CPB_MAXB = 24;     // modules per launch (the descriptor array travels in the kernel arguments)
constexpr int CPB_ROWS = 128;    // offsets per workgroup of the batched backward
struct CpbBatch {
  uz_cpb_item it[CPB_MAXB];
  long long off[CPB_MAXB];       // float offset of the module's partial sums in the workspace
};

__global__ __launch_bounds__(256) void cpb_fwd_batched_kernel(const CpbBatch b) {
  __shared__ float4 sW[CPB_MAXHID];  // (w1[k][0], w1[k][1], b1[k], w2[h][k])
  const uz_cpb_item& m = b.it[blockIdx.z];
  const int h = blockIdx.y, R = m.R, hidden = m.hidden;
  if (h >= m.heads || (int)(blockIdx.x * 256) >= R) return;   // whole workgroups
  for (int k = threadIdx.x; k < hidden; k += 256)
    sW[k] = make_float4(m.w1[2 * k], m.w1[2 * k + 1], m.b1[k], m.w2[h * hidden + k]);
  __syncthreads();
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  const float x0 = m.idx[2 * r], x1 = m.idx[2 * r + 1];
  float acc = m.b2[h];
#pragma unroll 8
  for (int k = 0; k < hidden; ++k) {
    const float4 wv = sW[k];
    acc = fmaf(wv.w, fmaxf(fmaf(wv.x, x0, fmaf(wv.y, x1, wv.z)), 0.f), acc);
  }
  m.bias[(size_t)h * R + r] = acc;
}

// grid (row blocks, modules), 512 threads: thread = hidden unit k (512 / hidden groups split the block's rows).
// Partial sums per (row block, group): [(heads + 3)][hidden] = d w2[h][k] (heads), d w1[k][0], d w1[k][1], d b1[k];
// per row block: [heads] sums of G (d b2).  cpb_bwd_finalize_kernel adds them in a fixed order.
__global__ __launch_bounds__(512) void cpb_bwd_batched_kernel(const CpbBatch b, float* __restrict__ ws) {
  __shared__ __attribute__((aligned(16))) float sG[CPB_ROWS][CPB_MAXH];
  __shared__ float sX[CPB_ROWS][2];
  const uz_cpb_item& m = b.it[blockIdx.y];
  const int R = m.R, hidden = m.hidden, heads = m.heads, tid = threadIdx.x;
  const int r0 = blockIdx.x * CPB_ROWS;
  if (r0 >= R) return;
  const int hp = (heads + 3) & ~3;
  for (int e = tid; e < CPB_ROWS * hp; e += 512) {
    const int h = e / CPB_ROWS, r = e - h * CPB_ROWS;
    sG[r][h] = (h < heads && r0 + r < R) ? m.G[(size_t)h * R + r0 + r] : 0.f;
  }
  for (int e = tid; e < CPB_ROWS; e += 512) {
    const bool in = r0 + e < R;
    sX[e][0] = in ? m.idx[2 * (r0 + e)] : 0.f;
    sX[e][1] = in ? m.idx[2 * (r0 + e) + 1] : 0.f;
  }
  __syncthreads();
  const int nsub = 512 / hidden, RB = (R + CPB_ROWS - 1) / CPB_ROWS;
  const int nmain = (heads + 3) * hidden;
  float* main_ws = ws + b.off[blockIdx.y];
  if (tid < heads) {   // d b2 partial of this row block
    float t = 0.f;
    for (int r = 0; r < CPB_ROWS; ++r) t += sG[r][tid];
    main_ws[(size_t)RB * nsub * nmain + (size_t)blockIdx.x * heads + tid] = t;
  }
  const int sub = tid / hidden, k = tid - sub * hidden;
  if (sub >= nsub) return;
  const int rows_per = CPB_ROWS / nsub + (CPB_ROWS % nsub != 0);
  const int rb = sub * rows_per, re = min(rb + rows_per, CPB_ROWS);
  const float wa = m.w1[2 * k], wb = m.w1[2 * k + 1], bk = m.b1[k];
  float w2c[CPB_MAXH], a2[CPB_MAXH];
#pragma unroll
  for (int h = 0; h < CPB_MAXH; ++h) {
    w2c[h] = h < heads ? m.w2[h * hidden + k] : 0.f;
    a2[h] = 0.f;
  }
  float a10 = 0.f, a11 = 0.f, ab = 0.f;
  for (int r = rb; r < re; ++r) {
    const float x0 = sX[r][0], x1 = sX[r][1];
    const float pre = fmaf(wa, x0, fmaf(wb, x1, bk));
    const float hv = fmaxf(pre, 0.f);
    float gs = 0.f;
#pragma unroll
    for (int h4 = 0; h4 < CPB_MAXH / 4; ++h4)
      if (4 * h4 < hp) {
        const float4 g = *reinterpret_cast<const float4*>(&sG[r][4 * h4]);
        gs = fmaf(g.x, w2c[4 * h4], gs);
        gs = fmaf(g.y, w2c[4 * h4 + 1], gs);
        gs = fmaf(g.z, w2c[4 * h4 + 2], gs);
        gs = fmaf(g.w, w2c[4 * h4 + 3], gs);
        a2[4 * h4] = fmaf(g.x, hv, a2[4 * h4]);
        a2[4 * h4 + 1] = fmaf(g.y, hv, a2[4 * h4 + 1]);
        a2[4 * h4 + 2] = fmaf(g.z, hv, a2[4 * h4 + 2]);
        a2[4 * h4 + 3] = fmaf(g.w, hv, a2[4 * h4 + 3]);
      }
    const float dl = pre > 0.f ? gs : 0.f;
    a10 = fmaf(dl, x0, a10);
    a11 = fmaf(dl, x1, a11);
    ab += dl;
  }
  float* row = main_ws + ((size_t)blockIdx.x * nsub + sub) * nmain;
#pragma unroll
  for (int h = 0; h < CPB_MAXH; ++h)
    if (h < heads) row[h * hidden + k] = a2[h];
  row[heads * hidden + k] = a10;
  row[(heads + 1) * hidden + k] = a11;
  row[(heads + 2) * hidden + k] = ab;
}

__global__ __launch_bounds__(256) void cpb_bwd_finalize_kernel(const CpbBatch b, const float* __restrict__ ws) {
  const uz_cpb_item& m = b.it[blockIdx.y];
  const int hidden = m.hidden, heads = m.heads;
  const int nsub = 512 / hidden, RB = (m.R + CPB_ROWS - 1) / CPB_ROWS, rows = RB * nsub;
  const int nmain = (heads + 3) * hidden;
  const float* main_ws = ws + b.off[blockIdx.y];
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < nmain) {
    float t = 0.f;
    for (int r = 0; r < rows; r += 8) {   // eight rows per trip, unconditional loads, same order of additions
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = main_ws[(size_t)(r + u < rows ? r + u : 0) * nmain + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(v[u]));
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (r + u < rows) t += v[u];
    }
    const int j = e / hidden, k = e - j * hidden;
    if (j < heads) m.dw2[j * hidden + k] = t;
    else if (j == heads) m.dw1[2 * k] = t;
    else if (j == heads + 1) m.dw1[2 * k + 1] = t;
    else m.db1[k] = t;
  } else if (e - nmain < heads) {
    const float* tail = main_ws + (size_t)rows * nmain;
    float t = 0.f;
    for (int r = 0; r < RB; ++r) t += tail[(size_t)r * heads + (e - nmain)];
    m.db2[e - nmain] = t;
  }
}

inline int grid_cap(long long units, int per_block, int per_cu) {
  long long g = (units + per_block - 1) / per_block;
  const long long cap = (long long)UZ_NUM_CU * per_cu;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// see fdiv(): m = ceil(2^(32+s) / d), d = 1 -> m = 0
static FastDiv make_fastdiv(int d) {
  FastDiv f{0u, 0};
  if (d <= 1) return f;
  int S = 0;
  while ((1LL << S) < d) ++S;   // ceil(log2 d) >= 1
  f.s = S - 1;
  f.m = (unsigned)(((1ULL << (31 + S)) + (unsigned long long)d - 1) / (unsigned long long)d);
  return f;
}
static void ln_geometry(const uz_ln_desc* d, LnArgs* a) {
  a->fWo = make_fastdiv(d->Wo);
  a->fHo = make_fastdiv(d->Ho);
  a->fr = make_fastdiv(d->mode == 2 ? d->r : 1);
  a->Hin = d->mode == 1 ? 2 * d->Ho : d->mode == 2 ? d->Ho / d->r : d->Ho;
  a->Win = d->mode == 1 ? 2 * d->Wo : d->mode == 2 ? d->Wo / d->r : d->Wo;
}

int ln_lpt(const uz_ln_desc* d) {
  int cc = d->C / (d->dtype == UZ_BF16 ? 8 : 4);
  if (!(uz_tune_flags() & 0x100000)) cc = (cc + 2) / 3;   // three chunks per lane (see ln_unroll)
  int l = 1;
  while (l < cc && l < 64) l <<= 1;
  return l;
}

int ln_check(const char* fn, const uz_ln_desc* d) {
  UZ_REQUIRE(d != nullptr, "%s: null descriptor", fn);
  UZ_REQUIRE(d->dtype == UZ_F32 || d->dtype == UZ_BF16, "%s: bad dtype", fn);
  const int vec = d->dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(d->N > 0 && d->Ho > 0 && d->Wo > 0 && d->C > 0 && d->C % vec == 0, "%s: bad shape", fn);
  UZ_REQUIRE(d->C / vec <= 64 * LN_MAXIT && d->C <= LN_MAXC, "%s: C=%d too large (max %d)", fn, d->C, LN_MAXC);
  // dynamic LDS of the backward: 4 * (64 / lanes-per-token) groups x 2 x C floats
  UZ_REQUIRE((long long)4 * (64 / ln_lpt(d)) * 2 * d->C * 4 <= 64 * 1024, "%s: C=%d: partial-row staging exceeds 64 KiB", fn, d->C);
  UZ_REQUIRE(d->mode >= 0 && d->mode <= 2, "%s: bad mode %d", fn, d->mode);
  if (d->mode == 1) UZ_REQUIRE(d->C % (4 * vec) == 0 && d->ldx % vec == 0 && d->ldx >= d->C / 4, "%s: merge needs C %% %d == 0", fn, 4 * vec);
  if (d->mode == 2) UZ_REQUIRE(d->r >= 1 && d->Ho % d->r == 0 && d->Wo % d->r == 0 && d->ldx >= d->r * d->r * d->C, "%s: bad expand factor", fn);
  if (d->mode == 0) UZ_REQUIRE(d->ldx >= d->C, "%s: bad ldx", fn);
  UZ_REQUIRE(d->ldx % vec == 0, "%s: ldx must be a multiple of %d", fn, vec);
  UZ_REQUIRE((long long)d->N * d->Ho * d->Wo < (1LL << 31), "%s: too many tokens", fn);
  return UZ_OK;
}

// chunks per lane and tokens per lane group and pass of the instantiation that serves d
int ln_its(const uz_ln_desc* d) {
  const int vec = d->dtype == UZ_BF16 ? 8 : 4, lpt = ln_lpt(d);
  return (d->C / vec + lpt - 1) / lpt;
}
// tokens per lane group and pass.  Measured on the 1M-token expand LayerNorm of swin_unet_v2 (201 MB in, 201 MB
// out, three chunks per lane): forward U = 1 / 2 -> 101 / 119 us, backward 200 / 185 us; U = 4 spills.
int ln_unroll(const uz_ln_desc* d, bool bwd) {
  const int its = ln_its(d);
  const int u = (int)((uz_tune_flags() >> 16) & 15);
  if (u == 1 || u == 2 || u == 4 || (u == 8 && its == 1)) return its > 3 ? 1 : u;
  if (its > 3) return 1;
  if (its > 1) return bwd ? 2 : 1;
  return 4;
}

int ln_grid(const uz_ln_desc* d, bool bwd) {
  const int tpb = 4 * (64 / ln_lpt(d)) * ln_unroll(d, bwd);
  return grid_cap((long long)d->N * d->Ho * d->Wo, tpb, 8);
}

template <bool BWD>
void ln_launch(const uz_ln_desc* d, dim3 grid, dim3 block, size_t shm, hipStream_t st, const LnArgs& a, int lpt) {
  const int its = ln_its(d), u = ln_unroll(d, BWD);
#define UZ_LN(T, I, U)                                                                                   \
  do {                                                                                                   \
    if (a.act) hipLaunchKernelGGL((layernorm_kernel<T, BWD, I, U, true>), grid, block, shm, st, a, lpt); \
    else hipLaunchKernelGGL((layernorm_kernel<T, BWD, I, U, false>), grid, block, shm, st, a, lpt);      \
  } while (0)
#define UZ_LN_U(T, I) \
  do { if (u == 1) UZ_LN(T, I, 1); else if (u == 4) UZ_LN(T, I, 4); else UZ_LN(T, I, 2); } while (0)
  if (d->dtype == UZ_BF16) {
    if (its == 1) { if (u == 8) UZ_LN(bf16_t, 1, 8); else UZ_LN_U(bf16_t, 1); }
    else if (its == 2) UZ_LN_U(bf16_t, 2);
    else if (its == 3) UZ_LN_U(bf16_t, 3);
    else if (its <= 6) UZ_LN(bf16_t, 6, 1);
    else UZ_LN(bf16_t, 8, 1);
  } else {
    if (its == 1) { if (u == 8) UZ_LN(float, 1, 8); else UZ_LN_U(float, 1); }
    else if (its == 2) UZ_LN_U(float, 2);
    else if (its == 3) UZ_LN_U(float, 3);
    else if (its <= 6) UZ_LN(float, 6, 1);
    else UZ_LN(float, 8, 1);
  }
#undef UZ_LN_U
#undef UZ_LN
}

}  // namespace

extern "C" int uz_patchify(int dtype, const float* x_nchw, int N, int C, int H, int W, int patch, int Kpad,
                           void* out, void* stream) {
  UZ_REQUIRE(dtype == UZ_F32 || dtype == UZ_BF16, "uz_patchify: bad dtype");
  UZ_REQUIRE(x_nchw && out && N > 0 && C > 0 && patch > 0 && H % patch == 0 && W % patch == 0, "uz_patchify: bad shape");
  UZ_REQUIRE(Kpad >= patch * patch * C, "uz_patchify: Kpad too small");
  const long long total = (long long)N * (H / patch) * (W / patch) * Kpad;
  const dim3 grid(grid_cap(total, 256, 16)), block(256);
  if (dtype == UZ_BF16) hipLaunchKernelGGL((patchify_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, x_nchw, N, C, H, W, patch, Kpad, (bf16_t*)out);
  else hipLaunchKernelGGL((patchify_kernel<float>), grid, block, 0, (hipStream_t)stream, x_nchw, N, C, H, W, patch, Kpad, (float*)out);
  UZ_LAUNCH_CHECK("uz_patchify");
  return UZ_OK;
}

extern "C" int uz_layernorm_fwd(const uz_ln_desc* d, const void* x, const float* gamma, const float* beta,
                                const void* res, const float* image_scale, void* y, float* stats, void* stream) {
  const int rc = ln_check("uz_layernorm_fwd", d);
  if (rc != UZ_OK) return rc;
  const int vec = d->dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(x && gamma && beta && y && stats, "uz_layernorm_fwd: null pointer");
  UZ_REQUIRE((((uintptr_t)gamma | (uintptr_t)beta) & 15) == 0 && ((uintptr_t)stats & 7) == 0,
             "uz_layernorm_fwd: gamma / beta must be 16-byte aligned, stats 8-byte aligned");
  UZ_REQUIRE(d->ldy % vec == 0 && d->ldy >= d->C, "uz_layernorm_fwd: bad ldy");
  if (res) UZ_REQUIRE(d->ldr % vec == 0 && d->ldr >= d->C, "uz_layernorm_fwd: bad ldr");
  LnArgs a{};
  a.x = x; a.y = y; a.res = res; a.gamma = gamma; a.beta = beta; a.sb = image_scale; a.stats = stats;
  a.N = d->N; a.Ho = d->Ho; a.Wo = d->Wo; a.C = d->C; a.ldx = d->ldx; a.ldy = d->ldy; a.ldr = d->ldr;
  a.mode = d->mode; a.r = d->r; a.eps = d->eps; a.act = d->act;
  UZ_REQUIRE(d->act == 0 || (d->act == 1 && !res && !image_scale), "uz_layernorm_fwd: act = %d (GELU = 1 takes no residual / image scale)", d->act);
  ln_geometry(d, &a);
  const dim3 grid(ln_grid(d, false)), block(256);
  const int lpt = ln_lpt(d);
  ln_launch<false>(d, grid, block, 0, (hipStream_t)stream, a, lpt);
  UZ_LAUNCH_CHECK("uz_layernorm_fwd");
  return UZ_OK;
}

extern "C" int uz_layernorm_bwd_rows(const uz_ln_desc* d) {
  const int rc = ln_check("uz_layernorm_bwd_rows", d);
  if (rc != UZ_OK) return rc;
  return ln_grid(d, true);
}

static int ln_bwd_common(const char* fn, const uz_ln_desc* d, const void* x, const float* gamma, const float* beta,
                         const float* stats, const void* g, const float* image_scale, void* dx, float* partial,
                         void* stream) {
  const int rc = ln_check(fn, d);
  if (rc != UZ_OK) return rc;
  const int vec = d->dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(x && gamma && stats && g && dx && partial, "%s: null pointer", fn);
  UZ_REQUIRE((((uintptr_t)gamma | (uintptr_t)beta) & 15) == 0 && ((uintptr_t)stats & 7) == 0,
             "%s: gamma / beta must be 16-byte aligned, stats 8-byte aligned", fn);
  UZ_REQUIRE(d->ldg % vec == 0 && d->ldg >= d->C && d->lddx % vec == 0, "%s: bad ldg / lddx", fn);
  LnArgs a{};
  a.x = x; a.g = g; a.dx = dx; a.gamma = gamma; a.beta = beta; a.sb = image_scale; a.stats = const_cast<float*>(stats);
  a.partial = partial;
  a.N = d->N; a.Ho = d->Ho; a.Wo = d->Wo; a.C = d->C; a.ldx = d->ldx; a.ldg = d->ldg; a.lddx = d->lddx;
  a.mode = d->mode; a.r = d->r; a.eps = d->eps; a.act = d->act;
  ln_geometry(d, &a);
  const dim3 grid(ln_grid(d, true)), block(256);
  const int lpt = ln_lpt(d);
  const size_t shm = (size_t)4 * (64 / lpt) * 2 * d->C * sizeof(float);
  ln_launch<true>(d, grid, block, shm, (hipStream_t)stream, a, lpt);
  UZ_LAUNCH_CHECK(fn);
  return UZ_OK;
}

extern "C" int uz_layernorm_bwd(const uz_ln_desc* d, const void* x, const float* gamma, const float* stats,
                                const void* g, const float* image_scale, void* dx, float* partial, void* stream) {
  UZ_REQUIRE(d && d->act == 0, "uz_layernorm_bwd: an activation needs beta: call uz_layernorm_act_bwd");
  return ln_bwd_common("uz_layernorm_bwd", d, x, gamma, nullptr, stats, g, image_scale, dx, partial, stream);
}

extern "C" int uz_layernorm_act_bwd(const uz_ln_desc* d, const void* x, const float* gamma, const float* beta,
                                    const float* stats, const void* g, void* dx, float* partial, void* stream) {
  UZ_REQUIRE(d && d->act == 1 && beta, "uz_layernorm_act_bwd: act must be 1 (GELU) and beta given");
  return ln_bwd_common("uz_layernorm_act_bwd", d, x, gamma, beta, stats, g, nullptr, dx, partial, stream);
}

// ---- LayerNorm + 1x1 head --------------------------------------------------------------------------
constexpr int LNH_MAXK = 4;

// lanes per token: one class -> three chunks per lane (the layout the plain kernel measures fastest with);
// more classes -> one chunk per lane, so that wg and S of all classes stay in registers
static int ln_head_its(int K) { return K == 1 ? 3 : 1; }
static int ln_head_lpt(const uz_ln_desc* d, int K) {
  const int its = ln_head_its(K);
  const int cc = (d->C / (d->dtype == UZ_BF16 ? 8 : 4) + its - 1) / its;
  int l = 1;
  while (l < cc && l < 64) l <<= 1;
  return l;
}
static int ln_head_check(const char* fn, const uz_ln_desc* d, int K) {
  const int rc = ln_check(fn, d);
  if (rc != UZ_OK) return rc;
  UZ_REQUIRE(K >= 1 && K <= LNH_MAXK, "%s: K=%d classes (max %d)", fn, K, LNH_MAXK);
  const int vec = d->dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(d->C / vec <= 64 * ln_head_its(K), "%s: C=%d too wide for %d classes", fn, d->C, K);
  const long long shm = (long long)4 * (64 / ln_head_lpt(d, K)) * (K * d->C + K) * 4;
  UZ_REQUIRE(shm <= 64 * 1024, "%s: partial-row staging exceeds 64 KiB", fn);
  return UZ_OK;
}
static int ln_head_unroll(bool bwd, int K) { return (K == 1 && !bwd) ? 1 : 2; }
static int ln_head_grid(const uz_ln_desc* d, int K, bool bwd) {
  return grid_cap((long long)d->N * d->Ho * d->Wo, 4 * (64 / ln_head_lpt(d, K)) * ln_head_unroll(bwd, K), 8);
}
static void ln_head_args(const uz_ln_desc* d, LnHeadArgs* h) {
  LnArgs& a = h->ln;
  a.N = d->N; a.Ho = d->Ho; a.Wo = d->Wo; a.C = d->C; a.ldx = d->ldx; a.lddx = d->lddx;
  a.mode = d->mode; a.r = d->r; a.eps = d->eps;
  ln_geometry(d, &a);
}
template <bool BWD>
static void ln_head_launch(const uz_ln_desc* d, const LnHeadArgs& h, size_t shm, hipStream_t st) {
  const dim3 grid(ln_head_grid(d, h.K, BWD)), block(256);
  const int lpt = ln_head_lpt(d, h.K);
#define UZ_LNH(T) \
  do { \
    if (h.K == 1) hipLaunchKernelGGL((ln_head_kernel<T, BWD, 1, 3, (BWD ? 2 : 1)>), grid, block, shm, st, h, lpt); \
    else hipLaunchKernelGGL((ln_head_kernel<T, BWD, 4, 1, 2>), grid, block, shm, st, h, lpt); \
  } while (0)
  if (d->dtype == UZ_BF16) UZ_LNH(bf16_t);
  else UZ_LNH(float);
#undef UZ_LNH
}

extern "C" int uz_ln_head_fwd(const uz_ln_desc* d, const void* x, const float* gamma, const float* beta,
                              const float* w, const float* b, int K, float* logits, float* stats, void* stream) {
  const int rc = ln_head_check("uz_ln_head_fwd", d, K);
  if (rc != UZ_OK) return rc;
  UZ_REQUIRE(x && gamma && beta && w && logits && stats, "uz_ln_head_fwd: null pointer");
  UZ_REQUIRE((((uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)w) & 15) == 0 && ((uintptr_t)stats & 7) == 0,
             "uz_ln_head_fwd: gamma / beta / w must be 16-byte aligned, stats 8-byte aligned");
  LnHeadArgs h{};
  ln_head_args(d, &h);
  h.ln.x = x; h.ln.gamma = gamma; h.ln.beta = beta; h.ln.stats = stats;
  h.w = w; h.b = b; h.logits = logits; h.K = K;
  ln_head_launch<false>(d, h, 0, (hipStream_t)stream);
  UZ_LAUNCH_CHECK("uz_ln_head_fwd");
  return UZ_OK;
}

extern "C" long long uz_ln_head_bwd_workspace_bytes(const uz_ln_desc* d, int K) {
  const int rc = ln_head_check("uz_ln_head_bwd_workspace_bytes", d, K);
  if (rc != UZ_OK) return rc;
  return (long long)ln_head_grid(d, K, true) * (K * d->C + K) * (long long)sizeof(float);
}

extern "C" int uz_ln_head_bwd(const uz_ln_desc* d, const void* x, const float* gamma, const float* beta,
                              const float* w, int K, const float* stats, const float* dlogits, void* dx,
                              float* dgamma, float* dbeta, float* dw, float* db, float* workspace, void* stream) {
  const int rc = ln_head_check("uz_ln_head_bwd", d, K);
  if (rc != UZ_OK) return rc;
  const int vec = d->dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(x && gamma && beta && w && stats && dlogits && dx && dgamma && dbeta && dw && workspace,
             "uz_ln_head_bwd: null pointer");
  UZ_REQUIRE((((uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)w) & 15) == 0 && ((uintptr_t)stats & 7) == 0,
             "uz_ln_head_bwd: gamma / beta / w must be 16-byte aligned, stats 8-byte aligned");
  UZ_REQUIRE(d->lddx % vec == 0, "uz_ln_head_bwd: bad lddx");
  LnHeadArgs h{};
  ln_head_args(d, &h);
  h.ln.x = x; h.ln.gamma = gamma; h.ln.beta = beta; h.ln.stats = const_cast<float*>(stats);
  h.ln.dx = dx; h.ln.partial = workspace;
  h.w = w; h.dlogits = dlogits; h.K = K;
  const size_t shm = (size_t)4 * (64 / ln_head_lpt(d, K)) * (K * d->C + K) * sizeof(float);
  ln_head_launch<true>(d, h, shm, (hipStream_t)stream);
  UZ_LAUNCH_CHECK("uz_ln_head_bwd");
  hipLaunchKernelGGL(ln_head_finalize_kernel, dim3(uz_cdiv(d->C, 8)), dim3(1024), 0, (hipStream_t)stream,
                     (const float*)workspace, ln_head_grid(d, K, true), d->C, K, gamma, beta, w, dgamma, dbeta, dw, db);
  UZ_LAUNCH_CHECK("uz_ln_head_bwd (finalize)");
  return UZ_OK;
}

static int attn_check(const char* fn, const uz_winattn_desc* d) {
  UZ_REQUIRE(d != nullptr, "%s: null descriptor", fn);
  UZ_REQUIRE(d->dtype == UZ_F32 || d->dtype == UZ_BF16, "%s: bad dtype", fn);
  UZ_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->heads > 0 && d->C == d->heads * AD,
             "%s: needs head_dim 32 (C=%d, heads=%d)", fn, d->C, d->heads);
  UZ_REQUIRE(d->ws >= 1 && d->ws * d->ws <= AN && d->H % d->ws == 0 && d->W % d->ws == 0,
             "%s: window %d does not tile %dx%d (or exceeds 8x8)", fn, d->ws, d->H, d->W);
  UZ_REQUIRE(d->shift >= 0 && d->shift < d->ws && d->Nt >= d->ws * d->ws, "%s: bad shift / tau size", fn);
  const int vec = d->dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(d->ldq % vec == 0 && d->ldq >= 3 * d->C && d->ldo % vec == 0 && d->ldo >= d->C, "%s: bad strides", fn);
  UZ_REQUIRE((long long)d->B * d->H * d->W < (1LL << 31), "%s: too many tokens", fn);
  return UZ_OK;
}

// Workgroups of the window-attention kernels stay resident for the whole launch (each walks its share of the
// windows of one head), so the grid must FIT: one workgroup more than the chip holds doubles the run time.
// slots = resident workgroups per CU of the kernel; windows are dealt evenly (per-workgroup count first).
static int attn_grid_fit(const uz_winattn_desc* d, long long units_x, int slots_per_cu) {
  const char* e = uz_ablate_env("UZ_ATTN_GX");   // measurement hook (tools/attn_bench.py)
  if (e && atoi(e) > 0) return (int)(atoi(e) < units_x ? atoi(e) : units_x);
  long long cap = (long long)UZ_NUM_CU * slots_per_cu / d->heads;
  if (cap < 1) cap = 1;
  const long long per = (units_x + cap - 1) / cap;
  const long long g = (units_x + per - 1) / per;
  return (int)(g < 1 ? 1 : g);
}
static int attn_grid_x(const uz_winattn_desc* d, int slots_per_cu) {
  return attn_grid_fit(d, (long long)d->B * (d->H / d->ws) * (d->W / d->ws), slots_per_cu);
}
// resident workgroups per CU (registers / LDS of the kernels below; check with the ISA when they change)
constexpr int ATTN_SLOTS_FWD = 2;        // winattn_fwd_kernel: 186 VGPRs, 54 KB LDS
constexpr int ATTN_SLOTS_FWD_MFMA = 1;   // winattn_fwd_mfma_kernel: 325 VGPRs
constexpr int ATTN_SLOTS_FWD_MFMA2 = 3;  // winattn_fwd_mfma2_kernel: <= 168 VGPRs, 48 KB LDS
constexpr int ATTN_SLOTS_BWD = 1;        // winattn_bwd_kernel: 152 KB LDS
constexpr int ATTN_SLOTS_BWD_MFMA = 2;   // winattn_bwd_mfma_kernel: 256 VGPRs, 65 KB LDS

// forward matrix-core kernel (bf16): one wave per (window, head), four per workgroup
static int attn_mfma_grid_x(const uz_winattn_desc* d) {
  const long long nwin = (long long)d->B * (d->H / d->ws) * (d->W / d->ws);
  return attn_grid_fit(d, (nwin + 3) / 4, ATTN_SLOTS_FWD_MFMA);
}
static bool attn_bwd_mfma(const uz_winattn_desc* d) { return d->dtype == UZ_BF16 && !(uz_tune_flags() & 0x2000); }

extern "C" int uz_winattn_fwd(const uz_winattn_desc* d, const void* qkv, const float* tau, const float* bias,
                              void* out, float* lse, void* stream) {
  const int rc = attn_check("uz_winattn_fwd", d);
  if (rc != UZ_OK) return rc;
  UZ_REQUIRE(qkv && tau && bias && out && lse, "uz_winattn_fwd: null pointer");
  AttnArgs a{};
  a.qkv = qkv; a.out = out; a.lse = lse; a.tau = tau; a.bias = bias;
  a.B = d->B; a.H = d->H; a.W = d->W; a.C = d->C; a.heads = d->heads; a.ws = d->ws; a.shift = d->shift; a.Nt = d->Nt;
  a.ldq = d->ldq; a.ldo = d->ldo; a.scale = d->scale; a.flags = uz_tune_flags();
  const dim3 grid(attn_grid_x(d, ATTN_SLOTS_FWD), d->heads), block(256);
  if (d->dtype == UZ_BF16 && !(uz_tune_flags() & 0x1000)) {
    // matrix-core path: one wave per (window, head), four per workgroup
    if (uz_tune_flags() & 0x4000)   // the one-wave-per-unit kernel, kept for A/B measurements
      hipLaunchKernelGGL(winattn_fwd_mfma_kernel, dim3(attn_mfma_grid_x(d), d->heads), dim3(256), 0, (hipStream_t)stream, a);
    else
      hipLaunchKernelGGL(winattn_fwd_mfma2_kernel, dim3(attn_grid_x(d, ATTN_SLOTS_FWD_MFMA2), d->heads), dim3(256), 0,
                         (hipStream_t)stream, a);
  } else if (d->dtype == UZ_BF16) {
    hipLaunchKernelGGL((winattn_fwd_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, a);
  } else {
    hipLaunchKernelGGL((winattn_fwd_kernel<float>), grid, block, 0, (hipStream_t)stream, a);
  }
  UZ_LAUNCH_CHECK("uz_winattn_fwd");
  return UZ_OK;
}

extern "C" int uz_winattn_bwd_rows(const uz_winattn_desc* d) {
  const int rc = attn_check("uz_winattn_bwd_rows", d);
  if (rc != UZ_OK) return rc;
  return attn_grid_x(d, attn_bwd_mfma(d) ? ATTN_SLOTS_BWD_MFMA : ATTN_SLOTS_BWD);
}

extern "C" int uz_winattn_bwd(const uz_winattn_desc* d, const void* qkv, const float* tau, const float* bias,
                              const void* out, const float* lse, const void* dout, int lddo, void* dqkv, int lddq,
                              float* partial, void* stream) {
  const int rc = attn_check("uz_winattn_bwd", d);
  if (rc != UZ_OK) return rc;
  const int vec = d->dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(qkv && tau && bias && out && lse && dout && dqkv && partial, "uz_winattn_bwd: null pointer");
  UZ_REQUIRE(lddo % vec == 0 && lddo >= d->C && lddq % vec == 0 && lddq >= 3 * d->C, "uz_winattn_bwd: bad strides");
  AttnArgs a{};
  a.qkv = qkv; a.out = const_cast<void*>(out); a.lse = const_cast<float*>(lse); a.tau = tau; a.bias = bias;
  a.dout = dout; a.dqkv = dqkv; a.partial = partial;
  a.B = d->B; a.H = d->H; a.W = d->W; a.C = d->C; a.heads = d->heads; a.ws = d->ws; a.shift = d->shift; a.Nt = d->Nt;
  a.ldq = d->ldq; a.ldo = d->ldo; a.lddo = lddo; a.lddq = lddq; a.scale = d->scale; a.flags = uz_tune_flags();
  const dim3 grid(attn_grid_x(d, attn_bwd_mfma(d) ? ATTN_SLOTS_BWD_MFMA : ATTN_SLOTS_BWD), d->heads), block(256);
  if (attn_bwd_mfma(d)) {
    hipLaunchKernelGGL(winattn_bwd_mfma_kernel, grid, block, 0, (hipStream_t)stream, a);
  } else if (d->dtype == UZ_BF16) {
    hipLaunchKernelGGL((winattn_bwd_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, a);
  } else {
    hipLaunchKernelGGL((winattn_bwd_kernel<float>), grid, block, 0, (hipStream_t)stream, a);
  }
  UZ_LAUNCH_CHECK("uz_winattn_bwd");
  return UZ_OK;
}

static int cpb_check(const char* fn, int R, int hidden, int heads) {
  UZ_REQUIRE(R > 0 && hidden > 0 && hidden <= CPB_MAXHID && heads > 0 && heads <= CPB_MAXH,
             "%s: bad shape (heads <= %d, hidden <= %d)", fn, CPB_MAXH, CPB_MAXHID);
  return UZ_OK;
}

extern "C" int uz_cpb_fwd(const float* idx, const float* w1, const float* b1, const float* w2, const float* b2, int R,
                          int hidden, int heads, float* bias, void* stream) {
  const int rc = cpb_check("uz_cpb_fwd", R, hidden, heads);
  if (rc != UZ_OK) return rc;
  UZ_REQUIRE(idx && w1 && b1 && w2 && b2 && bias, "uz_cpb_fwd: null pointer");
  hipLaunchKernelGGL(cpb_fwd_kernel, dim3(uz_cdiv(R, 256), heads), dim3(256), 0, (hipStream_t)stream, idx, w1, b1, w2,
                     b2, R, hidden, heads, bias);
  UZ_LAUNCH_CHECK("uz_cpb_fwd");
  return UZ_OK;
}

extern "C" int uz_cpb_bwd(const float* idx, const float* w1, const float* b1, const float* w2, const float* G, int R,
                          int hidden, int heads, float* dw1, float* db1, float* dw2, float* db2, void* stream) {
  const int rc = cpb_check("uz_cpb_bwd", R, hidden, heads);
  if (rc != UZ_OK) return rc;
  UZ_REQUIRE(idx && w1 && b1 && w2 && G && dw1 && db1 && dw2 && db2, "uz_cpb_bwd: null pointer");
  hipLaunchKernelGGL(cpb_bwd_kernel, dim3(uz_cdiv(hidden, CPB_KB)), dim3(64 * CPB_NW), 0, (hipStream_t)stream, idx, w1, b1, w2, G,
                     R, hidden, heads, dw1, db1, dw2, db2);
  UZ_LAUNCH_CHECK("uz_cpb_bwd");
  return UZ_OK;
}

static long long cpb_item_ws_floats(const uz_cpb_item* m) {
  const long long nsub = 512 / m->hidden, RB = (m->R + CPB_ROWS - 1) / CPB_ROWS;
  return RB * nsub * (long long)(m->heads + 3) * m->hidden + RB * m->heads;
}

static int cpb_batch_check(const char* fn, const uz_cpb_item* items, int n, bool bwd) {
  UZ_REQUIRE(items != nullptr && n > 0, "%s: empty batch", fn);
  for (int i = 0; i < n; ++i) {
    const uz_cpb_item* m = items + i;
    const int rc = cpb_check(fn, m->R, m->hidden, m->heads);
    if (rc != UZ_OK) return rc;
    UZ_REQUIRE(m->idx && m->w1 && m->b1 && m->w2, "%s: item %d: null pointer", fn, i);
    if (bwd) UZ_REQUIRE(m->G && m->dw1 && m->db1 && m->dw2 && m->db2, "%s: item %d: null gradient pointer", fn, i);
    else UZ_REQUIRE(m->b2 && m->bias, "%s: item %d: null pointer", fn, i);
  }
  return UZ_OK;
}

extern "C" int uz_cpb_fwd_batched(const uz_cpb_item* items, int n, void* stream) {
  const int rc = cpb_batch_check("uz_cpb_fwd_batched", items, n, false);
  if (rc != UZ_OK) return rc;
  for (int i0 = 0; i0 < n; i0 += CPB_MAXB) {
    const int nb = n - i0 < CPB_MAXB ? n - i0 : CPB_MAXB;
    CpbBatch b{};
    int rmax = 0, hmax = 0;
    for (int i = 0; i < nb; ++i) {
      b.it[i] = items[i0 + i];
      rmax = items[i0 + i].R > rmax ? items[i0 + i].R : rmax;
      hmax = items[i0 + i].heads > hmax ? items[i0 + i].heads : hmax;
    }
    hipLaunchKernelGGL(cpb_fwd_batched_kernel, dim3(uz_cdiv(rmax, 256), hmax, nb), dim3(256), 0, (hipStream_t)stream, b);
    UZ_LAUNCH_CHECK("uz_cpb_fwd_batched");
  }
  return UZ_OK;
}

extern "C" long long uz_cpb_bwd_batched_workspace_bytes(const uz_cpb_item* items, int n) {
  UZ_REQUIRE(items != nullptr && n > 0, "uz_cpb_bwd_batched_workspace_bytes: empty batch");
  long long tot = 0;
  for (int i = 0; i < n; ++i) {
    const int rc = cpb_check("uz_cpb_bwd_batched_workspace_bytes", items[i].R, items[i].hidden, items[i].heads);
    if (rc != UZ_OK) return rc;
    tot += cpb_item_ws_floats(items + i);
  }
  return tot * (long long)sizeof(float);
}

extern "C" int uz_cpb_bwd_batched(const uz_cpb_item* items, int n, float* workspace, void* stream) {
  const int rc = cpb_batch_check("uz_cpb_bwd_batched", items, n, true);
  if (rc != UZ_OK) return rc;
  UZ_REQUIRE(workspace != nullptr, "uz_cpb_bwd_batched: null workspace");
  long long off = 0;
  for (int i0 = 0; i0 < n; i0 += CPB_MAXB) {
    const int nb = n - i0 < CPB_MAXB ? n - i0 : CPB_MAXB;
    CpbBatch b{};
    int rmax = 0, omax = 0;
    for (int i = 0; i < nb; ++i) {
      const uz_cpb_item& m = items[i0 + i];
      b.it[i] = m;
      b.off[i] = off;
      off += cpb_item_ws_floats(&m);
      rmax = m.R > rmax ? m.R : rmax;
      const int outs = (m.heads + 3) * m.hidden + m.heads;
      omax = outs > omax ? outs : omax;
    }
    hipLaunchKernelGGL(cpb_bwd_batched_kernel, dim3(uz_cdiv(rmax, CPB_ROWS), nb), dim3(512), 0, (hipStream_t)stream, b, workspace);
    UZ_LAUNCH_CHECK("uz_cpb_bwd_batched");
    hipLaunchKernelGGL(cpb_bwd_finalize_kernel, dim3(uz_cdiv(omax, 256), nb), dim3(256), 0, (hipStream_t)stream, b,
                       (const float*)workspace);
    UZ_LAUNCH_CHECK("uz_cpb_bwd_batched (finalize)");
  }
  return UZ_OK;
}
