// Weight-gradient kernel for gfx950 (MI355X): out[i][j][tap] (+)= sum_p L[p][i] * R[pix(p,tap)][j]
//
// Stands in for the weight-gradient half of autograd for nn.Conv2d k3/k1 and
// nn.ConvTranspose2d k2s2 (reference call sites: unet_zoo/models/common_layers.py:28,31,104,125;
// entered from loss.backward(), unet_zoo/utils/training_loop.py:119).
//
// GEMM view per tap: D[BI x BJ] += L^T[BI x pixels] * R[pixels x BJ]; the reduction (pixel) index
// is the slow index of both NHWC operands, so tiles are staged pixel-major in LDS ([k][channel])
// and fragments are read
//   bf16: ds_read_b64_tr_b16 (hardware transpose: each lane receives 4 consecutive pixels of its
//         channel), rows padded by 64 B so the four pixel rows of a read land on distinct banks;
//   fp32: ds_read_b32 (the 32x32x2 fp32 MFMA wants one element per lane; 32 lanes = 32
//         consecutive channels = conflict-free).
// Pixels are split over gridDim.z; every split writes its own fp32 slab [z][tap][Ci][Cj] with
// coalesced plain stores and a second kernel sums the slabs in a fixed order while transposing to
// the parameter layout (deterministic; float atomics at a 36-byte lane stride ran 17x slower).
#include "uz_common.h"

namespace {

struct WgradArgs {
  const void* L;
  const void* R;
  float* out;
  int P, H, W, Hr, Wr, Ci, ldl, Cj, ldr, ntaps, mode, dil, split, chunk, tiles_j;
};

template <typename T> struct WgCfg;
template <> struct WgCfg<bf16_t> {
  static constexpr int BKP = 64;  // pixels per K-step
  static constexpr int PAD = 64;  // bytes of row padding
};
template <> struct WgCfg<float> {
  static constexpr int BKP = 32;
  static constexpr int PAD = 0;
};

template <typename T, int BI, int BJ>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradArgs a) {
  constexpr int VEC = ElemTraits<T>::VEC;
  constexpr int BKP = WgCfg<T>::BKP;
  constexpr int CPR_I = BI / VEC, CPR_J = BJ / VEC;        // 16-byte chunks per pixel row
  constexpr int RPP_I = 256 / CPR_I, RPP_J = 256 / CPR_J;  // pixel rows per pass
  constexpr int NI = BKP / RPP_I, NJ = BKP / RPP_J;
  static_assert(NI >= 1 && NJ >= 1, "tile too wide for BKP");
  constexpr int RS_I = BI * (int)sizeof(T) + WgCfg<T>::PAD;
  constexpr int RS_J = BJ * (int)sizeof(T) + WgCfg<T>::PAD;
  constexpr int L_BYTES = BKP * RS_I, R_BYTES = BKP * RS_J, STAGE = L_BYTES + R_BYTES;
  constexpr int WTI = BI / 2, WTJ = BJ / 2, TI = WTI / 32, TJ = WTJ / 32;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const int ti0 = (blockIdx.x / a.tiles_j) * BI, tj0 = (blockIdx.x % a.tiles_j) * BJ;
  const int tap = blockIdx.y;
  const int pbeg = blockIdx.z * a.chunk;
  const int pend = (pbeg + a.chunk < a.P) ? pbeg + a.chunk : a.P;
  const T* __restrict__ Lg = static_cast<const T*>(a.L);
  const T* __restrict__ Rg = static_cast<const T*>(a.R);

  int dy = 0, dx = 0;
  if (a.mode == UZ_TAPS_CONV || a.mode == UZ_TAPS_CONV_UP2) {
    if (a.ntaps == 9) {
      const int ty = tap / 3;
      dy = (ty - 1) * a.dil;
      dx = (tap - 3 * ty - 1) * a.dil;
    }
  } else {
    dy = tap >> 1;
    dx = tap & 1;
  }

  // L rows handled by this thread
  const int lcI = tid % CPR_I, lrI = tid / CPR_I;
  const bool cokI = ti0 + lcI * VEC < a.Ci;
  // R rows handled by this thread: keep (img, h, w) per row, advanced by BKP pixels per step
  const int lcJ = tid % CPR_J, lrJ = tid / CPR_J;
  const bool cokJ = tj0 + lcJ * VEC < a.Cj;
  int rn[NJ], rh[NJ], rw[NJ];
  const int HW = a.H * a.W;
#pragma unroll
  for (int i = 0; i < NJ; ++i) {
    const int p = pbeg + lrJ + RPP_J * i;
    const int img = p / HW;
    const int rem = p - img * HW;
    rn[i] = img;
    rh[i] = rem / a.W;
    rw[i] = rem - rh[i] * a.W;
  }

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  Vec16<T> rl[NI], rr[NJ];
  int pk = pbeg;  // first pixel of the K-step being loaded

  auto load_step = [&]() {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int p = pk + lrI + RPP_I * i;
      const bool ok = cokI && p < pend;
      rl[i] = ok ? ld16(Lg + (size_t)p * a.ldl + ti0 + lcI * VEC) : zero16<T>();
    }
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
      const int p = pk + lrJ + RPP_J * i;
      bool ok = cokJ && p < pend;
      size_t pix;
      if (a.mode == UZ_TAPS_CONV) {
        const int hh = rh[i] + dy, ww = rw[i] + dx;
        ok = ok && (unsigned)hh < (unsigned)a.Hr && (unsigned)ww < (unsigned)a.Wr;
        pix = ((size_t)rn[i] * a.Hr + hh) * a.Wr + ww;
      } else if (a.mode == UZ_TAPS_CONV_UP2) {  // R lives at (H/2, W/2), read through nearest x2
        const int hh = rh[i] + dy, ww = rw[i] + dx;
        ok = ok && (unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)a.W;
        pix = ((size_t)rn[i] * a.Hr + (hh >> 1)) * a.Wr + (ww >> 1);
      } else {
        pix = ((size_t)rn[i] * a.Hr + 2 * rh[i] + dy) * a.Wr + 2 * rw[i] + dx;
      }
      rr[i] = ok ? ld16(Rg + pix * a.ldr + tj0 + lcJ * VEC) : zero16<T>();
      // advance this row's coordinates to the next K-step
      rw[i] += BKP;
      while (rw[i] >= a.W) {
        rw[i] -= a.W;
        if (++rh[i] == a.H) {
          rh[i] = 0;
          ++rn[i];
        }
      }
    }
    pk += BKP;
  };
  auto store_step = [&](int buf) {
    char* sL = smem + buf * STAGE;
    char* sR = sL + L_BYTES;
#pragma unroll
    for (int i = 0; i < NI; ++i)
      *reinterpret_cast<Vec16<T>*>(sL + (lrI + RPP_I * i) * RS_I + lcI * 16) = rl[i];
#pragma unroll
    for (int i = 0; i < NJ; ++i)
      *reinterpret_cast<Vec16<T>*>(sR + (lrJ + RPP_J * i) * RS_J + lcJ * 16) = rr[i];
  };

  const int nk = (pend - pbeg + BKP - 1) / BKP;
  if (nk > 0) {
    load_step();
    store_step(0);
  }
  __syncthreads();
  for (int kb = 0; kb < nk; ++kb) {
    const bool more = kb + 1 < nk;
    if (more) load_step();
    const char* sL = smem + (kb & 1) * STAGE;
    const char* sR = sL + L_BYTES;
    if constexpr (sizeof(T) == 2) {
      // lane -> (16-lane group g, row q, column quad p4) of the transposed 4x16 block read
      const int g = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3;
      const int krow = 8 * (g >> 1) + q;
      const int ccol = 16 * (g & 1) + 4 * p4;
      typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;
#pragma unroll
      for (int ks = 0; ks < BKP / 16; ++ks) {
        bf16x8 af[TI], bfr[TJ];
#pragma unroll
        for (int i = 0; i < TI; ++i) {
          const char* p0 = sL + (ks * 16 + krow) * RS_I + (wi * WTI + i * 32 + ccol) * 2;
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p0));
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p0 + 4 * RS_I));
          af[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          const char* p0 = sR + (ks * 16 + krow) * RS_J + (wj * WTJ + j * 32 + ccol) * 2;
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p0));
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p0 + 4 * RS_J));
          bfr[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll 4
      for (int s = 0; s < BKP / 2; ++s) {
        const int k = 2 * s + lh;
        float af[TI], bfr[TJ];
#pragma unroll
        for (int i = 0; i < TI; ++i)
          af[i] = *reinterpret_cast<const float*>(sL + k * RS_I + (wi * WTI + i * 32 + l31) * 4);
#pragma unroll
        for (int j = 0; j < TJ; ++j)
          bfr[j] = *reinterpret_cast<const float*>(sR + k * RS_J + (wj * WTJ + j * 32 + l31) * 4);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
    }
    if (more) store_step((kb + 1) & 1);
    __syncthreads();
  }

  // partial slab [split][tap][Ci][Cj]: lanes 0..31 write 32 consecutive j (128 contiguous bytes)
  float* slab = a.out + ((size_t)blockIdx.z * a.ntaps + tap) * (size_t)a.Ci * a.Cj;
#pragma unroll
  for (int i = 0; i < TI; ++i) {
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int cj = tj0 + wj * WTJ + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ti0 + wi * WTI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (ci < a.Ci && cj < a.Cj) slab[(size_t)ci * a.Cj + cj] = acc[i][j][r];
      }
    }
  }
}

// (Round 3 measured the alternative of deferring these reductions to ONE batched launch per backward range: 23 launches
// of 5-17 us -> one of 442 us.  Worse: a layer's 40-75 MB of slabs are still in the 256 MB Infinity Cache when its own
// reduction follows the kernel that wrote them; deferred, all 1.2 GB come back from HBM.  The per-layer launch stays.)
// out[i][j][tap] = sum_z slab[z][tap][i][j]: thread (x, zg) owns element (i,j) = blockIdx.x*64 + x
// and the splits z = zg, zg+4, ...; reads are contiguous along j, each thread finally writes its
// element's ntaps values (ntaps*4 contiguous bytes; a wave writes one contiguous span).
// The same sums with the nine taps as part of the row: slab z is ONE contiguous row of NT * Ci * Cj floats, a thread owns
// four consecutive columns (16-byte loads, 1 KB per wave and row) and the splits z = zg, zg + ZG, ...  For the small
// layers (64 x 64 channels: 256 slabs of 147 KB, written a moment ago and still in the Infinity Cache) the kernel above
// ran 64 workgroups -- a quarter of the chip -- at 256-byte pieces; this one runs NT * Ci * Cj / 256 = 144.
template <int NT, int ZG>
__global__ __launch_bounds__(64 * ZG) void wgrad_reduce_flat_kernel(const float* __restrict__ slab, int split, long long CiCj,
                                                                   float* __restrict__ out) {
  __shared__ float4 red[ZG][64];
  const long long row = (long long)NT * CiCj;
  const int x = threadIdx.x & 63, zg = threadIdx.x >> 6;
  const long long col = ((long long)blockIdx.x * 64 + x) * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col < row) {
    int z = zg;
    for (; z + ZG < split; z += 2 * ZG) {   // two splits' loads in flight per pass
      const float4 a0 = *reinterpret_cast<const float4*>(slab + (size_t)z * row + col);
      const float4 a1 = *reinterpret_cast<const float4*>(slab + (size_t)(z + ZG) * row + col);
      acc.x = (acc.x + a0.x) + a1.x;
      acc.y = (acc.y + a0.y) + a1.y;
      acc.z = (acc.z + a0.z) + a1.z;
      acc.w = (acc.w + a0.w) + a1.w;
    }
    if (z < split) {
      const float4 a0 = *reinterpret_cast<const float4*>(slab + (size_t)z * row + col);
      acc.x += a0.x, acc.y += a0.y, acc.z += a0.z, acc.w += a0.w;
    }
  }
  red[zg][x] = acc;
  __syncthreads();
  if (zg == 0 && col < row) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < ZG; ++k) {
      const float4 r = red[k][x];
      v[0] += r.x, v[1] += r.y, v[2] += r.z, v[3] += r.w;
    }
    const int t = (int)(col / CiCj);   // Ci * Cj is a multiple of 4: the four columns are one tap's
    const long long e = col - (long long)t * CiCj;
#pragma unroll
    for (int i = 0; i < 4; ++i) out[(e + i) * NT + t] = v[i];
  }
}

template <int NT, int ZG>
__global__ __launch_bounds__(64 * ZG) void wgrad_reduce_kernel(const float* __restrict__ slab, int split,
                                                              long long CiCj, float* __restrict__ out,
                                                              long long out_b) {
  __shared__ float red[ZG][64][NT + 1];
  slab += (size_t)blockIdx.y * split * NT * CiCj;   // uz_wgrad_batched: blockIdx.y = problem
  out += (size_t)blockIdx.y * out_b;
  const int x = threadIdx.x & 63, zg = threadIdx.x >> 6;
  const long long e = (long long)blockIdx.x * 64 + x;
  float acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = 0.f;
  if (e < CiCj) {
    // two splits' loads in flight per pass (same summation order): a pass is one memory round trip
    int z = zg;
    for (; z + ZG < split; z += 2 * ZG) {
      float a0[NT], a1[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) a0[t] = slab[((size_t)z * NT + t) * CiCj + e];
#pragma unroll
      for (int t = 0; t < NT; ++t) a1[t] = slab[((size_t)(z + ZG) * NT + t) * CiCj + e];
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = (acc[t] + a0[t]) + a1[t];
    }
    if (z < split) {
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] += slab[((size_t)z * NT + t) * CiCj + e];
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) red[zg][x][t] = acc[t];
  __syncthreads();
  if (zg == 0 && e < CiCj) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < ZG; ++k) v += red[k][x][t];
      out[e * NT + t] = v;
    }
  }
}

struct Plan {
  int b;  // tile edge: 64 or 128
  int tiles_i, tiles_j, split, chunk;
};

int make_plan(const uz_wgrad_desc* d, Plan* p) {
  UZ_REQUIRE(d != nullptr, "uz_wgrad: null descriptor");
  UZ_REQUIRE(d->dtype == UZ_F32 || d->dtype == UZ_BF16, "uz_wgrad: bad dtype %d", d->dtype);
  const int vec = d->dtype == UZ_BF16 ? 8 : 4;
  const int bkp = d->dtype == UZ_BF16 ? 64 : 32;
  UZ_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Ci > 0 && d->Cj > 0, "uz_wgrad: bad shape");
  UZ_REQUIRE(d->Ci % vec == 0 && d->Cj % vec == 0, "uz_wgrad: channels must be multiples of %d", vec);
  UZ_REQUIRE(d->ldl % vec == 0 && d->ldr % vec == 0 && d->ldl >= d->Ci && d->ldr >= d->Cj,
             "uz_wgrad: bad leading dimension");
  if (d->taps_mode == UZ_TAPS_CONV) {
    UZ_REQUIRE(d->ntaps == 1 || d->ntaps == 9, "uz_wgrad: ntaps=%d", d->ntaps);
    UZ_REQUIRE(d->Hr == d->H && d->Wr == d->W && d->dil >= 1, "uz_wgrad: conv taps need Hr==H");
  } else if (d->taps_mode == UZ_TAPS_CONV_UP2) {
    UZ_REQUIRE(d->ntaps == 9 && d->dil == 1 && d->Hr * 2 == d->H && d->Wr * 2 == d->W,
               "uz_wgrad: upsampled R needs ntaps=9, dil=1, Hr=H/2, Wr=W/2");
  } else if (d->taps_mode == UZ_TAPS_CONV_S2) {
    UzWgrad2Plan s2p_;
    UZ_REQUIRE(d->ntaps == 9 && d->H == (d->Hr + 1) / 2 && d->W == (d->Wr + 1) / 2,
               "uz_wgrad: stride-2 taps need ntaps=9, H = ceil(Hr/2), W = ceil(Wr/2)");
    UZ_REQUIRE(uz_wgrad3x3_plan(d, &s2p_), "uz_wgrad: stride-2 taps need the LDS-DMA kernel (bf16, W in {16, 32, 64k}, channel multiples of 8)");
  } else {
    UZ_REQUIRE(d->taps_mode == UZ_TAPS_GATHER2X2 && d->ntaps == 4 && (d->Hr == 2 * d->H || d->Hr == 2 * d->H + 1) &&
                   (d->Wr == 2 * d->W || d->Wr == 2 * d->W + 1),
               "uz_wgrad: gather2x2 needs ntaps=4, Hr in {2H, 2H+1}, Wr in {2W, 2W+1}");
  }
  const long long P = (long long)d->N * d->H * d->W;
  UZ_REQUIRE(P < (1LL << 31) && (long long)d->N * d->Hr * d->Wr < (1LL << 31), "uz_wgrad: too large");
  p->b = (d->Ci > 64 && d->Cj > 64) ? 128 : 64;
  p->tiles_i = uz_cdiv(d->Ci, p->b);
  p->tiles_j = uz_cdiv(d->Cj, p->b);
  const long long base = (long long)p->tiles_i * p->tiles_j * d->ntaps;
  long long split = (4 * UZ_NUM_CU + base - 1) / base;
  long long max_split = P / (4LL * bkp) > 0 ? P / (4LL * bkp) : 1;
  if (max_split > 64) max_split = 64;
  if (split > max_split) split = max_split;
  if (split < 1) split = 1;
  long long chunk = (P + split - 1) / split;
  chunk = ((chunk + bkp - 1) / bkp) * bkp;
  split = (P + chunk - 1) / chunk;
  p->split = (int)split;
  p->chunk = (int)chunk;
  return UZ_OK;
}

template <typename T> int launch(const Plan& p, const WgradArgs& a, hipStream_t s) {
  dim3 grid(p.tiles_i * p.tiles_j, a.ntaps, p.split), block(256);
  if (p.b == 64) {
    hipLaunchKernelGGL((wgrad_kernel<T, 64, 64>), grid, block, 0, s, a);
  } else {
    hipLaunchKernelGGL((wgrad_kernel<T, 128, 128>), grid, block, 0, s, a);
  }
  UZ_LAUNCH_CHECK("uz_wgrad");
  return UZ_OK;
}

}  // namespace

extern "C" int uz_wgrad_split(const uz_wgrad_desc* d) {
  Plan p;
  const int rc = make_plan(d, &p);
  if (rc != UZ_OK) return rc;
  UzWgrad2Plan p2;
  if (uz_wgrad3x3_plan(d, &p2)) return p2.nslabs;
  return p.split;
}

extern "C" int uz_wgrad_kernel_name(const uz_wgrad_desc* d, char* buf, int cap) {
  Plan p;
  const int rc = make_plan(d, &p);
  if (rc != UZ_OK) return rc;
  UZ_REQUIRE(buf != nullptr && cap > 0, "uz_wgrad_kernel_name: no buffer");
  UzWgrad2Plan p2;
  const char* dt = d->dtype == UZ_BF16 ? "bf16" : "fp32";
  int n;
  if (!uz_wgrad3x3_plan(d, &p2)) {
    n = snprintf(buf, cap, "wgrad_%s_%dx%d", dt, p.b, p.b);
  } else if (p2.v9 == 2) {
    n = snprintf(buf, cap, "wgrad_g4_bf16_%dx64_4tap", p2.bi);
  } else if (p2.v9) {
    n = snprintf(buf, cap, "%s", uz_wgrad9_name(p2));
  } else {
    const char* tile = p2.wide9 == 1 ? "128x64" : (p2.wide9 == 2 ? "64x128" : (p2.big ? "128x128" : "64x64"));
    if (p2.gather == 3) n = snprintf(buf, cap, "wgrad3x3_bf16_%s_dilated9", tile);
    else if (p2.gather) n = snprintf(buf, cap, "wgrad3x3_bf16_%s_gather%d", tile, d->ntaps);
    else if (p2.one_tap) n = snprintf(buf, cap, "wgrad3x3_bf16_%s_1tap", tile);
    else n = snprintf(buf, cap, "wgrad3x3_bf16_%s_%s", tile, (p2.big && !p2.wide9) ? "3tap" : "9tap");
  }
  return n < cap ? n : cap - 1;
}

extern "C" long long uz_wgrad_workspace_bytes(const uz_wgrad_desc* d) {
  Plan p;
  const int rc = make_plan(d, &p);
  if (rc != UZ_OK) return rc;
  UzWgrad2Plan p2;
  const long long nslabs = uz_wgrad3x3_plan(d, &p2) ? p2.nslabs : p.split;
  return nslabs * d->ntaps * d->Ci * d->Cj * (long long)sizeof(float);
}

// phase 0: both launches; 1: the main kernel (partial slabs into the workspace); 2: the fixed-order slab reduction
static int wgrad_phases(const uz_wgrad_desc* d, const void* L, const void* R, float* out, void* workspace, void* stream,
                        int phase, const UzXf* xf = nullptr) {
  Plan p;
  const int rc = make_plan(d, &p);
  if (rc != UZ_OK) return rc;
  UZ_REQUIRE(L && R && out && workspace, "uz_wgrad: null pointer");
  UZ_REQUIRE(((uintptr_t)L & 15) == 0 && ((uintptr_t)R & 15) == 0, "uz_wgrad: L / R must be 16-byte aligned");
  WgradArgs a;
  a.L = L;
  a.R = R;
  a.out = static_cast<float*>(workspace);
  a.P = d->N * d->H * d->W;
  a.H = d->H;
  a.W = d->W;
  a.Hr = d->Hr;
  a.Wr = d->Wr;
  a.Ci = d->Ci;
  a.ldl = d->ldl;
  a.Cj = d->Cj;
  a.ldr = d->ldr;
  a.ntaps = d->ntaps;
  a.mode = d->taps_mode;
  a.dil = d->dil;
  a.split = p.split;
  a.chunk = p.chunk;
  a.tiles_j = p.tiles_j;
  hipStream_t s = static_cast<hipStream_t>(stream);
  UzWgrad2Plan p2;
  int nslabs = p.split;
  int rc2;
  const bool lds_dma = uz_wgrad3x3_plan(d, &p2) != 0;
  if (lds_dma) nslabs = p2.nslabs;
  if (phase != 2) {
    UZ_REQUIRE(xf == nullptr || (lds_dma && p2.v9 == 1 && p2.bi == 64), "uz_wgrad_xf: not a problem of the row-walk kernel (ask uz_wgrad_xf_supported)");
    if (lds_dma) rc2 = uz_wgrad3x3_launch(d, p2, L, R, static_cast<float*>(workspace), s, 1, 0, 0, 0, 1, 0, 0, xf);
    else rc2 = d->dtype == UZ_BF16 ? launch<bf16_t>(p, a, s) : launch<float>(p, a, s);
    if (rc2 != UZ_OK) return rc2;
  }
  if (phase == 1) return UZ_OK;
  const long long cicj = (long long)d->Ci * d->Cj;
  const dim3 grid((unsigned)((cicj + 63) / 64));
  const float* slab = static_cast<const float*>(workspace);
  if (d->ntaps == 9) {
    if (nslabs >= 32 && cicj % 4 == 0 && (cicj <= 64 * 128 || uz_ablate_env("UZ_RED_FLAT_ALL")) && !(uz_tune_flags() & 0x100))
      hipLaunchKernelGGL((wgrad_reduce_flat_kernel<9, 16>), dim3((unsigned)((9 * cicj / 4 + 63) / 64)), dim3(1024), 0, s, slab,
                         nslabs, cicj, out);
    else if (nslabs >= 32)
      hipLaunchKernelGGL((wgrad_reduce_kernel<9, 16>), grid, dim3(1024), 0, s, slab, nslabs, cicj, out, 0LL);
    else
      hipLaunchKernelGGL((wgrad_reduce_kernel<9, 4>), grid, dim3(256), 0, s, slab, nslabs, cicj, out, 0LL);
  } else if (d->ntaps == 4) {
    hipLaunchKernelGGL((wgrad_reduce_kernel<4, 4>), grid, dim3(256), 0, s, slab, nslabs, cicj, out, 0LL);
  } else {
    hipLaunchKernelGGL((wgrad_reduce_kernel<1, 4>), grid, dim3(256), 0, s, slab, nslabs, cicj, out, 0LL);
  }
  UZ_LAUNCH_CHECK("uz_wgrad(reduce)");
  return UZ_OK;
}

extern "C" int uz_wgrad(const uz_wgrad_desc* d, const void* L, const void* R, float* out, void* workspace, void* stream) {
  return wgrad_phases(d, L, R, out, workspace, stream, 0);
}

// ---- weight gradient whose R operand is read through the BatchNorm + ReLU in front of the layer (include/unetzoo_hip.h) ---
extern "C" int uz_wgrad_xf_supported(const uz_wgrad_desc* d) {
  UzWgrad2Plan p2;
  if (d == nullptr || d->dtype != UZ_BF16 || d->ntaps != 9) return 0;
  return (uz_wgrad3x3_plan(d, &p2) && p2.v9 == 1 && p2.bi == 64) ? 1 : 0;
}

extern "C" int uz_wgrad_xf(const uz_wgrad_desc* d, const void* L, const void* R, const float* r_scale, const float* r_shift,
                           float* out, void* workspace, void* stream, int phase) {
  UZ_REQUIRE(r_scale && r_shift, "uz_wgrad_xf: null pointer");
  UZ_REQUIRE(phase >= 0 && phase <= 2, "uz_wgrad_xf: phase %d (0: both launches, 1: main kernel, 2: slab reduction)", phase);
  if (!uz_wgrad_xf_supported(d)) {
    uz_set_error("uz_wgrad_xf: only nine-tap bf16 problems of the row-walk kernel (ask uz_wgrad_xf_supported)");
    return UZ_ENOTIMPL;
  }
  const UzXf xf = {r_scale, r_shift};
  return wgrad_phases(d, L, R, out, workspace, stream, phase, &xf);
}

extern "C" int uz_wgrad_phase(const uz_wgrad_desc* d, const void* L, const void* R, float* out, void* workspace,
                              void* stream, int phase) {
  UZ_REQUIRE(phase == 1 || phase == 2, "uz_wgrad_phase: phase %d (1: main kernel, 2: slab reduction)", phase);
  return wgrad_phases(d, L, R, out, workspace, stream, phase);
}


// ---- several independent problems in one launch pair (uz_wgrad_multi) ---------------------------------------------------
// The nn.Linear weight gradients of a backward range are not on the critical path (nothing reads them before the
// optimizer): issued one by one each was a ~12 us launch on 3 ... 30 tiles plus a ~4 us reduction, ~100 launches per
// swin_unet_v2 step.  Deferred and issued together they fill the chip with ~64-step workgroups.
constexpr int UZ_RED_MULTI = 64;
struct RedItem {
  const float* slab;
  float* out;
  long long cicj;
  int split, first;
};
struct RedMulti {
  int n, total;
  RedItem it[UZ_RED_MULTI];
};
static_assert(sizeof(RedMulti) <= 4096, "kernel arguments");
// the slab sums of wgrad_reduce_kernel<1, 4>, problem by lookup: same order of additions per element.  V = 4: four
// consecutive elements per thread (16-byte loads; Ci * Cj is a multiple of 64 for every problem of the shared launches).
template <int V>
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const RedMulti m) {
  constexpr int ZG = 4;
  __shared__ float red[ZG][64][V];
  int lo = 0, hi = m.n;   // the problem of this workgroup: binary search, first[] is increasing
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int)blockIdx.x >= m.it[mid].first) lo = mid;
    else hi = mid;
  }
  const RedItem& q = m.it[lo];
  const int x = threadIdx.x & 63, zg = threadIdx.x >> 6;
  const long long e = ((long long)((int)blockIdx.x - q.first) * 64 + x) * V;
  float acc[V];
#pragma unroll
  for (int v = 0; v < V; ++v) acc[v] = 0.f;
  auto ld = [&](int z, float* f) {
    const float* p = q.slab + (size_t)z * q.cicj + e;
    if constexpr (V == 4) {
      const float4 t = *reinterpret_cast<const float4*>(p);
      f[0] = t.x, f[1] = t.y, f[2] = t.z, f[3] = t.w;
    } else {
      f[0] = p[0];
    }
  };
  if (e < q.cicj) {
    int z = zg;
    for (; z + ZG < q.split; z += 2 * ZG) {
      float a0[V], a1[V];
      ld(z, a0);
      ld(z + ZG, a1);
#pragma unroll
      for (int v = 0; v < V; ++v) acc[v] = (acc[v] + a0[v]) + a1[v];
    }
    if (z < q.split) {
      float a0[V];
      ld(z, a0);
#pragma unroll
      for (int v = 0; v < V; ++v) acc[v] += a0[v];
    }
  }
#pragma unroll
  for (int v = 0; v < V; ++v) red[zg][x][v] = acc[v];
  __syncthreads();
  if (zg == 0 && e < q.cicj) {
    float r[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      r[v] = 0.f;
#pragma unroll
      for (int k = 0; k < ZG; ++k) r[v] += red[k][x][v];
    }
    if constexpr (V == 4) *reinterpret_cast<float4*>(q.out + e) = make_float4(r[0], r[1], r[2], r[3]);
    else q.out[e] = r[0];
  }
}

namespace {
struct MultiPlan {
  int n = 0;
  bool* together = nullptr;       // problem i goes into the shared launches
  UzWgrad2Plan* p2 = nullptr;
  long long* ws_off = nullptr;    // byte offset of problem i's slabs in the workspace
  long long ws_total = 0;
  int* order = nullptr;           // the shared problems in launch order: tile shape, then longest pixel list first
  int* launch_of = nullptr;       // order[k] goes into shared launch launch_of[k]
  int n_shared = 0;
  ~MultiPlan() {
    delete[] together;
    delete[] p2;
    delete[] ws_off;
    delete[] order;
    delete[] launch_of;
  }
};

// Finish time (in K-steps) of one shared launch under in-order dispatch of its workgroups to UZ_NUM_CU one-workgroup CUs:
// problem k of the launch contributes tiles x ceil(units / T) workgroups of ~min(units, T) steps + FIX (pipeline fill and the
// 64 KB slab store, in step units).
long long multi_makespan(const UzWgrad2Plan* p2, const int* idx, int n, long long T) {
  constexpr int FIX = 6;
  constexpr int NCU = UZ_NUM_CU_HW;   // the hardware's count, not the reserve-adjusted one: the chosen splits (summation orders) must not depend on that setting
  long long cu[NCU];
  for (int c = 0; c < NCU; ++c) cu[c] = 0;
  // a binary min-heap over the CUs' free times (all equal at the start: any array is a heap)
  auto sift = [&](int i) {
    for (;;) {
      int l = 2 * i + 1, r = l + 1, m = i;
      if (l < NCU && cu[l] < cu[m]) m = l;
      if (r < NCU && cu[r] < cu[m]) m = r;
      if (m == i) return;
      const long long t = cu[i];
      cu[i] = cu[m];
      cu[m] = t;
      i = m;
    }
  };
  long long end = 0;
  for (int k = 0; k < n; ++k) {
    const UzWgrad2Plan& p = p2[idx[k]];
    long long upb = p.units < T ? p.units : T;
    const long long split = (p.units + upb - 1) / upb;
    upb = (p.units + split - 1) / split;
    const long long wgs = (long long)p.tiles_i * p.tiles_j * ((p.units + upb - 1) / upb);
    for (long long w = 0; w < wgs; ++w) {
      cu[0] += upb + FIX;
      if (cu[0] > end) end = cu[0];
      sift(0);
    }
  }
  return end;
}
}  // namespace

static int multi_plan(const uz_wgrad_item* items, int n, MultiPlan* mp) {
  UZ_REQUIRE(items && n >= 1 && n <= 4096, "uz_wgrad_multi: items / n");
  mp->n = n;
  mp->together = new bool[n];
  mp->p2 = new UzWgrad2Plan[n];
  mp->ws_off = new long long[n];
  long long steps = 0;
  for (int i = 0; i < n; ++i) {
    const uz_wgrad_desc* d = &items[i].desc;
    const long long wb = uz_wgrad_workspace_bytes(d);   // validates the descriptor
    if (wb < 0) return (int)wb;
    UzWgrad2Plan& p = mp->p2[i];
    mp->together[i] = uz_wgrad3x3_plan(d, &p) != 0 && p.one_tap && !p.gather && !p.v9 && d->ntaps == 1;
    if (mp->together[i]) steps += (long long)p.units * p.tiles_i * p.tiles_j;
  }
  // K-steps (64 pixels each) per workgroup: long enough to carry the pipeline fill and the 64 KB slab store, short
  // enough that the whole set is several rounds of workgroups
  long long Tmax = steps / (4LL * UZ_NUM_CU);
  Tmax = Tmax < 8 ? 8 : (Tmax > 64 ? 64 : Tmax);
  // Launch order: per tile shape, the problems with the longest pixel lists first (stable), so that a launch ends on its
  // SHORT workgroups (a backward range hands the full-resolution layers over last: the last round of a launch was a few
  // 64-step workgroups on an otherwise idle chip).  Then, per launch, the T <= Tmax whose in-order dispatch finishes first:
  // swin_unet_v2 B = 16 256 x 256: 1046 + 318 workgroups = 4.09 + 1.24 rounds of 64 steps.  Results do not depend on the
  // order (every problem has its own slabs); T changes a problem's split, i.e. its fixed summation order.
  mp->order = new int[n];
  mp->launch_of = new int[n];
  int ns = 0;
  for (int big = 0; big < 2; ++big) {
    const int beg = ns;
    for (int i = 0; i < n; ++i)
      if (mp->together[i] && mp->p2[i].big == big) mp->order[ns++] = i;
    for (int a = beg + 1; a < ns; ++a) {   // insertion sort: stable, n <= 4096 and mostly a few dozen
      const int v = mp->order[a];
      int b = a;
      for (; b > beg && mp->p2[mp->order[b - 1]].units < mp->p2[v].units; --b) mp->order[b] = mp->order[b - 1];
      mp->order[b] = v;
    }
  }
  mp->n_shared = ns;
  long long* Tof = new long long[n];
  {
    const int cap = uz_wgrad3x3_multi_max();
    int launch = 0;
    for (int beg = 0; beg < ns;) {
      int end = beg;
      while (end < ns && end - beg < cap && mp->p2[mp->order[end]].big == mp->p2[mp->order[beg]].big) ++end;
      long long bestT = Tmax, best = -1;
      const bool search = !(uz_tune_flags() & 0x8);
      for (long long T = Tmax; T >= (search ? (Tmax + 1) / 2 : Tmax) && T >= 8; T -= 2) {
        const long long m = multi_makespan(mp->p2, mp->order + beg, end - beg, T);
        if (best < 0 || m < best) best = m, bestT = T;
      }
      for (int k = beg; k < end; ++k) {
        Tof[mp->order[k]] = bestT;
        mp->launch_of[k] = launch;
      }
      ++launch;
      beg = end;
    }
  }
  long long off = 0;
  for (int i = 0; i < n; ++i) {
    const uz_wgrad_desc* d = &items[i].desc;
    mp->ws_off[i] = off;
    long long bytes;
    if (mp->together[i]) {
      UzWgrad2Plan& p = mp->p2[i];
      const long long T = Tof[i];
      long long upb = p.units < T ? p.units : T;
      const long long split = (p.units + upb - 1) / upb;
      upb = (p.units + split - 1) / split;
      p.upb = (int)upb;
      p.split = (int)((p.units + upb - 1) / upb);
      p.nslabs = p.split;
      bytes = (long long)p.nslabs * d->Ci * d->Cj * (long long)sizeof(float);
    } else {
      bytes = uz_wgrad_workspace_bytes(d);
    }
    off += (bytes + 255) & ~255LL;
  }
  delete[] Tof;
  mp->ws_total = off;
  return UZ_OK;
}

extern "C" long long uz_wgrad_multi_workspace_bytes(const uz_wgrad_item* items, int n) {
  MultiPlan mp;
  const int rc = multi_plan(items, n, &mp);
  return rc != UZ_OK ? rc : mp.ws_total;
}

extern "C" int uz_wgrad_multi(const uz_wgrad_item* items, int n, void* workspace, void* stream) {
  MultiPlan mp;
  const int rc = multi_plan(items, n, &mp);
  if (rc != UZ_OK) return rc;
  UZ_REQUIRE(workspace && ((uintptr_t)workspace & 255) == 0, "uz_wgrad_multi: workspace must be 256-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  char* ws = static_cast<char*>(workspace);
  for (int i = 0; i < n; ++i) {
    UZ_REQUIRE(items[i].L && items[i].R && items[i].out, "uz_wgrad_multi: null pointer in item %d", i);
    UZ_REQUIRE((((uintptr_t)items[i].L | (uintptr_t)items[i].R) & 15) == 0, "uz_wgrad_multi: L / R must be 16-byte aligned");
    if (!mp.together[i]) {
      const int r = uz_wgrad(&items[i].desc, items[i].L, items[i].R, items[i].out, ws + mp.ws_off[i], stream);
      if (r != UZ_OK) return r;
    }
  }
  {
    UzWgradMultiItem grp[64];
    int g = 0;
    for (int k = 0; k <= mp.n_shared; ++k) {
      if (g > 0 && (k == mp.n_shared || mp.launch_of[k] != mp.launch_of[k - 1])) {
        const int r = uz_wgrad3x3_multi_launch(grp, g, s);
        if (r != UZ_OK) return r;
        g = 0;
      }
      if (k == mp.n_shared) break;
      const int i = mp.order[k];
      // a problem that is not split writes its one "slab" where the result belongs
      grp[g++] = UzWgradMultiItem{&items[i].desc, mp.p2[i], items[i].L, items[i].R,
                                  mp.p2[i].nslabs == 1 ? items[i].out : reinterpret_cast<float*>(ws + mp.ws_off[i])};
    }
  }
  bool vec4 = true;   // 16-byte stores need aligned destinations (a gradient may be a view into a flat buffer)
  for (int i = 0; i < n; ++i)
    if (mp.together[i] && mp.p2[i].nslabs > 1 && (((uintptr_t)items[i].out & 15) != 0 || ((long long)items[i].desc.Ci * items[i].desc.Cj) % 4 != 0))
      vec4 = false;
  const int epb = vec4 ? 256 : 64;   // elements per workgroup
  RedMulti rm;
  rm.n = 0;
  int blocks = 0;
  for (int i = 0; i <= n; ++i) {
    if (i < n && mp.together[i] && mp.p2[i].nslabs > 1) {
      const long long cicj = (long long)items[i].desc.Ci * items[i].desc.Cj;
      rm.it[rm.n++] = RedItem{reinterpret_cast<const float*>(ws + mp.ws_off[i]), items[i].out, cicj, mp.p2[i].nslabs, blocks};
      blocks += (int)((cicj + epb - 1) / epb);
    }
    if (rm.n == UZ_RED_MULTI || (i == n && rm.n > 0)) {
      rm.total = blocks;
      if (vec4) hipLaunchKernelGGL(wgrad_reduce_multi_kernel<4>, dim3(blocks), dim3(256), 0, s, rm);
      else hipLaunchKernelGGL(wgrad_reduce_multi_kernel<1>, dim3(blocks), dim3(256), 0, s, rm);
      UZ_LAUNCH_CHECK("uz_wgrad_multi(reduce)");
      rm.n = 0;
      blocks = 0;
    }
  }
  return UZ_OK;
}

// ---- batched one-tap products: out_b[i][j] = sum_p L_b[p][i] R_b[p][j] ------------------------------------------------
static int batched_plan(const uz_wgrad_desc* d, int batch, UzWgrad2Plan* p2) {
  UZ_REQUIRE(d && batch >= 1 && batch <= 65535, "uz_wgrad_batched: batch");
  UZ_REQUIRE(d->ntaps == 1 && d->taps_mode == UZ_TAPS_CONV, "uz_wgrad_batched: one-tap problems only");
  return uz_wgrad3x3_plan(d, p2, batch) ? 1 : 0;
}

extern "C" long long uz_wgrad_batched_workspace_bytes(const uz_wgrad_desc* d, int batch) {
  UzWgrad2Plan p2;
  const int rc = batched_plan(d, batch, &p2);
  if (rc < 0) return rc;
  if (rc == 1) return (long long)batch * p2.nslabs * d->Ci * d->Cj * (long long)sizeof(float);
  return uz_wgrad_workspace_bytes(d);   // problem by problem through uz_wgrad
}

extern "C" int uz_wgrad_batched2(const uz_wgrad_desc* d, int batch, int batch2, const void* L, long long lb, long long lb2,
                                 const void* R, long long rb, long long rb2, float* out, long long ob, void* workspace,
                                 void* stream) {
  UZ_REQUIRE(batch2 >= 1 && (long long)batch * batch2 <= 65535, "uz_wgrad_batched2: batch2");
  const int nprob = batch * batch2;
  UzWgrad2Plan p2;
  const int rc = batched_plan(d, nprob, &p2);
  if (rc < 0) return rc;
  UZ_REQUIRE(L && R && out && workspace, "uz_wgrad_batched: null pointer");
  const int es = d->dtype == UZ_BF16 ? 2 : 4;
  if (rc == 0) {   // fp32 (parity mode) and shapes outside the LDS-DMA kernel: one uz_wgrad per problem
    for (int b = 0; b < batch; ++b)
      for (int h = 0; h < batch2; ++h) {
        const int r = uz_wgrad(d, static_cast<const char*>(L) + ((long long)b * lb + (long long)h * lb2) * es,
                               static_cast<const char*>(R) + ((long long)b * rb + (long long)h * rb2) * es,
                               out + ((long long)b * batch2 + h) * ob, workspace, stream);
        if (r != UZ_OK) return r;
      }
    return UZ_OK;
  }
  UZ_REQUIRE((((uintptr_t)L | (uintptr_t)R) & 15) == 0 && lb % 8 == 0 && rb % 8 == 0 && lb2 % 8 == 0 && rb2 % 8 == 0,
             "uz_wgrad_batched: L / R must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (p2.nslabs == 1) {   // no pixel split: the kernel's slab IS the result (ntaps = 1: same layout)
    UZ_REQUIRE(ob >= (long long)d->Ci * d->Cj, "uz_wgrad_batched: results overlap (ob < Ci * Cj)");
    return uz_wgrad3x3_launch(d, p2, L, R, out, s, nprob, lb * es, rb * es, ob, batch2, lb2 * es, rb2 * es);
  }
  const int r2 = uz_wgrad3x3_launch(d, p2, L, R, static_cast<float*>(workspace), s, nprob, lb * es, rb * es, 0, batch2, lb2 * es,
                                    rb2 * es);
  if (r2 != UZ_OK) return r2;
  const long long cicj = (long long)d->Ci * d->Cj;
  const dim3 grid((unsigned)((cicj + 63) / 64), nprob);
  hipLaunchKernelGGL((wgrad_reduce_kernel<1, 4>), grid, dim3(256), 0, s, static_cast<const float*>(workspace), p2.nslabs, cicj, out, (long long)ob);
  UZ_LAUNCH_CHECK("uz_wgrad_batched(reduce)");
  return UZ_OK;
}

extern "C" int uz_wgrad_batched(const uz_wgrad_desc* d, int batch, const void* L, long long lb, const void* R, long long rb,
                                float* out, long long ob, void* workspace, void* stream) {
  return uz_wgrad_batched2(d, batch, 1, L, lb, 0, R, rb, 0, out, ob, workspace, stream);
}
