// Implicit-GEMM convolution for gfx950 (MI355X): 3x3 (any dilation) / 1x1 / ConvTranspose-k2s2
// forward and input-gradient on the MFMA matrix cores, NHWC activations.
//
// Stands in for the ATen convolution the reference reaches through nn.Conv2d /
// nn.ConvTranspose2d (reference: unet_zoo/models/common_layers.py:28,31,47,52,71,104,
// u2net.py:10) and for its input-gradient under autograd (SURVEY.md §8a rows a1,a2,a4,a7,a10,a19).
//
// GEMM view:  C[M = pixels][N = out channels] = A[M][K = taps*Cin] * B[N][K]^T
//   A row m, K-slab (tap, c0..c0+BK): 128 contiguous bytes of the tap-shifted input pixel
//   (zero when the tap falls outside the image), B row n: packed weights, K contiguous.
// Block tile BM x BN, 4 waves (64 lanes each), K-step = 128 bytes per row (64 bf16 / 32 fp32),
// register-staged global->LDS double buffer, XOR-swizzled 16-byte chunks so every
// ds_read_b128 of a 16-lane group hits 16 distinct slots of the 256-byte bank row.
// bf16: v_mfma_f32_32x32x16_bf16;  fp32: v_mfma_f32_32x32x2_f32 (exact fp32 fma chain).
// Each block walks several M tiles (persistent over M) so the per-channel BatchNorm partial
// sums stay in registers and leave the block once, as one deterministic partial row.
#include "uz_common.h"
#include <string.h>

namespace {

struct IgemmArgs {
  const void* x;
  const void* w;
  void* y;
  const float* bias;
  float* stats;
  int M, H, W, Hin, Win, Cin, ldx, Nout, ldy, K, ntaps, mode, dil, store, Co, tiles_m, Hout, Wout;
  // tap split (small-M problems: 16-128 output tiles cannot fill 256 CUs and each walks a long,
  // latency-bound K loop): blockIdx.z owns taps [z*tpg, (z+1)*tpg) and writes its fp32 partial tile to
  // part[z][M][Nout]; igemm_split_reduce_kernel sums them in fixed order (deterministic)
  float* part;
  int tpg;
};

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static __device__ __forceinline__ void run(const Vec16<bf16_t>& a, const Vec16<bf16_t>& b,
                                             f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&a),
                                                *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  // The K order inside a 128-byte slab is permuted identically for A and B (lane half h owns
  // elements 4h..4h+3 of each 32-byte pair of chunks), which leaves the dot product unchanged.
  static __device__ __forceinline__ void run(const Vec16<float>& a, const Vec16<float>& b,
                                             f32x16& c) {
#pragma unroll
    for (int t = 0; t < 4; ++t) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[t], b.v[t], c, 0, 0, 0);
  }
};

template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void igemm_kernel(const IgemmArgs a) {
  constexpr int VEC = ElemTraits<T>::VEC;
  constexpr int BK = 8 * VEC;  // elements per 128-byte row slab
  constexpr int AR = BM / 32, BR = BN / 32;
  constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
  static_assert(WM * WN == 4, "4 waves per block");
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lc = tid & 7, lr = tid >> 3;
  const int l31 = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.y * BN;
  const T* __restrict__ xg = static_cast<const T*>(a.x);
  const T* __restrict__ wg = static_cast<const T*>(a.w);
  T* __restrict__ yg = static_cast<T*>(a.y);

  const T* bptr[BR];
  bool bval[BR];
#pragma unroll
  for (int i = 0; i < BR; ++i) {
    const int n = n0 + lr + 32 * i;
    bval[i] = n < a.Nout;
    bptr[i] = wg + (size_t)(bval[i] ? n : 0) * a.K;
  }

  float s1[TN], s2[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i) s1[i] = s2[i] = 0.f;

  const int cpt = (a.Cin + BK - 1) / BK;  // K-steps per tap; the last slab may be partial (zeros)
  const int tap_lo = a.part != nullptr ? (int)blockIdx.z * a.tpg : 0;
  const int tap_hi = a.part != nullptr ? min(a.ntaps, tap_lo + a.tpg) : a.ntaps;
  const int nk = (tap_hi - tap_lo) * cpt;
  const int HW = a.H * a.W;
  const int st_sw = ((lr >> 1) & 7);       // store-side swizzle (row = lr + 32 i)
  const int ld_sw = ((l31 >> 1) & 7);      // read-side swizzle (row = 32 j + l31)

  for (int tile = blockIdx.x; tile < a.tiles_m; tile += gridDim.x) {
    const int m0 = tile * BM;
    int rh[AR], rw[AR], rpix[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int m = m0 + lr + 32 * i;
      const bool ok = m < a.M;
      const int mm = ok ? m : 0;
      const int img = mm / HW;
      const int rem = mm - img * HW;
      const int h = rem / a.W;
      const int w = rem - h * a.W;
      rh[i] = ok ? h : -(1 << 28);
      rw[i] = w;
      rpix[i] = (a.mode == UZ_TAPS_CONV) ? (img * a.Hin + h) * a.Win + w
                                          : (img * a.Hin + 2 * h) * a.Win + 2 * w;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    Vec16<T> ra[AR], rb[BR];
    int tap = tap_lo, cb = 0;  // position of the K-step being loaded

    auto load_step = [&](int kb) {
      int dy = 0, dx = 0;
      if (a.mode == UZ_TAPS_CONV) {
        if (a.ntaps == 9) {
          const int ty = tap / 3;
          dy = (ty - 1) * a.dil;
          dx = (tap - 3 * ty - 1) * a.dil;
        }
      } else {
        dy = tap >> 1;
        dx = tap & 1;
      }
      const int coff = cb * BK + lc * VEC;
      const bool cok = coff < a.Cin;
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        bool ok;
        if (a.mode == UZ_TAPS_CONV) {
          const int hh = rh[i] + dy, ww = rw[i] + dx;
          ok = cok && (unsigned)hh < (unsigned)a.Hin && (unsigned)ww < (unsigned)a.Win;
        } else {
          ok = cok && rh[i] >= 0;
        }
        const size_t off = (size_t)(rpix[i] + dy * a.Win + dx) * (size_t)a.ldx + coff;
        ra[i] = ok ? ld16(xg + off) : zero16<T>();
      }
#pragma unroll
      for (int i = 0; i < BR; ++i)
        rb[i] = (bval[i] && cok) ? ld16(bptr[i] + (size_t)tap * a.Cin + coff) : zero16<T>();
      if (++cb == cpt) {
        cb = 0;
        ++tap;
      }
    };
    auto store_step = [&](int buf) {
      char* sA = smem + buf * STAGE;
      char* sB = sA + A_BYTES;
#pragma unroll
      for (int i = 0; i < AR; ++i)
        *reinterpret_cast<Vec16<T>*>(sA + (lr + 32 * i) * 128 + ((lc ^ st_sw) << 4)) = ra[i];
#pragma unroll
      for (int i = 0; i < BR; ++i)
        *reinterpret_cast<Vec16<T>*>(sB + (lr + 32 * i) * 128 + ((lc ^ st_sw) << 4)) = rb[i];
    };

    load_step(0);
    store_step(0);
    __syncthreads();
    for (int kb = 0; kb < nk; ++kb) {
      const bool more = kb + 1 < nk;
      if (more) load_step(kb + 1);
      const char* sA = smem + (kb & 1) * STAGE;
      const char* sB = sA + A_BYTES;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int chunk = ((2 * q + lh) ^ ld_sw) << 4;
        Vec16<T> af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          af[i] = *reinterpret_cast<const Vec16<T>*>(sA + (wm * WTM + i * 32 + l31) * 128 + chunk);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bf[j] = *reinterpret_cast<const Vec16<T>*>(sB + (wn * WTN + j * 32 + l31) * 128 + chunk);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) Mma<T>::run(af[i], bf[j], acc[i][j]);
      }
      if (more) store_step((kb + 1) & 1);
      __syncthreads();
    }

    if (a.part != nullptr) {  // split: raw fp32 partial tile, finished by the reduce kernel
      float* __restrict__ pz = a.part + (size_t)blockIdx.z * a.M * a.Nout;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WTN + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m < a.M && n < a.Nout) pz[(size_t)m * a.Nout + n] = acc[i][j][r];
          }
      }
      continue;
    }
    // Epilogue: bias, store, per-channel statistics of the stored value.
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * WTN + j * 32 + l31;
      const bool nok = n < a.Nout;
      const float bv = (a.bias != nullptr && nok) ? a.bias[n] : 0.f;
      int ab = 0, co = n;
      if (a.store == UZ_STORE_SHUFFLE2X2) {
        ab = n / a.Co;
        co = n - ab * a.Co;
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m < a.M && nok) {
            const T tv = (T)(acc[i][j][r] + bv);
            size_t o;
            if (a.store == UZ_STORE_PLAIN) {
              o = (size_t)m * a.ldy + n;
            } else {
              const int img = m / HW;
              const int rem = m - img * HW;
              const int h = rem / a.W;
              const int w = rem - h * a.W;
              const size_t opix =
                  ((size_t)img * a.Hout + 2 * h + (ab >> 1)) * (size_t)a.Wout + 2 * w + (ab & 1);
              o = opix * a.ldy + co;
            }
            yg[o] = tv;
            const float fv = (float)tv;
            s1[j] += fv;
            s2[j] += fv * fv;
          }
        }
      }
    }
  }

  if (a.stats != nullptr && a.part == nullptr) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      s1[j] += __shfl_xor(s1[j], 32);
      s2[j] += __shfl_xor(s2[j], 32);
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);  // [WM][BN][2]
    if (lh == 0) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = wn * WTN + j * 32 + l31;
        red[(wm * BN + col) * 2 + 0] = s1[j];
        red[(wm * BN + col) * 2 + 1] = s2[j];
      }
    }
    __syncthreads();
    if (tid < BN) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int k = 0; k < WM; ++k) {
        t1 += red[(k * BN + tid) * 2 + 0];
        t2 += red[(k * BN + tid) * 2 + 1];
      }
      const int n = n0 + tid;
      if (n < a.Nout) {
        a.stats[((size_t)blockIdx.x * 2 + 0) * a.Nout + n] = t1;
        a.stats[((size_t)blockIdx.x * 2 + 1) * a.Nout + n] = t2;
      }
    }
  }
}

// y[m][n] = T(sum_z part[z][m][n] + bias[n]); statistics of the stored value as per-workgroup rows.
// block (bx chunk lanes, by pixel lanes); grid (gx pixel groups, gy chunk groups)
// res (nullable): a tensor of y's shape added AFTER the rounding of the sum, as a separate add of the stored result would
// (the residual form of the LDS-DMA GEMM, uz_conv_igemm_res)
template <typename T>
__global__ __launch_bounds__(256) void igemm_split_reduce_kernel(const float* __restrict__ part, int split, int M,
                                                                 int Nout, const float* __restrict__ bias,
                                                                 T* __restrict__ y, int ldy,
                                                                 float* __restrict__ stats,
                                                                 const T* __restrict__ res, int ldres) {
  constexpr int VEC = ElemTraits<T>::VEC;
  extern __shared__ __attribute__((aligned(16))) float red[];  // [by][bx][2*VEC]
  const int CC = Nout / VEC;
  const int cc = blockIdx.y * blockDim.x + threadIdx.x;
  const bool cok = cc < CC;
  const int c0 = (cok ? cc : 0) * VEC;
  float bv[VEC], s1[VEC], s2[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    bv[i] = (bias != nullptr && cok) ? bias[c0 + i] : 0.f;
    s1[i] = s2[i] = 0.f;
  }
  const size_t slab = (size_t)M * Nout;
  for (int m = blockIdx.x * blockDim.y + threadIdx.y; m < M && cok; m += gridDim.x * blockDim.y) {
    float v[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = bv[i];
    const float* src = part + (size_t)m * Nout + c0;
    for (int z = 0; z < split; ++z) {
#pragma unroll
      for (int i = 0; i < VEC; i += 4) {
        const float4 t = *reinterpret_cast<const float4*>(src + (size_t)z * slab + i);
        v[i] += t.x;
        v[i + 1] += t.y;
        v[i + 2] += t.z;
        v[i + 3] += t.w;
      }
    }
    Vec16<T> o;
#pragma unroll
    for (int i = 0; i < VEC; ++i) o.v[i] = (T)v[i];
    if (res != nullptr) {
      const Vec16<T> r = ld16(res + (size_t)m * ldres + c0);
#pragma unroll
      for (int i = 0; i < VEC; ++i) o.v[i] = (T)((float)o.v[i] + (float)r.v[i]);
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float f = (float)o.v[i];
      s1[i] += f;
      s2[i] += f * f;
    }
    st16(y + (size_t)m * ldy + c0, o);
  }
  if (stats == nullptr) return;
  float* mine = red + ((size_t)threadIdx.y * blockDim.x + threadIdx.x) * 2 * VEC;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    mine[i] = s1[i];
    mine[VEC + i] = s2[i];
  }
  __syncthreads();
  if (threadIdx.y == 0 && cok) {
    for (int r = 1; r < (int)blockDim.y; ++r) {
      const float* o = red + ((size_t)r * blockDim.x + threadIdx.x) * 2 * VEC;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        s1[i] += o[i];
        s2[i] += o[VEC + i];
      }
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      stats[((size_t)blockIdx.x * 2 + 0) * Nout + c0 + i] = s1[i];
      stats[((size_t)blockIdx.x * 2 + 1) * Nout + c0 + i] = s2[i];
    }
  }
}

struct Plan {
  int bn;       // 64 or 128
  int tiles_m, tiles_n, grid_m;
  int split;    // tap groups (1 = none); > 1 needs a workspace
  int rbx, rby, rgx, rgy;  // reduce kernel launch shape
};

// launch geometry of igemm_split_reduce_kernel for an (M, Nout) result (sized by the hardware's CU count: the number of
// statistics rows it writes, and with it the order in which uz_bn_finalize adds them, must not follow a CU reserve)
void reduce_geometry(long long M, int Nout, int vec, Plan* p) {
  const int CC = Nout / vec;
  int bx = 1;
  while (bx < CC && bx < 64) bx <<= 1;
  p->rbx = bx;
  p->rby = 256 / bx;
  p->rgy = (CC + bx - 1) / bx;
  long long gx = (M + p->rby * 4 - 1) / (p->rby * 4);
  long long capx = (long long)UZ_NUM_CU_HW * 2 / p->rgy;
  if (capx < 1) capx = 1;
  if (gx > capx) gx = capx;
  if (gx < 1) gx = 1;
  p->rgx = (int)gx;
}

int make_plan(const uz_conv_desc* d, Plan* p) {
  UZ_REQUIRE(d != nullptr, "uz_conv_igemm: null descriptor");
  UZ_REQUIRE(d->dtype == UZ_F32 || d->dtype == UZ_BF16, "uz_conv_igemm: bad dtype %d", d->dtype);
  const int vec = d->dtype == UZ_BF16 ? 8 : 4;
  const int bk = 8 * vec;
  UZ_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Nout > 0,
             "uz_conv_igemm: non-positive shape");
  {
    UzGemmPlan gp_;
    (void)gp_;
    (void)bk;
    UZ_REQUIRE(d->Cin % vec == 0, "uz_conv_igemm: Cin=%d must be a multiple of %d", d->Cin, vec);
  }
  UZ_REQUIRE(d->ldx % vec == 0 && d->ldx >= d->Cin, "uz_conv_igemm: bad ldx=%d (Cin=%d)", d->ldx,
             d->Cin);
  if (d->taps_mode == UZ_TAPS_CONV_UP2) {
    UzDirectPlan dp_;
    UZ_REQUIRE(uz_direct_plan(d, &dp_), "uz_conv_igemm: upsampled input needs the direct 3x3 kernel "
               "(ntaps=9, dil=1, even H/W, Hin=H/2, channel multiples, tensor < 2 GiB)");
  } else if (d->taps_mode == UZ_TAPS_CONV) {
    UZ_REQUIRE(d->ntaps == 1 || d->ntaps == 9, "uz_conv_igemm: ntaps=%d", d->ntaps);
    UZ_REQUIRE(d->Hin == d->H && d->Win == d->W, "uz_conv_igemm: conv taps need Hin==H, Win==W");
    UZ_REQUIRE(d->dil >= 1, "uz_conv_igemm: dil=%d", d->dil);
  } else if (d->taps_mode == UZ_TAPS_CONV_S2) {
    UzGemmPlan gs2_;
    UZ_REQUIRE(d->ntaps == 9 && d->store_mode == UZ_STORE_PLAIN && d->H == (d->Hin + 1) / 2 && d->W == (d->Win + 1) / 2,
               "uz_conv_igemm: stride-2 taps need ntaps=9, a plain store and H = ceil(Hin/2), W = ceil(Win/2)");
    UZ_REQUIRE(uz_gemm_dma_plan(d, &gs2_), "uz_conv_igemm: stride-2 taps need the LDS-DMA GEMM (channel multiples, tensor < 2 GiB)");
  } else {
    UZ_REQUIRE(d->taps_mode == UZ_TAPS_GATHER2X2 && d->ntaps == 4,
               "uz_conv_igemm: gather2x2 needs ntaps=4");
    UZ_REQUIRE((d->Hin == 2 * d->H || d->Hin == 2 * d->H + 1) && (d->Win == 2 * d->W || d->Win == 2 * d->W + 1),
               "uz_conv_igemm: gather2x2 needs Hin in {2H, 2H+1}, Win in {2W, 2W+1}");
  }
  if (d->store_mode == UZ_STORE_SHUFFLE2X2) {
    UZ_REQUIRE(d->Co > 0 && d->Nout == 4 * d->Co, "uz_conv_igemm: shuffle store needs Nout=4*Co");
    UZ_REQUIRE((d->Hout == 0 || d->Hout == 2 * d->H || d->Hout == 2 * d->H + 1) &&
                   (d->Wout == 0 || d->Wout == 2 * d->W || d->Wout == 2 * d->W + 1),
               "uz_conv_igemm: shuffle store destination must be 2H..2H+1 x 2W..2W+1");
    UZ_REQUIRE(d->ldy >= d->Co, "uz_conv_igemm: bad ldy");
  } else {
    UZ_REQUIRE(d->store_mode == UZ_STORE_PLAIN, "uz_conv_igemm: bad store_mode");
    UZ_REQUIRE(d->ldy >= d->Nout, "uz_conv_igemm: bad ldy=%d (Nout=%d)", d->ldy, d->Nout);
  }
  const long long M = (long long)d->N * d->H * d->W;
  UZ_REQUIRE(M * (long long)(d->ldx > d->ldy ? d->ldx : d->ldy) < (1LL << 40) && M < (1LL << 31),
             "uz_conv_igemm: tensor too large");
  UZ_REQUIRE((long long)d->N * d->Hin * d->Win < (1LL << 31), "uz_conv_igemm: input too large");
  p->bn = d->Nout <= 64 ? 64 : 128;
  p->tiles_m = uz_cdiv(M, 128);
  p->tiles_n = uz_cdiv(d->Nout, p->bn);
  int cap = (2 * UZ_NUM_CU) / p->tiles_n;
  if (cap < 1) cap = 1;
  p->grid_m = p->tiles_m < cap ? p->tiles_m : cap;
  // tap split for the small-M generic-kernel problems (u2net's dilated layers at <= 32x32 maps)
  p->split = 1;
  const int vec_ = d->dtype == UZ_BF16 ? 8 : 4;
  if (d->taps_mode == UZ_TAPS_CONV && d->ntaps == 9 && d->store_mode == UZ_STORE_PLAIN &&
      p->tiles_m * p->tiles_n <= UZ_NUM_CU / 2 && d->Nout % vec_ == 0 && d->ldy % vec_ == 0 &&
      !(uz_tune_flags() & 8)) {
    p->split = 9;
    reduce_geometry(M, d->Nout, vec_, p);
  }
  return UZ_OK;
}

template <typename T>
int launch(const uz_conv_desc* d, const Plan& p, const IgemmArgs& a, hipStream_t s) {
  dim3 grid(p.grid_m, p.tiles_n, a.part != nullptr ? p.split : 1), block(256);
  if (p.bn == 64) {
    hipLaunchKernelGGL((igemm_kernel<T, 128, 64, 2, 2>), grid, block, 0, s, a);
  } else {
    hipLaunchKernelGGL((igemm_kernel<T, 128, 128, 2, 2>), grid, block, 0, s, a);
  }
  UZ_LAUNCH_CHECK("uz_conv_igemm");
  if (a.part != nullptr) {
    constexpr int VEC = ElemTraits<T>::VEC;
    const size_t shm = (size_t)256 * 2 * VEC * sizeof(float);
    hipLaunchKernelGGL((igemm_split_reduce_kernel<T>), dim3(p.rgx, p.rgy), dim3(p.rbx, p.rby), shm, s,
                       static_cast<const float*>(a.part), p.split, a.M, a.Nout, a.bias, static_cast<T*>(a.y), a.ldy, a.stats,
                       static_cast<const T*>(nullptr), 0);
    UZ_LAUNCH_CHECK("uz_conv_igemm(split reduce)");
  }
  return UZ_OK;
}

// 1 when the descriptor takes the generic kernel (no direct / LDS-DMA GEMM plan)
bool generic_path(const uz_conv_desc* d) {
  UzDirectPlan dp;
  UzGemmPlan gp;
  return !uz_direct_plan(d, &dp) && !uz_gemm_dma_plan(d, &gp);
}

}  // namespace

extern "C" int uz_conv_igemm_grid_m(const uz_conv_desc* d) {
  Plan p;
  const int rc = make_plan(d, &p);
  if (rc != UZ_OK) return rc;
  UzDirectPlan dp;
  if (uz_direct_plan(d, &dp)) return dp.grid_m;
  UzGemmPlan gp;
  if (uz_gemm_dma_plan(d, &gp)) return gp.grid_m;
  return p.grid_m;
}

extern "C" int uz_conv_igemm_kernel_name(const uz_conv_desc* d, int with_workspace, char* buf, int cap) {
  Plan p;
  const int rc = make_plan(d, &p);
  if (rc != UZ_OK) return rc;
  UZ_REQUIRE(buf != nullptr && cap > 0, "uz_conv_igemm_kernel_name: no buffer");
  const char* dt = d->dtype == UZ_BF16 ? "bf16" : "f32";
  char name[96];
  UzDirectPlan dp;
  UzGemmPlan gp;
  if (uz_direct_plan(d, &dp)) {
    const char* up = d->taps_mode == UZ_TAPS_CONV_UP2 ? "_up2" : "";
    static const char* const ppn[8] = {"pp512", "pp512x64", "pp256", "pp256w16", "pp128w16", "?", "?", "?"};
    if (dp.bres == 3) snprintf(name, sizeof(name), "conv3x3_%s_%s%s%s", ppn[dp.ppcfg & 7], dt, up, (dp.ksplit > 1 && with_workspace) ? "_splitk" : "");
    else if (dp.bres == 2) snprintf(name, sizeof(name), "conv3x3_res64_%s%s", dt, up);
    else snprintf(name, sizeof(name), "conv3x3_direct_%s_bn%d%s%s", dt, dp.bn, dp.bres == 1 ? "_resident" : "", up);
  } else if (uz_gemm_dma_plan(d, &gp)) {
    snprintf(name, sizeof(name), "gemm_dma_%s", dt);
  } else {
    snprintf(name, sizeof(name), "igemm_%s_128x%d%s", dt, p.bn, (p.split > 1 && with_workspace) ? "_tapsplit" : "");
  }
  const int len = (int)strlen(name);
  snprintf(buf, (size_t)cap, "%s", name);
  return len;
}

extern "C" long long uz_conv_igemm_workspace_bytes(const uz_conv_desc* d) {
  Plan p;
  const int rc = make_plan(d, &p);
  if (rc != UZ_OK) return rc;
  {
    UzDirectPlan dp;
    if (uz_direct_plan(d, &dp) && dp.bres == 3 && dp.ksplit > 1)   // split-K ping-pong convolution
      return (long long)dp.ksplit * d->N * d->H * d->W * d->Nout * (long long)sizeof(float);
  }
  {
    UzDirectPlan dp;
    const long long gb = uz_direct_plan(d, &dp) ? 0 : uz_gemm_dma_workspace_bytes(d);   // split-K LDS-DMA GEMM
    if (gb > 0) return gb;
  }
  if (p.split <= 1 || !generic_path(d)) return 0;
  return (long long)p.split * d->N * d->H * d->W * d->Nout * (long long)sizeof(float);
}

extern "C" int uz_conv_igemm_ws_grid_m(const uz_conv_desc* d) {
  Plan p;
  const int rc = make_plan(d, &p);
  if (rc != UZ_OK) return rc;
  {
    UzDirectPlan dp;
    if (uz_direct_plan(d, &dp) && dp.bres == 3 && dp.ksplit > 1) {   // the statistics rows come from the reduce pass
      Plan rp;
      reduce_geometry((long long)d->N * d->H * d->W, d->Nout, 8, &rp);
      return rp.rgx;
    }
  }
  {
    UzDirectPlan dp;
    if (!uz_direct_plan(d, &dp) && uz_gemm_dma_workspace_bytes(d) > 0) {   // split-K LDS-DMA GEMM: rows of its reduce pass
      Plan rp;
      reduce_geometry((long long)d->N * d->H * d->W, d->Nout, 8, &rp);
      return rp.rgx;
    }
  }
  if (p.split > 1 && generic_path(d)) return p.rgx;
  return uz_conv_igemm_grid_m(d);
}

extern "C" int uz_conv_igemm(const uz_conv_desc* d, const void* x, const void* w_packed,
                             const float* bias, void* y, float* stats_partial, void* stream) {
  return uz_conv_igemm_ws(d, x, w_packed, bias, y, stats_partial, nullptr, stream);
}

// split-K LDS-DMA GEMM: fp32 partial tiles of the K ranges, then the fixed-order reduce + bias (+ residual) pass
static int gemm_split_k(const uz_conv_desc* d, const UzGemmPlan& gp, const void* x, const void* w_packed, const float* bias,
                        const void* res, int ldres, void* y, void* workspace, hipStream_t s, float* stats = nullptr) {
  const int r1 = uz_gemm_dma_launch(d, gp, x, w_packed, nullptr, y, nullptr, s, nullptr, 0, nullptr, static_cast<float*>(workspace));
  if (r1 != UZ_OK) return r1;
  Plan rp;
  const long long M = (long long)d->N * d->H * d->W;
  reduce_geometry(M, d->Nout, 8, &rp);
  const size_t shm = (size_t)256 * 2 * 8 * sizeof(float);
  hipLaunchKernelGGL((igemm_split_reduce_kernel<bf16_t>), dim3(rp.rgx, rp.rgy), dim3(rp.rbx, rp.rby), shm, s,
                     static_cast<const float*>(workspace), gp.ksplit, (int)M, d->Nout, bias, static_cast<bf16_t*>(y), d->ldy,
                     stats, static_cast<const bf16_t*>(res), ldres);
  UZ_LAUNCH_CHECK("uz_conv_igemm(split-K GEMM reduce)");
  return UZ_OK;
}

extern "C" int uz_conv_igemm_res(const uz_conv_desc* d, const void* x, const void* w_packed, const float* bias,
                                 const void* res, int ldres, void* y, void* stream) {
  return uz_conv_igemm_res_ws(d, x, w_packed, bias, res, ldres, y, nullptr, stream);
}

extern "C" int uz_conv_igemm_res_ws(const uz_conv_desc* d, const void* x, const void* w_packed, const float* bias,
                                    const void* res, int ldres, void* y, void* workspace, void* stream) {
  Plan p;
  const int rc = make_plan(d, &p);
  if (rc != UZ_OK) return rc;
  UZ_REQUIRE(x && w_packed && y && res, "uz_conv_igemm_res: null pointer");
  UZ_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w_packed & 15) == 0 && ((uintptr_t)y & 15) == 0 &&
                 ((uintptr_t)res & 15) == 0, "uz_conv_igemm_res: x / w / y / res must be 16-byte aligned");
  const int vec = d->dtype == UZ_BF16 ? 8 : 4;
  UZ_REQUIRE(ldres >= d->Nout && ldres % vec == 0, "uz_conv_igemm_res: bad ldres %d", ldres);
  UzDirectPlan dp;
  UzGemmPlan gp;
  if (d->store_mode != UZ_STORE_PLAIN || uz_direct_plan(d, &dp) || !uz_gemm_dma_plan(d, &gp)) {
    uz_set_error("uz_conv_igemm_res: only problems of the LDS-DMA GEMM with a plain store take a residual");
    return UZ_ENOTIMPL;
  }
  if (gp.ksplit > 1 && workspace != nullptr)
    return gemm_split_k(d, gp, x, w_packed, bias, res, ldres, y, workspace, static_cast<hipStream_t>(stream));
  return uz_gemm_dma_launch(d, gp, x, w_packed, bias, y, nullptr, static_cast<hipStream_t>(stream), res, ldres);
}

// ---- convolution reading its input through the BatchNorm + ReLU in front of it (include/unetzoo_hip.h) -----------------
extern "C" int uz_conv_igemm_xf_supported(const uz_conv_desc* d) {
  UzDirectPlan dp;
  if (d == nullptr || d->dtype != UZ_BF16) return 0;
  if (!uz_direct_plan(d, &dp) || dp.bres != 3 || dp.ksplit > 1) return 0;
  UzPpPlan pp = {dp.ppcfg, dp.bn, dp.th_n, dp.tw_n, dp.ntiles, dp.tiles_n, dp.grid_m, dp.ksplit, dp.cps};
  return d->Cin <= uz_pp_xf_channels(pp) ? 1 : 0;
}

extern "C" int uz_conv_igemm_xf(const uz_conv_desc* d, const void* x, const float* in_scale, const float* in_shift,
                                const void* w_packed, const float* bias, void* y, float* stats_partial, void* stream) {
  Plan p;
  const int rc = make_plan(d, &p);
  if (rc != UZ_OK) return rc;
  UZ_REQUIRE(x && w_packed && y && in_scale && in_shift, "uz_conv_igemm_xf: null pointer");
  UZ_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w_packed & 15) == 0 && ((uintptr_t)y & 15) == 0,
             "uz_conv_igemm_xf: x / w / y must be 16-byte aligned");
  if (!uz_conv_igemm_xf_supported(d)) {
    uz_set_error("uz_conv_igemm_xf: only bf16 3x3 problems of the ping-pong configurations whose LDS image leaves room for "
                 "the channel table (ask uz_conv_igemm_xf_supported)");
    return UZ_ENOTIMPL;
  }
  UzDirectPlan dp;
  UZ_REQUIRE(uz_direct_plan(d, &dp), "uz_conv_igemm_xf: no plan");
  const UzXf xf = {in_scale, in_shift};
  return uz_direct_launch(d, dp, x, w_packed, bias, y, stats_partial, static_cast<hipStream_t>(stream), nullptr, nullptr, &xf);
}

extern "C" int uz_conv_igemm_bnred_supported(const uz_conv_desc* d) {
  UzDirectPlan dp;
  UzGemmPlan gp;
  if (d == nullptr || d->dtype != UZ_BF16) return 0;
  if (uz_direct_plan(d, &dp)) return dp.bres != 2 ? 1 : 0;
  return (d->store_mode == UZ_STORE_PLAIN && uz_gemm_dma_plan(d, &gp)) ? 1 : 0;
}

extern "C" int uz_conv_igemm_bnred(const uz_conv_desc* d, const void* x, const void* w_packed, void* y,
                                   const void* bn_y, int ld_bny, const float* scale, const float* shift,
                                   const float* mean, const float* invstd, float* partial, void* stream) {
  Plan p;
  const int rc = make_plan(d, &p);
  if (rc != UZ_OK) return rc;
  UZ_REQUIRE(x && w_packed && y && bn_y && scale && shift && mean && invstd && partial,
             "uz_conv_igemm_bnred: null pointer");
  UZ_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w_packed & 15) == 0 && ((uintptr_t)y & 15) == 0 &&
                 ((uintptr_t)bn_y & 15) == 0, "uz_conv_igemm_bnred: x / w / y / bn_y must be 16-byte aligned");
  UZ_REQUIRE(ld_bny >= d->Nout && ld_bny % 8 == 0, "uz_conv_igemm_bnred: bad ld_bny %d", ld_bny);
  UzDirectPlan dp;
  UzGemmPlan gp;
  if (!uz_conv_igemm_bnred_supported(d)) {
    uz_set_error("uz_conv_igemm_bnred: only bf16 problems of the direct 3x3 kernels with the LDS-staged epilogue and of "
                 "the LDS-DMA GEMM with a plain store (ask uz_conv_igemm_bnred_supported)");
    return UZ_ENOTIMPL;
  }
  const UzBnRed br = {bn_y, ld_bny, scale, shift, mean, invstd};
  if (uz_direct_plan(d, &dp))
    return uz_direct_launch(d, dp, x, w_packed, nullptr, y, partial, static_cast<hipStream_t>(stream), &br);
  UZ_REQUIRE(uz_gemm_dma_plan(d, &gp), "uz_conv_igemm_bnred: no plan");
  return uz_gemm_dma_launch(d, gp, x, w_packed, nullptr, y, partial, static_cast<hipStream_t>(stream), nullptr, 0, &br);
}

extern "C" int uz_conv_igemm_ws(const uz_conv_desc* d, const void* x, const void* w_packed,
                                const float* bias, void* y, float* stats_partial, void* workspace,
                                void* stream) {
  Plan p;
  const int rc = make_plan(d, &p);
  if (rc != UZ_OK) return rc;
  UZ_REQUIRE(x && w_packed && y, "uz_conv_igemm: null pointer");
  UZ_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w_packed & 15) == 0 && ((uintptr_t)y & 15) == 0,
             "uz_conv_igemm: x / w / y must be 16-byte aligned");
  UzDirectPlan dp;
  if (uz_direct_plan(d, &dp)) {
    if (dp.bres == 3 && dp.ksplit > 1 && workspace != nullptr) {
      // split-K ping-pong convolution: fp32 partial tiles, then the fixed-order reduce + bias + statistics pass
      hipStream_t s = static_cast<hipStream_t>(stream);
      const int r1 = uz_direct_launch(d, dp, x, w_packed, nullptr, y, nullptr, s, nullptr, static_cast<float*>(workspace));
      if (r1 != UZ_OK) return r1;
      Plan rp;
      const long long M = (long long)d->N * d->H * d->W;
      reduce_geometry(M, d->Nout, 8, &rp);
      const size_t shm = (size_t)256 * 2 * 8 * sizeof(float);
      hipLaunchKernelGGL((igemm_split_reduce_kernel<bf16_t>), dim3(rp.rgx, rp.rgy), dim3(rp.rbx, rp.rby), shm, s,
                         static_cast<const float*>(workspace), dp.ksplit, (int)M, d->Nout, bias, static_cast<bf16_t*>(y), d->ldy,
                         stats_partial, static_cast<const bf16_t*>(nullptr), 0);
      UZ_LAUNCH_CHECK("uz_conv_igemm(split-K reduce)");
      return UZ_OK;
    }
    return uz_direct_launch(d, dp, x, w_packed, bias, y, stats_partial, static_cast<hipStream_t>(stream));
  }
  UzGemmPlan gp;
  if (uz_gemm_dma_plan(d, &gp)) {
    if (gp.ksplit > 1 && workspace != nullptr)   // (statistics rows: uz_conv_igemm_ws_grid_m, written by the reduce pass)
      return gemm_split_k(d, gp, x, w_packed, bias, nullptr, 0, y, workspace, static_cast<hipStream_t>(stream), stats_partial);
    return uz_gemm_dma_launch(d, gp, x, w_packed, bias, y, stats_partial, static_cast<hipStream_t>(stream));
  }
  IgemmArgs a;
  a.x = x;
  a.w = w_packed;
  a.y = y;
  a.bias = bias;
  a.stats = stats_partial;
  a.M = d->N * d->H * d->W;
  a.H = d->H;
  a.W = d->W;
  a.Hin = d->Hin;
  a.Win = d->Win;
  a.Cin = d->Cin;
  a.ldx = d->ldx;
  a.Nout = d->Nout;
  a.ldy = d->ldy;
  a.K = d->ntaps * d->Cin;
  a.ntaps = d->ntaps;
  a.mode = d->taps_mode;
  a.dil = d->dil;
  a.store = d->store_mode;
  a.Co = d->Co;
  a.Hout = d->Hout ? d->Hout : 2 * d->H;
  a.Wout = d->Wout ? d->Wout : 2 * d->W;
  a.tiles_m = p.tiles_m;
  a.part = (p.split > 1 && workspace != nullptr) ? static_cast<float*>(workspace) : nullptr;
  a.tpg = (d->ntaps + p.split - 1) / p.split;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return d->dtype == UZ_BF16 ? launch<bf16_t>(d, p, a, s) : launch<float>(d, p, a, s);
}
