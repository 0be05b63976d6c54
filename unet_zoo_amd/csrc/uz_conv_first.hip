// The network's FIRST convolution as direct kernels, bf16, gfx950 (MI355X): Conv2d(C <= 3, Cout in {32, 64}, k3, p1) on the
// fp32 NCHW input image, and its weight gradient (reference: the first layer of every UNet-family model, e.g.
// unet_zoo/models/common_layers.py:28 reached from unet.py:15; loss.backward(), training_loop.py:119).
//
// Rounds 1-3 ran this layer as im2col (fp32 NCHW -> [P][32] bf16 patches, 67 MB written at B = 16 256 x 256) + a K = 32 GEMM
// + a one-tap weight gradient over the patches: 35 + 54 us forward, 67 us backward, of which the patch buffer's write and
// two reads are pure overhead.  Here the 27 (c, ty, tx) inputs of a pixel are gathered from an LDS halo tile of the image
// straight into MFMA fragments:
//   forward   D[co][pixel] = sum_k W[co][k] patch[pixel][k]      k = c * 9 + ty * 3 + tx  (OIHW order), K padded to 32:
//             the weights are the row operand (16 registers, loaded once), a lane builds its pixel's 8-value K blocks from
//             8 ds_read_b32 + 4 v_cvt_pk; output rounded once to bf16, staged through a wave-private LDS strip into full
//             128-byte NHWC rows; BatchNorm sums of the STORED values per workgroup (partial rows as the other convolution
//             epilogues write them).
//   weight gradient  dW[co][k] = sum_pixel dy[pixel][co] patch[pixel][k]: dy tiles pixel-major in LDS, read transposed
//             (ds_read_b64_tr_b16); a patch fragment is 8 CONSECUTIVE pixels of one (c, ty, tx) = 8 consecutive floats of an
//             LDS halo row.  Persistent workgroups keep their 64 x 32 accumulators over their tiles; partial slabs
//             [workgroup][Cout][32] + uz_sum-style fixed-order reduction (bitwise reproducible).
// x and the weights are rounded to bf16 exactly as the im2col path stored them, so both paths compute the same products.
#include "uz_common.h"

namespace {

constexpr int TH = 8, TW = 32, PH = TH + 2, PW = TW + 2;   // workgroup tile: 8 rows x 32 columns of one image
constexpr int STG = 144;                                    // staging row: 64 channels bf16 + 16 bytes (bank spread)

struct CfArgs {
  const float* x;
  const float* w;
  const float* bias;
  void* y;
  float* stats;
  const void* dy;
  float* slab;
  int N, C, H, W, Cout, ldy, th_n, tw_n, ntiles;
};

__device__ __forceinline__ float round_bf16(float v) { return (float)(bf16_t)v; }

// halo tile of image `img` at (h0, w0): sx[c][PH][PW] fp32 (bf16-rounded values), zero outside the image.  Register-staged
// in two halves so that a persistent workgroup fetches its NEXT tile (four independent loads per thread, one round trip)
// while it computes the current one.
// The loads are UNCONDITIONAL (element 0 of the tensor for a halo position outside the image) and their RAW values are kept;
// bit j of the returned mask says whether h[j] is real, and halo_store() -- one tile later -- zeroes the rest.  As loads
// under `if (inside)` each was followed by s_waitcnt vmcnt(0) (hipcc 7.2): four round trips, waited for on the spot, where
// the design wants one that lands during the current tile.
__device__ __forceinline__ unsigned halo_fetch(const CfArgs& a, float (&h)[4], int tile, int tid) {
  const int per = a.th_n * a.tw_n;
  const int img = tile / per, rem = tile - img * per;
  const int h0 = (rem / a.tw_n) * TH, w0 = (rem % a.tw_n) * TW;
  const int n = a.C * PH * PW;
  unsigned mask = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = tid + 256 * j;
    const int c = idx / (PH * PW), r2 = idx - c * (PH * PW);
    const int r = r2 / PW, col = r2 - r * PW;
    const int gh = h0 - 1 + r, gw = w0 - 1 + col;
    const bool ok = idx < n && (unsigned)gh < (unsigned)a.H && (unsigned)gw < (unsigned)a.W;
    h[j] = a.x[ok ? (((size_t)img * a.C + c) * a.H + gh) * a.W + gw : 0];
    mask |= ok ? 1u << j : 0u;
  }
  return mask;
}
__device__ __forceinline__ void halo_store(float* sx, const float (&h)[4], unsigned mask, int tid, int n) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (tid + 256 * j < n) sx[tid + 256 * j] = (mask >> j & 1u) ? round_bf16(h[j]) : 0.f;
}
static_assert(3 * PH * PW <= 4 * 256, "four halo elements per thread");

// ---- forward: one workgroup (4 waves) per 8 x 32 tile; wave w owns rows 2w, 2w + 1; CT = Cout / 32 ------------------------
template <int CT>
__global__ __launch_bounds__(256) void conv_first_fwd_kernel(const CfArgs a) {
  // (the bias is added in fp32 from an LDS table when the accumulators are rounded: carried in two spare K slots as bf16 value +
  // bf16 remainder it was off by 2^-17 relative, enough to round 2e-4 of the outputs the other way)
  __shared__ float sx[3 * PH * PW];
  __shared__ __attribute__((aligned(16))) float sbias[64];
  __shared__ __attribute__((aligned(16))) char sstg[4][32 * STG];
  __shared__ float sred[4][2][CT * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 31, b = lane >> 5;
  const int per = a.th_n * a.tw_n;
  const int K = a.C * 9;
  float hreg[4];
  unsigned hmask = 0;
  if ((int)blockIdx.x < a.ntiles) hmask = halo_fetch(a, hreg, blockIdx.x, tid);
  if (tid < 64) sbias[tid] = (a.bias != nullptr && tid < a.Cout) ? a.bias[tid] : 0.f;

  // weights as the row operand: lane (co = 32 t + lane % 32, K block b) holds w[co][16 kh + 8 b + i], zero beyond K
  bf16x8 wfr[CT][2];
  float wraw[CT][2][8];
#pragma unroll
  for (int t = 0; t < CT; ++t)
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = 16 * kh + 8 * b + i, co = 32 * t + px;
        wraw[t][kh][i] = a.w[(k < K && co < a.Cout) ? (size_t)co * K + k : 0];   // unconditional loads, all in flight ...
      }
#pragma unroll
  for (int t = 0; t < CT; ++t)
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(wraw[t][kh][i]));   // ... opaque, so that the selects below cannot pull them back under a branch
#pragma unroll
  for (int t = 0; t < CT; ++t)
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = 16 * kh + 8 * b + i, co = 32 * t + px;
        wfr[t][kh][i] = (bf16_t)((k < K && co < a.Cout) ? wraw[t][kh][i] : 0.f);
      }
  // this lane's 16 patch offsets (floats): k -> (c, ty, tx) -> (c PH + ty) PW + tx, plus its pixel column
  int koff[2][8];
#pragma unroll
  for (int kh = 0; kh < 2; ++kh)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = 16 * kh + 8 * b + i;
      const int c = k / 9, t9 = k - c * 9, ty = t9 / 3, tx = t9 - ty * 3;
      koff[kh][i] = (k < K ? (c * PH + ty) * PW + tx : 0) + px;
    }
  // BatchNorm sums of the stored values: taken where the staged rows are read back, a lane always reads the same 8 channels
  constexpr int CPP = CT * 4;                  // 16-byte chunks per pixel
  constexpr int PPI = 64 / CPP;                // pixels per store instruction
  const int ch = lane % CPP;
  float st1[8], st2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) st1[e] = st2[e] = 0.f;

  char* stg = sstg[wave];
  bf16_t* yg = static_cast<bf16_t*>(a.y);
#pragma unroll 1
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int img = tile / per, rem = tile - img * per;
    const int h0 = (rem / a.tw_n) * TH, w0 = (rem % a.tw_n) * TW;
    __syncthreads();   // the previous tile's halo is consumed
    halo_store(sx, hreg, hmask, tid, a.C * PH * PW);
    __syncthreads();
    {   // the next tile, in flight during this tile's rows (unconditionally: the last tile fetches itself again -- a load under
        // a branch is waited for where the branch ends)
      const int nxt = tile + (int)gridDim.x;
      hmask = halo_fetch(a, hreg, nxt < a.ntiles ? nxt : tile, tid);
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int row = 2 * wave + rr, gh = h0 + row;
      bf16x8 pfr[2];
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int i = 0; i < 8; ++i) pfr[kh][i] = (bf16_t)sx[koff[kh][i] + row * PW];
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfr[t][0], pfr[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfr[t][1], pfr[1], acc, 0, 0, 0);
        // + bias, round once; 4 consecutive channels = 8 bytes into the staging strip
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4 bq = *reinterpret_cast<const f32x4*>(sbias + 32 * t + 8 * g4 + 4 * b);
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(acc[4 * g4 + e] + bq[e]);
          *reinterpret_cast<bf16x4*>(stg + px * STG + (32 * t + 8 * g4 + 4 * b) * 2) = o;
        }
      }
      // full NHWC rows: CPP lanes x 16 bytes = the CT * 64 bytes of one pixel (CT = 2: a 128-byte line).  The strip is this
      // wave's own: lanes hand data to other lanes of the SAME wave, whose LDS operations complete in program order -- no
      // s_barrier needed; the wave barriers keep the compiler from moving accesses across the hand-over.
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < 32 / PPI; ++j) {
        const int p = j * PPI + lane / CPP;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(stg + p * STG + ch * 16);
        const int gw = w0 + p;
        if (gh < a.H && gw < a.W && ch * 8 < a.Cout) {
          *reinterpret_cast<bf16x8*>(yg + (((size_t)img * a.H + gh) * a.W + gw) * a.ldy + ch * 8) = v;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float f = (float)v[e];
            st1[e] += f;
            st2[e] += f * f;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();   // the strip is rewritten by the next row
    }
  }
  if (a.stats == nullptr) return;
  // ---- this workgroup's partial row: the PPI lanes of a wave that read chunk ch, then the four waves, in fixed order
  __syncthreads();
  float* red = reinterpret_cast<float*>(sstg[wave]);   // [64 lanes][16 sums] = 4 KB of the wave's 4.5 KB strip
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[lane * 16 + e] = st1[e];
    red[lane * 16 + 8 + e] = st2[e];
  }
  __builtin_amdgcn_wave_barrier();   // same-wave hand-over, as above
#pragma unroll
  for (int idx = lane; idx < CPP * 16; idx += 64) {   // (chunk c2, value v2 of its 16 sums)
    const int c2 = idx / 16, v2 = idx % 16;
    float sacc = 0.f;
#pragma unroll
    for (int q = 0; q < PPI; ++q) sacc += red[(q * CPP + c2) * 16 + v2];
    sred[wave][v2 >> 3][c2 * 8 + (v2 & 7)] = sacc;
  }
  __syncthreads();
  if (tid < 2 * CT * 32) {
    const int q = tid / (CT * 32), co = tid % (CT * 32);
    const float sum = (sred[0][q][co] + sred[1][q][co]) + (sred[2][q][co] + sred[3][q][co]);
    if (co < a.Cout) a.stats[((size_t)blockIdx.x * 2 + q) * a.Cout + co] = sum;
  }
}

// ---- weight gradient: persistent workgroups over tiles; wave w owns rows 2w, 2w + 1 of a tile (4 sub-steps of 16 pixels) ----
template <int CT>
__global__ __launch_bounds__(256) void conv_first_wgrad_kernel(const CfArgs a) {
  constexpr int RB = CT * 64;                       // bytes per dy pixel row in LDS
  __shared__ float sx[3 * PH * PW];
  __shared__ __attribute__((aligned(16))) char sdy[TH * TW * RB];
  static_assert(4 * CT * 16 * 64 * 4 == TH * TW * RB, "the waves' accumulators meet in the dy tile's LDS");
  float (*sacc)[CT][16][64] = reinterpret_cast<float (*)[CT][16][64]>(sdy);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, b = lane >> 5;
  const int per = a.th_n * a.tw_n;
  const int K = a.C * 9;
  const bf16_t* dyg = static_cast<const bf16_t*>(a.dy);

  // patch fragment: lane (k = lane % 32 -> (c, ty, tx), pixel block b): 8 consecutive pixels of one halo row
  const int k = l31;
  const int kc = k / 9, kt = k - kc * 9, kty = kt / 3, ktx = kt - kty * 3;
  const int pbase = (k < K) ? (kc * PH + kty) * PW + ktx + 8 * b : 0;
  // dy fragment (transposed read): 16-lane group g, pixel row q4, channel quad p4; 64-byte granules swizzled with (pixel >> 1) & 1
  const int g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int lk = 8 * (g >> 1) + q4, lcol = 16 * (g & 1) + 4 * p4;
  typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;

  f32x16 acc[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  constexpr int CPP = RB / 16;                      // 16-byte chunks per dy pixel
  constexpr int NDY = TH * TW * CPP / 256;          // dy chunks per thread and tile (8 or 4)
  float hreg[4];
  f32x4 dreg[NDY];
  unsigned hmask = 0, dmask = 0;
  auto dy_fetch = [&](int tile) __attribute__((always_inline)) {
    const int img = tile / per, rem = tile - img * per;
    const int h0 = (rem / a.tw_n) * TH, w0 = (rem % a.tw_n) * TW;
#pragma unroll
    for (int j = 0; j < NDY; ++j) {
      const int idx = tid + 256 * j;
      const int p = idx / CPP, ch = idx % CPP;
      const int r = p / TW, c = p % TW, gh = h0 + r, gw = w0 + c;
      // unconditional (pixel 0 for a chunk outside the image / beyond the channels), RAW; zeroed where it is stored to LDS
      const bool ok = gh < a.H && gw < a.W && ch * 8 < a.Cout;
      dreg[j] = *reinterpret_cast<const f32x4*>(ok ? dyg + (((size_t)img * a.H + gh) * a.W + gw) * a.ldy + ch * 8 : dyg);
      dmask |= ok ? 1u << j : 0u;
    }
  };
  if ((int)blockIdx.x < a.ntiles) {
    hmask = halo_fetch(a, hreg, blockIdx.x, tid);
    dy_fetch(blockIdx.x);
  }
#pragma unroll 1
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    __syncthreads();   // the previous tile is consumed
    halo_store(sx, hreg, hmask, tid, a.C * PH * PW);
    // dy tile, pixel-major [row][col][channel]; 64-byte granule swizzle on the way in
#pragma unroll
    for (int j = 0; j < NDY; ++j) {
      const int idx = tid + 256 * j;
      const int p = idx / CPP, ch = idx % CPP;
      const int gran = ch >> 2, sw = (p >> 1) & 1;
      const int pg = (CT == 2) ? (gran ^ sw) : gran;   // (one granule per pixel at CT = 1: nothing to swizzle)
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(sdy + p * RB + pg * 64 + (ch & 3) * 16) = (dmask >> j & 1u) ? dreg[j] : z4;
    }
    __syncthreads();
    {   // the next tile streams in while this one is multiplied (unconditionally: the last tile fetches itself again)
      const int nxt = tile + (int)gridDim.x, ft = nxt < a.ntiles ? nxt : tile;
      hmask = halo_fetch(a, hreg, ft, tid);
      dmask = 0;
      dy_fetch(ft);
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int row = 2 * wave + rr, c0 = 16 * half;
        bf16x8 pfr;
#pragma unroll
        for (int i = 0; i < 8; ++i) pfr[i] = (bf16_t)sx[pbase + row * PW + c0 + i];
#pragma unroll
        for (int t = 0; t < CT; ++t) {
          const int p = row * TW + c0 + lk;   // this lane's pixel row of the transposed read (and + 4)
          const int sw = (p >> 1) & 1;
          const int gr = (CT == 2) ? (t ^ sw) : 0;
          const char* ad = sdy + p * RB + gr * 64 + lcol * 2;
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(ad));
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(ad + 4 * RB));
          const bf16x8 dfr = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dfr, pfr, acc[t], 0, 0, 0);
        }
      }
  }
  // ---- the four waves' accumulators, fixed order, then this workgroup's slab [Cout][32] ----------------------------------
  __syncthreads();
#pragma unroll
  for (int t = 0; t < CT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[wave][t][r][lane] = acc[t][r];
  __syncthreads();
  for (int idx = tid; idx < CT * 16 * 64; idx += 256) {
    const int t = idx / (16 * 64), r = (idx / 64) % 16, l = idx % 64;
    const float s = (sacc[0][t][r][l] + sacc[1][t][r][l]) + (sacc[2][t][r][l] + sacc[3][t][r][l]);
    const int co = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), kk = l & 31;
    if (co < a.Cout) a.slab[((size_t)blockIdx.x * a.Cout + co) * 32 + kk] = s;
  }
}

// dw[co][k < K] = sum over workgroups of slab[wg][co][k] in a fixed order: thread (o, zg) of a block sums the slabs zg, zg + 16,
// ... of output 16 blockIdx.x + o with eight independent loads in flight (one thread walking all 768 slabs was a chain of
// 384 memory round trips: 150 us), then the 16 partial sums meet in LDS in ascending zg
__global__ __launch_bounds__(256) void conv_first_wgrad_reduce_kernel(const float* __restrict__ slab, int nwg, int Cout, int K,
                                                                     float* __restrict__ dw) {
  __shared__ float part[16][17];
  const int o = threadIdx.x & 15, zg = threadIdx.x >> 4;
  const int idx = blockIdx.x * 16 + o;          // output (co, k) = (idx / 32, idx % 32); Cout * 32 is a multiple of 16
  const size_t stride = (size_t)Cout * 32;
  float s = 0.f;
  int z = zg;
  for (; z + 7 * 16 < nwg; z += 8 * 16) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = slab[(size_t)(z + 16 * u) * stride + idx];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; z < nwg; z += 16) s += slab[(size_t)z * stride + idx];
  part[zg][o] = s;
  __syncthreads();
  if (zg == 0) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += part[q][o];
    const int co = idx >> 5, k = idx & 31;
    if (k < K) dw[(size_t)co * K + k] = t;
  }
}

int check(int dtype, int N, int C, int H, int W, int Cout, int ld, const char* what) {
  UZ_REQUIRE(dtype == UZ_BF16, "%s: bf16 only (the fp32 run mode takes the im2col path)", what);
  UZ_REQUIRE(N > 0 && H > 0 && W > 0 && C >= 1 && C <= 3, "%s: C = %d not in 1 .. 3", what, C);
  UZ_REQUIRE((Cout == 32 || Cout == 64) && ld >= Cout && ld % 8 == 0, "%s: Cout = %d (32 or 64), ld = %d", what, Cout, ld);
  UZ_REQUIRE((long long)N * H * W * ld * 2 < (1LL << 40), "%s: too large", what);
  return UZ_OK;
}

void fill(CfArgs& a, int N, int C, int H, int W, int Cout, int ld) {
  a.N = N;
  a.C = C;
  a.H = H;
  a.W = W;
  a.Cout = Cout;
  a.ldy = ld;
  a.th_n = (H + TH - 1) / TH;
  a.tw_n = (W + TW - 1) / TW;
  a.ntiles = N * a.th_n * a.tw_n;
}

}  // namespace

extern "C" int uz_conv3x3_first_supported(int dtype, int C, int Cout) {
  return dtype == UZ_BF16 && C >= 1 && C <= 3 && (Cout == 32 || Cout == 64);
}

static int first_fwd_grid(int ntiles) {
  const int cap = 3 * UZ_NUM_CU_HW;   // 160 VGPRs, 24 KB of LDS: three workgroups per CU; each walks its tiles (sized by the
  return ntiles < cap ? ntiles : cap;   // hardware's CU count: the number of partial rows must not follow a CU reserve)
}

extern "C" int uz_conv3x3_first_rows(int N, int H, int W) {
  return first_fwd_grid(N * ((H + TH - 1) / TH) * ((W + TW - 1) / TW));
}

extern "C" int uz_conv3x3_first_fwd(int dtype, const float* x, int N, int C, int H, int W, const float* w, const float* bias,
                                    int Cout, void* y, int ldy, float* stats, void* stream) {
  const int rc = check(dtype, N, C, H, W, Cout, ldy, "uz_conv3x3_first_fwd");
  if (rc != UZ_OK) return rc;
  UZ_REQUIRE(x && w && y && ((uintptr_t)y & 15) == 0, "uz_conv3x3_first_fwd: null / unaligned pointer");
  CfArgs a = {};
  a.x = x;
  a.w = w;
  a.bias = bias;
  a.y = y;
  a.stats = stats;
  fill(a, N, C, H, W, Cout, ldy);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int grid = first_fwd_grid(a.ntiles);
  if (Cout == 64) hipLaunchKernelGGL(conv_first_fwd_kernel<2>, dim3(grid), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(conv_first_fwd_kernel<1>, dim3(grid), dim3(256), 0, s, a);
  UZ_LAUNCH_CHECK("uz_conv3x3_first_fwd");
  return UZ_OK;
}

static int first_wgrad_grid(int ntiles) {
  const int cap = 3 * UZ_NUM_CU_HW;   // 132 VGPRs: three workgroups per CU, all resident: one round
  return ntiles < cap ? ntiles : cap;
}

extern "C" long long uz_conv3x3_first_wgrad_workspace_bytes(int N, int H, int W, int Cout) {
  const int ntiles = N * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
  return (long long)first_wgrad_grid(ntiles) * Cout * 32 * (long long)sizeof(float);
}

extern "C" int uz_conv3x3_first_wgrad(int dtype, const float* x, int N, int C, int H, int W, const void* dy, int lddy,
                                      int Cout, float* dw, void* workspace, void* stream) {
  const int rc = check(dtype, N, C, H, W, Cout, lddy, "uz_conv3x3_first_wgrad");
  if (rc != UZ_OK) return rc;
  UZ_REQUIRE(x && dy && dw && workspace && ((uintptr_t)dy & 15) == 0, "uz_conv3x3_first_wgrad: null / unaligned pointer");
  CfArgs a = {};
  a.x = x;
  a.dy = dy;
  a.slab = static_cast<float*>(workspace);
  fill(a, N, C, H, W, Cout, lddy);
  const int grid = first_wgrad_grid(a.ntiles);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (Cout == 64) hipLaunchKernelGGL(conv_first_wgrad_kernel<2>, dim3(grid), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(conv_first_wgrad_kernel<1>, dim3(grid), dim3(256), 0, s, a);
  UZ_LAUNCH_CHECK("uz_conv3x3_first_wgrad");
  hipLaunchKernelGGL(conv_first_wgrad_reduce_kernel, dim3(Cout * 32 / 16), dim3(256), 0, s, a.slab, grid, Cout, C * 9, dw);
  UZ_LAUNCH_CHECK("uz_conv3x3_first_wgrad(reduce)");
  return UZ_OK;
}
