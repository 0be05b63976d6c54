// Pixel-major GEMM with an LDS-DMA pipeline for gfx950 (MI355X): the shapes of the hot path that are
// plain matrix products over the pixel index —
//   * 1x1 convolution and the im2col'd first convolution (reference: nn.Conv2d k1,
//     unet_zoo/models/attention_unet.py:11,18,25; first conv of unet.py:31 after uz_im2col3x3_nchw)
//   * ConvTranspose2d k2 s2 forward  = [P_in, Cin] x [Cin, 4*Cout] with a pixel-shuffle store
//   * ConvTranspose2d k2 s2 input-gradient = 2x2 gather of the fine grid, K = 4*Cout
//     (reference: unet_zoo/models/common_layers.py:104, backward via autograd a19)
// y[m][n] = bias[n] + sum_k A[m][k] * W[n][k],  A row m, K-step s = (tap, 128-byte channel slab).
//
// 512-thread workgroup = 256 consecutive pixels x BN output channels; every K-step brings one
// [256][128 B] activation tile and one [BN][128 B] weight tile into a 3-stage LDS ring by LDS-DMA
// (buffer_load ... lds, rows past the end are out of range in the descriptor -> zero fill), two
// steps ahead of the MFMAs behind a counted s_waitcnt vmcnt; one raw s_barrier per step.
// Same 128-byte-row / XOR-swizzled 16-byte-chunk LDS image and the same epilogue (bias, BatchNorm
// partial sums, bf16 tile transposed through LDS to 16-byte row stores) as uz_conv3x3.hip.
#include "uz_common.h"

namespace {

struct GArgs {
  const void* x;
  const void* w;
  void* y;
  const float* bias;
  float* stats;
  const void* res;   // optional (M, ldres) tensor added to the result (plain store only): residual sums, gradient sums
  int ldres;
  // BatchNorm-backward reduction in the epilogue (uz_conv_igemm_bnred, see uz_conv3x3.hip): y is the gradient of
  // relu(bn(bn_y)); the rows of `stats` receive sum(dz), sum(dz * xhat) instead of output statistics (plain store only)
  const void* bn_y;
  const float *bn_scale, *bn_shift, *bn_mean, *bn_invstd;
  int ld_bny;
  unsigned xbytes, wbytes;
  int M, H, W, Hin, Win, Cin, ldx, Nout, ldy, K, ntaps, mode, store, Co, tiles_m, Hout, Wout, dil;
  // batched products (uz_gemm_nt): blockIdx.z = matrix index, byte strides between consecutive matrices (0 = shared
  // operand); ldw = row stride of w in elements (the convolution paths: K)
  int ldw;
  long long xb, wb, yb, resb;
  // second batch level (uz_gemm_nt): blockIdx.z = b * nb2 + h; byte strides of the inner index h
  int nb2;
  long long xb2, wb2, yb2, resb2;
  // split-K (one-tap problems with few tiles and a long K loop: the 8 x 8 / 16 x 16 token maps of swin_unet_v2, the
  // spatial-reduction products of MISSFormer): blockIdx.z owns K slabs [z * cps, (z + 1) * cps) and leaves its fp32
  // accumulators in part[z][M][Nout]; igemm_split_reduce_kernel adds them in a fixed order, then bias / residual / rounding
  float* part;
  int ksplit, cps;
};

template <typename T> struct Mma3;
template <> struct Mma3<bf16_t> {
  // bf16 tiles are accumulated transposed (weights as the row operand, as in uz_conv3x3.hip): a lane owns 4
  // consecutive output channels of one pixel per register quad -> 8-byte LDS staging writes
  static __device__ __forceinline__ void run(const Vec16<bf16_t>& a, const Vec16<bf16_t>& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&b),
                                                *reinterpret_cast<const bf16x8*>(&a), c, 0, 0, 0);
  }
};
template <> struct Mma3<float> {
  static __device__ __forceinline__ void run(const Vec16<float>& a, const Vec16<float>& b, f32x16& c) {
#pragma unroll
    for (int t = 0; t < 4; ++t) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[t], b.v[t], c, 0, 0, 0);
  }
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr unsigned OOB = 0x80000000u;

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// keeps a 16-byte load unconditional: the loaded registers pass through an empty asm, so the optimiser cannot conclude that the
// value is needed on one path only and move the load under that path's branch (where hipcc 7.2 ends the block with vmcnt(0))
template <typename T> __device__ __forceinline__ void pin16(Vec16<T>& v) {
  unsigned* r = reinterpret_cast<unsigned*>(&v);
  asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));
}

// BM x BN tile per workgroup, NST stages of one 128-byte K slab each in the LDS ring (NST - 1 in flight).
// (256, 3) is the throughput shape.  A GEMM whose 256-row tiling leaves most CUs idle (the 16x16 and 8x8
// token maps of swin_unet_v2) is bound by the latency of its K loop -- slab s + 2 is requested when slab s is
// consumed, so a step costs half a memory round trip whatever the tile -- and takes (128, 4): twice the
// workgroups, and a third of a round trip per step.  NST = 2 serves K <= 128 (at most two slabs, both requested
// up front, the host guarantees it): 64 KB of LDS, so TWO workgroups share a CU and one's epilogue overlaps the
// other's loads -- the full-resolution Linear layers of swin_unet_v2 (K = 96) are one short tile per workgroup.
// SPLITK is a template argument, not a run-time branch: with `if (a.ksplit > 1)` in the shared epilogue every bf16
// instantiation kept its fp32 accumulators and their store addresses live beside the staging epilogue (+35..45 VGPRs;
// <128, 256, 3, BNRED> went from 253 registers to 256 + 105 spilled, round 4).  tests/test_kernel_resources.py guards it.
template <typename T, int BN, int BM, int NST, bool BNRED = false, bool SPLITK = false>
__global__ __launch_bounds__(512, (NST == 2 ? 2 : 1)) void gemm_dma_kernel(const GArgs a) {
  static_assert(!(SPLITK && BNRED), "the split-K form stores fp32 partial tiles only");
  constexpr int VEC = ElemTraits<T>::VEC;
  constexpr int ES = (int)sizeof(T);
  constexpr int BK = 8 * VEC;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int NAP = BM / 64;    // A pieces per wave per stage (BM / 8 pieces of 8 rows)
  constexpr int NBP = BN / 64;    // B pieces per wave per stage
  constexpr int WM = BM / 64, WN = 8 / WM;        // waves: WM (M) x WN (N), wave tile 64 x BN / WN
  constexpr int WTN = BN / WN, TN = WTN / 32;
  static_assert(TN >= 1 && (BM == 256 || BM == 128) && NST >= 2 && NST <= 4, "unsupported tile");
  __shared__ __attribute__((aligned(16))) char smem[NST * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.y * BN;
  const int kz = SPLITK ? (int)blockIdx.z : 0;
  const long long bz = SPLITK ? 0 : (int)blockIdx.z / a.nb2, bh = SPLITK ? 0 : (int)blockIdx.z - (int)bz * a.nb2;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(static_cast<const char*>(a.x)) + bz * a.xb + bh * a.xb2, 0, a.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(static_cast<const char*>(a.w)) + bz * a.wb + bh * a.wb2, 0, a.wbytes, 0x00020000);
  T* __restrict__ yg = reinterpret_cast<T*>(static_cast<char*>(a.y) + bz * a.yb + bh * a.yb2);
  const T* __restrict__ resg =
      a.res ? reinterpret_cast<const T*>(static_cast<const char*>(a.res) + bz * a.resb + bh * a.resb2) : nullptr;
  const int HW = a.H * a.W;
  const int ncb = (a.Cin + BK - 1) / BK;  // the last slab of a tap may be partial: zero-filled
  const int s_beg = SPLITK ? kz * a.cps : 0;                 // first K slab of this workgroup
  const int nsteps = SPLITK ? ((a.ntaps * ncb - s_beg < a.cps) ? a.ntaps * ncb - s_beg : a.cps) : a.ntaps * ncb;   // and how many

  unsigned b_row_off[NBP];
  int b_coff[NBP];   // byte offset of this lane's logical chunk inside a 128-byte slab
#pragma unroll
  for (int i = 0; i < NBP; ++i) {
    const int n = (wave + 8 * i) * 8 + (lane >> 3);
    b_row_off[i] = (n0 + n < a.Nout) ? (unsigned)(n0 + n) * (unsigned)a.ldw * ES : OOB;
    b_coff[i] = (((lane & 7) ^ ((n >> 1) & 7)) * VEC) * ES;
  }
  const int cin_bytes = a.Cin * ES;
  int b_frag_off[TN], b_sw[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int brow = wn * WTN + j * 32 + l31;
    b_frag_off[j] = brow * 128;
    b_sw[j] = (brow >> 1) & 7;
  }
  int a_frag_off[2], a_sw[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int arow = wm * 64 + i * 32 + l31;
    a_frag_off[i] = arow * 128;
    a_sw[i] = (arow >> 1) & 7;
  }
  float bv[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * WTN + j * 32 + l31;
    bv[j] = (a.bias != nullptr && n < a.Nout) ? a.bias[n] : 0.f;
  }
  float s1[TN], s2[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) s1[j] = s2[j] = 0.f;
  // bf16 path: bias of accumulator register r of N tile j (channel wn*WTN + 32 j + (r&3) + 8(r>>2) + 4 lh) and
  // the statistics of this thread's fixed 16-byte channel chunk in the coalesced read-back
  float bq[TN][16], sq1[VEC], sq2[VEC];
  if constexpr (sizeof(T) == 2 && !BNRED) {   // (the fused-reduction launches carry no bias)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * WTN + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        bq[j][r] = (a.bias != nullptr && n < a.Nout) ? a.bias[n] : 0.f;
      }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) sq1[e] = sq2[e] = 0.f;

  // stage loads of tile `m0t` (pixel offsets recomputed per call: a few VALU ops, no per-tile register arrays,
  // so the NEXT tile's first stages can be issued from the current tile's epilogue)
  auto issue = [&](int m0t, int stage, int sl) {
      const int s = s_beg + sl;
      const int tap = s / ncb, cb = s - tap * ncb;
      const int dpix = (a.mode == UZ_TAPS_GATHER2X2) ? (tap >> 1) * a.Win + (tap & 1) : 0;
      const int s2y = (tap * 11) >> 5, s2x = tap - 3 * s2y;   // UZ_TAPS_CONV_S2: tap = 3 ty + tx
      const int slab = cb * BK * ES;            // byte offset of the slab inside the tap's channels
      char* sA = smem + stage * STAGE;
      char* sBt = sA + A_BYTES;
#pragma unroll
      for (int i = 0; i < NAP; ++i) {
        const int row = (wave + 8 * i) * 8 + (lane >> 3);
        const int m = m0t + row;
        const int a_coff = (((lane & 7) ^ ((row >> 1) & 7)) * VEC) * ES;
        int a_pix = -1;
        if (m < a.M) {
          if (a.mode == UZ_TAPS_CONV && a.ntaps == 1) {
            a_pix = m;
          } else if (a.mode == UZ_TAPS_CONV) {   // 3x3, dilation d, zero padding d (REBNCONV, u2net.py:10-13)
            const int img = m / HW, rem = m - img * HW;
            const int h = rem / a.W, w = rem - h * a.W;
            const int hh = h + (s2y - 1) * a.dil, ww = w + (s2x - 1) * a.dil;
            if ((unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)a.W) a_pix = (img * a.H + hh) * a.W + ww;
          } else {
            const int img = m / HW, rem = m - img * HW;
            const int h = rem / a.W, w = rem - h * a.W;
            if (a.mode == UZ_TAPS_CONV_S2) {   // stride 2, padding 1: (2h + ty - 1, 2w + tx - 1), zero outside
              const int hh = 2 * h + s2y - 1, ww = 2 * w + s2x - 1;
              if ((unsigned)hh < (unsigned)a.Hin && (unsigned)ww < (unsigned)a.Win) a_pix = (img * a.Hin + hh) * a.Win + ww;
            } else {
              a_pix = (img * a.Hin + 2 * h) * a.Win + 2 * w;
            }
          }
        }
        const bool ok = a_pix >= 0 && slab + a_coff < cin_bytes;
        const unsigned off = ok ? (unsigned)(a_pix + dpix) * (unsigned)(a.ldx * ES) + slab + a_coff : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(sA + (wave + 8 * i) * 1024), 16, off, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < NBP; ++i) {
        const bool ok = b_row_off[i] != OOB && slab + b_coff[i] < cin_bytes;
        const unsigned off = ok ? b_row_off[i] + (unsigned)(tap * cin_bytes) + slab + b_coff[i] : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr_t)(sBt + (wave + 8 * i) * 1024), 16, off, 0, 0, 0);
      }
  };

  // C staging sits at the END of the ring, so stage 0 (BN = 64: stages 0 and 1) stays free during an epilogue
  constexpr int RSCB = BN * 2 + 16;                      // bf16 staging row stride
  constexpr int SC_OFF = NST * STAGE - BM * RSCB;
  constexpr int PRE = (SC_OFF >= 2 * STAGE) ? 2 : ((SC_OFF >= STAGE) ? 1 : 0);   // stages that may be prefetched
  int pre = 0;   // stages of the current tile already issued by the previous tile's epilogue
  for (int tile = blockIdx.x; tile < a.tiles_m; tile += gridDim.x) {
    const int m0 = tile * BM;

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    __builtin_amdgcn_s_barrier();  // previous tile's staging reads are finished everywhere
    if (pre < 1) issue(m0, 0, 0);
    if (nsteps > 1 && pre < 2) issue(m0, 1, 1);
    if (NST == 4 && nsteps > 2) issue(m0, 2, 2);
#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
      // stage s must have landed; up to NST - 2 younger stages stay in flight
      if (NST == 4 && s + 2 < nsteps) {
        wait_vmcnt<2 * (NAP + NBP)>();
      } else if (s + 1 < nsteps) {
        wait_vmcnt<NAP + NBP>();
      } else {
        wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();
      if (NST > 2 && s + NST - 1 < nsteps) issue(m0, (s + NST - 1) % NST, s + NST - 1);
      const char* sA = smem + (s % NST) * STAGE;
      const char* sBt = sA + A_BYTES;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int lc = 2 * q + lh;
        Vec16<T> af[2], bf[TN];
#pragma unroll
        for (int i = 0; i < 2; ++i)
          af[i] = *reinterpret_cast<const Vec16<T>*>(sA + a_frag_off[i] + ((lc ^ a_sw[i]) << 4));
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bf[j] = *reinterpret_cast<const Vec16<T>*>(sBt + b_frag_off[j] + ((lc ^ b_sw[j]) << 4));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) Mma3<T>::run(af[i], bf[j], acc[i][j]);
      }
    }

    // ---- epilogue ----------------------------------------------------------------------------
    // output row of local pixel ml (0..255): plain -> pixel m0+ml; shuffle -> (2h+a, 2w+b)
    auto out_row = [&](int ml, int ab) -> long long {
      const int m = m0 + ml;
      if (m >= a.M) return -1;
      if (a.store == UZ_STORE_PLAIN) return m;
      const int img = m / HW, rem = m - img * HW;
      const int h = rem / a.W, w = rem - h * a.W;
      return ((long long)img * a.Hout + 2 * h + (ab >> 1)) * a.Wout + 2 * w + (ab & 1);
    };
    // fp32 path: a whole BN tile belongs to one sub-pixel (Co % BN == 0, checked on the host); the bf16 path decides per chunk
    const int ab = (a.store == UZ_STORE_SHUFFLE2X2) ? n0 / a.Co : 0;
    const int co0 = (a.store == UZ_STORE_SHUFFLE2X2) ? n0 - ab * a.Co : n0;
    pre = 0;
    if constexpr (SPLITK) {   // fp32 accumulators of this K range: accumulator column = pixel, register quad = 4 consecutive channels
      float* part = a.part + (size_t)kz * a.M * a.Nout;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm * 64 + i * 32 + l31;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int n = n0 + wn * WTN + j * 32 + 8 * q + 4 * lh;
            if (m < a.M && n < a.Nout)   // Nout is a multiple of 8 here: a quad is inside or outside as a whole
              *reinterpret_cast<float4*>(part + (size_t)m * a.Nout + n) =
                  make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
          }
      }
      continue;
    } else if constexpr (sizeof(T) == 2) {
      constexpr int RSC = RSCB;
      static_assert(BM * RSC <= NST * STAGE, "C staging must fit the ring");
      __builtin_amdgcn_s_barrier();   // every wave has finished reading this tile's A / B stages
      {   // the next tile's first stage(s) stream in while this tile is staged and stored
        const int next = tile + gridDim.x;
        if (next < a.tiles_m && PRE > 0) {
          issue(next * BM, 0, 0);
          pre = 1;
          if (PRE > 1 && nsteps > 1) {
            issue(next * BM, 1, 1);
            pre = 2;
          }
        }
      }
      char* sC = smem + SC_OFF;
      // accumulator column = pixel (l31) of M tile (wm, i); register quad q = 4 consecutive channels
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        char* rowp = sC + (wm * 64 + i * 32 + l31) * RSC;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            bf16x4 pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = (bf16_t)(BNRED ? acc[i][j][4 * q + e] : acc[i][j][4 * q + e] + bq[j][4 * q + e]);
            *reinterpret_cast<bf16x4*>(rowp + (wn * WTN + j * 32 + 8 * q + 4 * lh) * ES) = pk;
          }
      }
      // the staging writes must have EXECUTED, not just issued, before another wave reads them: s_barrier alone
      // does not wait for the LDS queue (it went unnoticed until LDS-DMA traffic shared the write path)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      constexpr int CPR = BN * ES / 16;
      static_assert(512 % CPR == 0, "a thread keeps one channel chunk");
      const int cc = tid % CPR;
      const bool cok = n0 + cc * VEC < a.Nout;
      // this thread's 16-byte channel chunk: its sub-pixel and its channel inside the sub-pixel.  Per chunk, not per tile: a
      // tile may straddle sub-pixels when Co is not a multiple of BN (PatchExpand with Co = 96: swin_unet_v2.py:343-351)
      const int nthr = n0 + cc * VEC;
      const bool shuf = !BNRED && a.store == UZ_STORE_SHUFFLE2X2;   // (the fused-reduction form stores plain only)
      const int abt = (shuf && cok) ? nthr / a.Co : 0;
      const int cot = shuf ? nthr - abt * a.Co : nthr;
      // all of a thread's chunks are requested from LDS before the first store (see uz_conv3x3.hip)
      constexpr int NPASS = BM * CPR / 512, RPP = 512 / CPR;
      Vec16<T> vb[NPASS];
#pragma unroll
      for (int k = 0; k < NPASS; ++k)
        vb[k] = *reinterpret_cast<const Vec16<T>*>(sC + (tid / CPR + k * RPP) * RSC + cc * 16);
      Vec16<T> yb[NPASS];
      float bsc[VEC], bsh[VEC];   // (sum(dz * y) is accumulated here; mean / invstd enter once, in the row written at the end)
      if constexpr (BNRED) {   // pre-activation values of this thread's pixels and the channel constants
        const T* by = static_cast<const T*>(a.bn_y);
#pragma unroll
        for (int k = 0; k < NPASS; ++k) {
          // unconditional (element 0 for a row outside the image / a chunk beyond the channels: such a chunk is never used):
          // as `ok ? load : 0` every pass's load was waited for -- with the next tile's LDS-DMA stages in front of it --
          // before the next pass's was issued
          const long long orow = out_row(tid / CPR + k * RPP, 0);
          yb[k] = ld16(by + ((orow >= 0 && cok) ? (size_t)orow * a.ld_bny + n0 + cc * VEC : 0));
        }
#pragma unroll
        for (int k = 0; k < NPASS; ++k) pin16(yb[k]);   // (opaque: a value used only under `ok` is otherwise loaded under a branch again)
        const int ch0 = cok ? n0 + cc * VEC : 0;
#pragma unroll
        for (int e = 0; e < VEC; e += 4) {
          *reinterpret_cast<f32x4*>(&bsc[e]) = *reinterpret_cast<const f32x4*>(a.bn_scale + ch0 + e);
          *reinterpret_cast<f32x4*>(&bsh[e]) = *reinterpret_cast<const f32x4*>(a.bn_shift + ch0 + e);
        }
      }
      if (resg != nullptr) {   // y = (x W^T + b) + res, rounded as a separate add of the stored result would be
        const T* rg = resg;
        Vec16<T> rb[NPASS];
#pragma unroll
        for (int k = 0; k < NPASS; ++k) {
          const long long orow = out_row(tid / CPR + k * RPP, abt);
          rb[k] = ld16(rg + ((orow >= 0 && cok) ? (size_t)orow * a.ldres + cot : 0));   // (unconditional, as above; the sum of a chunk that is not stored is not used)
        }
#pragma unroll
        for (int k = 0; k < NPASS; ++k) pin16(rb[k]);
#pragma unroll
        for (int k = 0; k < NPASS; ++k)
#pragma unroll
          for (int e = 0; e < VEC; ++e) vb[k].v[e] = (T)((float)vb[k].v[e] + (float)rb[k].v[e]);
      }
#pragma unroll
      for (int k = 0; k < NPASS; ++k) {
        const long long orow = out_row(tid / CPR + k * RPP, abt);
        if (orow >= 0 && cok) {
          st16(yg + (size_t)orow * a.ldy + cot, vb[k]);
          if constexpr (BNRED) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
              const float yv = (float)yb[k].v[e];
              const float dz = fmaf(yv, bsc[e], bsh[e]) > 0.f ? (float)vb[k].v[e] : 0.f;
              sq1[e] += dz;
              sq2[e] += dz * yv;
            }
          } else {
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const float fv = (float)vb[k].v[e];
            sq1[e] += fv;
            sq2[e] += fv * fv;
          }
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = wn * WTN + j * 32 + l31;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int ml = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const long long orow = out_row(ml, ab);
            if (orow >= 0 && n0 + col < a.Nout) {
              T tv = (T)(acc[i][j][r] + bv[j]);
              if (resg != nullptr) tv += resg[(size_t)orow * a.ldres + co0 + col];
              yg[(size_t)orow * a.ldy + co0 + col] = tv;
              const float fv = (float)tv;
              s1[j] += fv;
              s2[j] += fv * fv;
            }
          }
        }
      }
    }
  }

  if constexpr (SPLITK) return;
  if constexpr (sizeof(T) == 2) {
    if (a.stats != nullptr) {
      constexpr int CPR = BN * ES / 16;
      wait_vmcnt<0>();
      __syncthreads();
      float* red = reinterpret_cast<float*>(smem);  // [512][2 * VEC]
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        red[tid * 2 * VEC + e] = sq1[e];
        red[tid * 2 * VEC + VEC + e] = sq2[e];
      }
      __syncthreads();
      if (tid < 2 * BN) {  // (which, channel): sum the 512 / CPR threads that own this channel's chunk
        const int which = tid / BN, ch = tid - which * BN;
        const int cc = ch / VEC, e = ch - cc * VEC;
        float t = 0.f;
        for (int k = cc; k < 512; k += CPR) t += red[k * 2 * VEC + which * VEC + e];
        if constexpr (BNRED) {   // sum(dz * xhat) = invstd * (sum(dz * y) - mean * sum(dz))
          if (which == 1 && n0 + ch < a.Nout) {
            float t0 = 0.f;
            for (int k = cc; k < 512; k += CPR) t0 += red[k * 2 * VEC + e];
            t = a.bn_invstd[n0 + ch] * (t - a.bn_mean[n0 + ch] * t0);
          }
        }
        if (n0 + ch < a.Nout) a.stats[((size_t)blockIdx.x * 2 + which) * a.Nout + n0 + ch] = t;
      }
    }
    return;
  }
  if (a.stats != nullptr) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      s1[j] += __shfl_xor(s1[j], 32);
      s2[j] += __shfl_xor(s2[j], 32);
    }
    wait_vmcnt<0>();
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

}  // namespace

int uz_gemm_dma_plan(const uz_conv_desc* d, UzGemmPlan* p) {
  const int vec = d->dtype == UZ_BF16 ? 8 : 4, es = d->dtype == UZ_BF16 ? 2 : 4, bk = 8 * vec;
  const bool conv1 = d->taps_mode == UZ_TAPS_CONV && d->ntaps == 1;
  // dilated 3x3 (the direct kernel takes dilation 1; REBNCONV with dirate 2 / 4 / 8, u2net.py:10-13): nine taps at
  // (h + (ty-1) d, w + (tx-1) d), zero outside.  Measured on the u2net step against the first-generation kernel with
  // its 9-way tap split: 27.8 vs 29.1 ms taking every dilated layer here, 28.8 ms taking only the large ones.
  const bool conv9 = d->taps_mode == UZ_TAPS_CONV && d->ntaps == 9 && d->dil > 1 && d->store_mode == UZ_STORE_PLAIN &&
                     !(uz_tune_flags() & 0x20000000);
  const bool gath = (d->taps_mode == UZ_TAPS_GATHER2X2 && d->ntaps == 4) || (d->taps_mode == UZ_TAPS_CONV_S2 && d->ntaps == 9);
  if (!conv1 && !gath && !conv9) return 0;
  (void)bk;
  if (d->Cin % vec != 0 || d->Nout % vec != 0 || d->ldy % vec != 0) return 0;
  const long long pin = (long long)d->N * d->Hin * d->Win;
  const long long xbytes = (pin - 1) * d->ldx * es + (long long)d->Cin * es;
  const long long wbytes = (long long)d->Nout * d->ntaps * d->Cin * es;
  if (xbytes >= (1LL << 31) || wbytes >= (1LL << 31)) return 0;
  p->bn = d->Nout <= 64 ? 64 : 128;
  if (d->store_mode == UZ_STORE_SHUFFLE2X2) {
    if (gath) return 0;
    if (d->Co % 128 == 0) p->bn = 128;
    else if (d->Co % 64 == 0) p->bn = 64;
    else if (d->dtype == UZ_BF16 && d->Co % vec == 0) p->bn = d->Nout <= 64 ? 64 : 128;   // tiles straddle sub-pixels: per-chunk store
    else return 0;
  }
  const long long M = (long long)d->N * d->H * d->W;
  p->tiles_n = (d->Nout + p->bn - 1) / p->bn;
  // latency shape (see the kernel): when 256-row tiles would occupy at most half of the CUs
  p->bm = (p->bn == 128 && ((M + 255) / 256) * p->tiles_n * 2 <= UZ_NUM_CU && !(uz_tune_flags() & 0x200000)) ? 128 : 256;
  const int nsteps = d->ntaps * ((d->Cin + 8 * vec - 1) / (8 * vec));
  p->nst = p->bm == 128 ? 4 : 3;
  if (p->bn == 128 && nsteps <= 2 && !(uz_tune_flags() & 0x800000)) {   // two resident workgroups per CU
    p->bm = 128;
    p->nst = 2;
  }
  // 64 output channels and at most two K slabs (the first convolution through im2col: K = 32; ConvTranspose 128 -> 64):
  // a tile is one load, a few MFMAs and a 32 KB store burst -- two workgroups per CU (80 KB of LDS each) overlap them
  if (p->bn == 64 && nsteps <= 2 && !(uz_tune_flags() & 0x1000000)) p->nst = 2;
  p->tiles_m = (int)((M + p->bm - 1) / p->bm);
  int cap = (p->nst == 2 ? 2 : 1) * UZ_NUM_CU / p->tiles_n;
  if (cap < 1) cap = 1;
  p->grid_m = p->tiles_m < cap ? p->tiles_m : cap;
  // split-K: few tiles walking a long K loop (a step costs a third of a memory round trip whatever the tile, see the
  // kernel): M = 1024 x N = 768 x K = 2304 ran 48 workgroups for 20.7 us.  Only with a workspace, a plain store and
  // neither statistics nor a fused reduction (the launch checks those); >= 4 slabs per range.
  p->ksplit = 1;
  p->cps = nsteps;
  const long long tiles = (long long)p->tiles_m * p->tiles_n;
  // (the reduce pass is a launch of its own, ~8 us: 48 tiles x 36 slabs -- swin's 8 x 8 maps -- came out even, 20.7 us either way)
  // The dilated nine-tap form (K = 9 Cin: 72 slabs for u2net's 512-channel RSU4F on 16 x 16 / 32 x 32 maps) splits too.
  if ((conv1 || conv9) && d->dtype == UZ_BF16 && d->store_mode == UZ_STORE_PLAIN && p->nst > 2 &&
      ((tiles <= 32 && nsteps >= 16) || (tiles <= 64 && nsteps >= 48) || (tiles <= 128 && nsteps >= 64)) &&
      !(uz_tune_flags() & 0x80)) {
    long long ks = UZ_NUM_CU / tiles;
    if (ks > nsteps / 4) ks = nsteps / 4;
    if (ks > 8) ks = 8;
    if (ks >= 2) {
      p->cps = (int)((nsteps + ks - 1) / ks);
      p->ksplit = (nsteps + p->cps - 1) / p->cps;
      p->grid_m = p->tiles_m;   // one tile per workgroup
    }
  }
  return 1;
}

long long uz_gemm_dma_workspace_bytes(const uz_conv_desc* d) {
  UzGemmPlan p;
  if (!uz_gemm_dma_plan(d, &p) || p.ksplit <= 1) return 0;
  return (long long)p.ksplit * d->N * d->H * d->W * d->Nout * (long long)sizeof(float);
}

template <typename T>
static int gemm_launch_grid(const UzGemmPlan& p, const GArgs& a, dim3 grid, hipStream_t s) {
  dim3 block(512);
  if constexpr (sizeof(T) == 2) {
    if (a.ksplit > 1) {   // the plan splits three-/four-stage shapes only
      if (p.bn == 64) hipLaunchKernelGGL((gemm_dma_kernel<T, 64, 256, 3, false, true>), grid, block, 0, s, a);
      else if (p.bm == 128) hipLaunchKernelGGL((gemm_dma_kernel<T, 128, 128, 4, false, true>), grid, block, 0, s, a);
      else hipLaunchKernelGGL((gemm_dma_kernel<T, 128, 256, 3, false, true>), grid, block, 0, s, a);
      UZ_LAUNCH_CHECK("uz_conv_igemm(gemm_dma, split-K)");
      return UZ_OK;
    }
    if (a.bn_y != nullptr) {
      if (p.bn == 64 && p.nst == 2) hipLaunchKernelGGL((gemm_dma_kernel<T, 64, 256, 2, true>), grid, block, 0, s, a);
      else if (p.bn == 64) hipLaunchKernelGGL((gemm_dma_kernel<T, 64, 256, 3, true>), grid, block, 0, s, a);
      else if (p.nst == 2) hipLaunchKernelGGL((gemm_dma_kernel<T, 128, 128, 2, true>), grid, block, 0, s, a);
      else if (p.bm == 128) hipLaunchKernelGGL((gemm_dma_kernel<T, 128, 128, 4, true>), grid, block, 0, s, a);
      else hipLaunchKernelGGL((gemm_dma_kernel<T, 128, 256, 3, true>), grid, block, 0, s, a);
      UZ_LAUNCH_CHECK("uz_conv_igemm_bnred(gemm_dma)");
      return UZ_OK;
    }
  }
  if (p.bn == 64 && p.nst == 2) hipLaunchKernelGGL((gemm_dma_kernel<T, 64, 256, 2>), grid, block, 0, s, a);
  else if (p.bn == 64) hipLaunchKernelGGL((gemm_dma_kernel<T, 64, 256, 3>), grid, block, 0, s, a);
  else if (p.nst == 2) hipLaunchKernelGGL((gemm_dma_kernel<T, 128, 128, 2>), grid, block, 0, s, a);
  else if (p.bm == 128) hipLaunchKernelGGL((gemm_dma_kernel<T, 128, 128, 4>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((gemm_dma_kernel<T, 128, 256, 3>), grid, block, 0, s, a);
  UZ_LAUNCH_CHECK("uz_conv_igemm(gemm_dma)");
  return UZ_OK;
}

template <typename T>
static int gemm_launch_t(const UzGemmPlan& p, const GArgs& a, hipStream_t s) {
  return gemm_launch_grid<T>(p, a, dim3(p.grid_m, p.tiles_n), s);
}

int uz_gemm_dma_launch(const uz_conv_desc* d, const UzGemmPlan& p_in, const void* x, const void* w,
                       const float* bias, void* y, float* stats, hipStream_t s, const void* res, int ldres,
                       const UzBnRed* br, float* part) {
  const int es = d->dtype == UZ_BF16 ? 2 : 4;
  UzGemmPlan p = p_in;
  if (p.ksplit > 1 && part == nullptr) {   // no workspace: the unsplit plan
    UzGemmPlan q = p;
    int cap = UZ_NUM_CU / p.tiles_n;
    if (cap < 1) cap = 1;
    q.grid_m = p.tiles_m < cap ? p.tiles_m : cap;
    q.ksplit = 1;
    p = q;
  }
  GArgs a;
  a.part = p.ksplit > 1 ? part : nullptr;
  a.ksplit = p.ksplit;
  a.cps = p.cps;
  if (p.ksplit > 1)
    UZ_REQUIRE(stats == nullptr && br == nullptr && d->dtype == UZ_BF16 && d->store_mode == UZ_STORE_PLAIN,
               "uz_conv_igemm(split-K GEMM): plain bf16 products; statistics come from the reduce pass");
  a.bn_y = br ? br->y : nullptr;
  a.bn_scale = br ? br->scale : nullptr;
  a.bn_shift = br ? br->shift : nullptr;
  a.bn_mean = br ? br->mean : nullptr;
  a.bn_invstd = br ? br->invstd : nullptr;
  a.ld_bny = br ? br->ldy : 0;
  if (br) UZ_REQUIRE(d->dtype == UZ_BF16 && d->store_mode == UZ_STORE_PLAIN && stats != nullptr && res == nullptr,
                     "uz_conv_igemm_bnred: bf16 LDS-DMA GEMMs with a plain store only");
  a.x = x;
  a.w = w;
  a.y = y;
  a.bias = bias;
  a.stats = stats;
  a.res = res;
  a.ldres = ldres;
  a.xbytes = (unsigned)(((long long)d->N * d->Hin * d->Win - 1) * d->ldx * es + (long long)d->Cin * es);
  a.wbytes = (unsigned)((long long)d->Nout * d->ntaps * d->Cin * es);
  a.M = d->N * d->H * d->W;
  a.H = d->H;
  a.W = d->W;
  a.Hin = d->Hin;
  a.Win = d->Win;
  a.Hout = d->Hout ? d->Hout : 2 * d->H;
  a.Wout = d->Wout ? d->Wout : 2 * d->W;
  a.Cin = d->Cin;
  a.ldx = d->ldx;
  a.Nout = d->Nout;
  a.ldy = d->ldy;
  a.K = d->ntaps * d->Cin;
  a.ntaps = d->ntaps;
  a.mode = d->taps_mode;
  a.store = d->store_mode;
  a.Co = d->Co;
  a.dil = d->dil;
  a.tiles_m = p.tiles_m;
  a.ldw = a.K;
  a.xb = a.wb = a.yb = a.resb = 0;
  a.nb2 = 1;
  a.xb2 = a.wb2 = a.yb2 = a.resb2 = 0;
  if (p.ksplit > 1) return gemm_launch_grid<bf16_t>(p, a, dim3(p.grid_m, p.tiles_n, p.ksplit), s);
  return d->dtype == UZ_BF16 ? gemm_launch_t<bf16_t>(p, a, s) : gemm_launch_t<float>(p, a, s);
}

// ---- uz_gemm_nt: batched y_b = x_b w_b^T (+ bias, + res_b) on the same kernel --------------------------------------
static int gemm_nt_plan(const uz_gemm_desc* d, UzGemmPlan* p) {
  const int vec = d->dtype == UZ_BF16 ? 8 : 4, es = d->dtype == UZ_BF16 ? 2 : 4;
  UZ_REQUIRE(d->dtype == UZ_BF16 || d->dtype == UZ_F32, "uz_gemm_nt: dtype");
  UZ_REQUIRE(d->batch >= 1 && d->M >= 1 && d->N >= 1 && d->K >= 1 && d->batch2 >= 0 &&
                 (long long)d->batch * (d->batch2 > 1 ? d->batch2 : 1) <= 65535,
             "uz_gemm_nt: empty problem / more than 65535 matrices");
  UZ_REQUIRE(d->xb2 % vec == 0 && d->wb2 % vec == 0 && d->yb2 % vec == 0 && d->resb2 % vec == 0,
             "uz_gemm_nt: inner matrix strides must be multiples of 16 bytes");
  UZ_REQUIRE(d->K % vec == 0 && d->N % vec == 0 && d->ldx % vec == 0 && d->ldw % vec == 0 && d->ldy % vec == 0 &&
                 d->ldres % vec == 0,
             "uz_gemm_nt: K, N and the row strides must be multiples of 16 bytes");
  UZ_REQUIRE(d->ldx >= d->K && d->ldw >= d->K && d->ldy >= d->N, "uz_gemm_nt: row strides shorter than the rows");
  UZ_REQUIRE(d->xb % vec == 0 && d->wb % vec == 0 && d->yb % vec == 0 && d->resb % vec == 0,
             "uz_gemm_nt: matrix strides must be multiples of 16 bytes");
  const long long xbytes = ((long long)d->M - 1) * d->ldx * es + (long long)d->K * es;
  const long long wbytes = ((long long)d->N - 1) * d->ldw * es + (long long)d->K * es;
  UZ_REQUIRE(xbytes < (1LL << 31) && wbytes < (1LL << 31), "uz_gemm_nt: one matrix must stay below 2 GB");
  p->bn = d->N <= 64 ? 64 : 128;
  p->tiles_n = (d->N + p->bn - 1) / p->bn;
  const long long per = (long long)p->tiles_n * d->batch * (d->batch2 > 1 ? d->batch2 : 1);
  p->bm = (p->bn == 128 && (((long long)d->M + 255) / 256) * per * 2 <= UZ_NUM_CU) ? 128 : 256;
  const int nsteps = (d->K + 8 * vec - 1) / (8 * vec);
  p->nst = p->bm == 128 ? 4 : 3;
  if (nsteps <= 2) {
    if (p->bn == 128) p->bm = 128;
    p->nst = 2;
  }
  p->ksplit = 1;
  p->cps = nsteps;
  p->tiles_m = (d->M + p->bm - 1) / p->bm;
  long long cap = (p->nst == 2 ? 2 : 1) * (long long)UZ_NUM_CU / per;
  if (cap < 1) cap = 1;
  p->grid_m = p->tiles_m < cap ? p->tiles_m : (int)cap;
  return UZ_OK;
}

extern "C" int uz_gemm_nt(const uz_gemm_desc* d, const void* x, const void* w, const float* bias, const void* res,
                          void* y, void* stream) {
  UZ_REQUIRE(d && x && w && y, "uz_gemm_nt: null pointer");
  UzGemmPlan p;
  const int rc = gemm_nt_plan(d, &p);
  if (rc != UZ_OK) return rc;
  UZ_REQUIRE((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)res) & 15) == 0, "uz_gemm_nt: operands must be 16-byte aligned");
  const int es = d->dtype == UZ_BF16 ? 2 : 4;
  GArgs a;
  a.bn_y = nullptr;
  a.bn_scale = a.bn_shift = a.bn_mean = a.bn_invstd = nullptr;
  a.ld_bny = 0;
  a.part = nullptr;
  a.ksplit = 1;
  a.cps = 0;
  a.x = x;
  a.w = w;
  a.y = y;
  a.bias = bias;
  a.stats = nullptr;
  a.res = res;
  a.ldres = d->ldres;
  a.xbytes = (unsigned)(((long long)d->M - 1) * d->ldx * es + (long long)d->K * es);
  a.wbytes = (unsigned)(((long long)d->N - 1) * d->ldw * es + (long long)d->K * es);
  a.M = d->M;
  a.H = 1;
  a.W = d->M;
  a.Hin = 1;
  a.Win = d->M;
  a.Hout = 2;
  a.Wout = 2 * d->M;
  a.Cin = d->K;
  a.ldx = d->ldx;
  a.Nout = d->N;
  a.ldy = d->ldy;
  a.K = d->K;
  a.ntaps = 1;
  a.mode = UZ_TAPS_CONV;
  a.store = UZ_STORE_PLAIN;
  a.Co = d->N;
  a.dil = 1;
  a.tiles_m = p.tiles_m;
  a.ldw = d->ldw;
  a.xb = d->xb * es;
  a.wb = d->wb * es;
  a.yb = d->yb * es;
  a.resb = d->resb * es;
  a.nb2 = d->batch2 > 1 ? d->batch2 : 1;
  a.xb2 = d->xb2 * es;
  a.wb2 = d->wb2 * es;
  a.yb2 = d->yb2 * es;
  a.resb2 = d->resb2 * es;
  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 grid(p.grid_m, p.tiles_n, d->batch * a.nb2), block(512);
  if (d->dtype == UZ_BF16) return gemm_launch_grid<bf16_t>(p, a, grid, s);
  return gemm_launch_grid<float>(p, a, grid, s);
}
