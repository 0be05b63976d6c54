// Weight gradient of a 3x3 convolution (stride 1, dilation 1), bf16, for gfx950 (MI355X):
//     dW[co][ci][ty][tx] = sum_p dy[p][co] * x[p + (ty-1, tx-1)][ci]
// (autograd weight-gradient of nn.Conv2d k3 p1: reference unet_zoo/models/common_layers.py:28,31,
// entered from loss.backward(), unet_zoo/utils/training_loop.py:119).
//
// The reduction index is the pixel.  A K-step is 64 pixels of one image (one 64-pixel row
// segment, or 64/W whole rows when W is 16 or 32), so every 16-pixel MFMA sub-step lies inside
// one image row and a tap is a constant row offset in the LDS image of x:
//   L tile (dy):  64 pixels x BI channels
//   R tile (x) :  the same pixels plus a one-pixel halo left/right (and, when the workgroup owns
//                 all nine taps, one row above/below), out-of-image pixels zero-filled by the
//                 buffer descriptor's range check
// both brought in by LDS-DMA (buffer_load ... lds) through a 3-stage ring, two K-steps ahead of
// the MFMAs behind a counted s_waitcnt vmcnt, one raw s_barrier per K-step.  One R tile serves
// three (config A: 128x128 channel tile, one kernel row per workgroup) or nine taps (config D:
// 64x64 channel tile) of MFMAs, so LDS-fill traffic per MFMA is 3-9x lower than one tap per load.
// Tiles are pixel-major [pixel][channel]; fragments come through ds_read_b64_tr_b16 (hardware
// transpose).  64-byte granules of a row are XOR-swizzled with the pixel index (on the DMA
// source address and on the read address) so the four pixel rows of a transposed read fall into
// distinct banks for any tap shift.
// Each (split, k-group) writes a coalesced fp32 slab [tap][Ci][Cj]; uz_wgrad's reduce kernel sums
// the slabs in fixed order and transposes to OIHW.
// (Round 3 tried the ping-pong schedule of uz_conv3x3_pp.hip on the 128 x 128 three-tap configuration -- one wave group
// reads all 40 transposed fragments of a K-step and issues its five DMA pieces while the other issues 24 MFMAs: correct,
// 5-10 % SLOWER on every layer (profiles/r03_wgrad_pingpong_rejected.txt).  With 40 KB of LDS-DMA and 20 KB of transposed
// reads per wave and 768 MFMA cycles the read phase is the longer one; the interleaved issue below hides more of it.)
#include "uz_common.h"

namespace {

struct Wg2Args {
  const void* L;
  const void* R;
  float* slab;
  unsigned lbytes, rbytes;
  int N, H, W, Ci, ldl, Cj, ldr;
  int KW, kw_log2, KR;  // K-step = KR rows x KW columns = 64 pixels
  int units;            // K-steps in the whole tensor
  int upb;              // K-steps per split
  int tiles_j;
  int r_up;  // 1: R lives at (H/2, W/2) and is read through nearest x2 upsampling
  int gather;   // 1: 2x2 gather (ConvTranspose2d k2 s2 / PatchExpand weight gradient): R lives on the (Hr, Wr) grid
                // of 2H (+1) x 2W (+1) pixels and tap t = blockIdx.y reads pixel (2h + (t >> 1), 2w + (t & 1));
                // 2: stride-2 3x3 convolution (UZ_TAPS_CONV_S2): tap t = 3 ty + tx reads (2h + ty - 1, 2w + tx - 1), zero outside
  int Hr, Wr;
  int dil;   // gather == 3: dilated 3x3 (REBNCONV dirate 2 / 4 / 8, u2net.py:10-13): tap t reads (h + (ty-1) dil, w + (tx-1) dil), zero outside
  int flags; // tuning switches (env UZ_TUNE): bit 2 = plain workgroup order
  // uz_wgrad_batched (one-tap, no gather): blockIdx.y = problem index; byte strides of L / R between problems, float
  // stride of the slabs
  long long lb, rb, sb;
  int nb2;            // second batch level: problem index = b * nb2 + h, L / R at + b * lb + h * lb2 (slabs: index * sb)
  long long lb2, rb2;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;
template <int V> struct IntC { static constexpr int value = V; };
constexpr unsigned OOB = 0x80000000u;
#ifndef UZ_WG_PPS
#define UZ_WG_PPS 5    // DMA pieces per sub-step slot: one K-step of the wave per slot
#endif
#ifndef UZ_WG_LATE
#define UZ_WG_LATE 2   // sub-steps by which the second wave of a SIMD trails the first with its DMA pieces
#endif

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Transposed LDS reads are issued from inline asm: for the ds_read_tr builtin hipcc (ROCm 7.2)
// inserts `s_waitcnt vmcnt(0)` before every read while LDS-DMA is in flight, which serialises the
// whole pipeline; asm reads are invisible to that pass, their lgkmcnt is counted by hand below.
template <int OFF, int OFF0 = 0>
__device__ __forceinline__ void tr_pair(bf16x4& lo, bf16x4& hi, unsigned lds_addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
               : "=&v"(lo), "=&v"(hi)
               : "v"(lds_addr), "i"(OFF0), "i"(OFF0 + OFF));
}
template <int N> __device__ __forceinline__ void wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void pin(bf16x4& v) { asm volatile("" : "+v"(v)); }

// BI x BJ channel tile, NTY kernel rows per workgroup (1 -> blockIdx.y selects the row; 3 -> all),
// 8 waves = WI x WJ x TG (TG tap groups share the NTX*NTY taps of the workgroup)
// MODE: how the x operand is addressed -- 0 as stored, 1 through nearest x2 upsampling (Wg2Args::r_up), 2 / 3 the
// 2x2 / stride-2 3x3 gathers (Wg2Args::gather 1 / 2).  Compile-time: the pieces are issued from ten places of the
// unrolled double-step, each would carry the three-way branch.
template <int BI, int BJ, int NTY, int NTX, int WI, int WJ, int TG, int MODE>
__device__ __forceinline__ void wgrad3x3_body(const Wg2Args& a, const int blk_x, const int blk_y, const int blk_z, const int grd_x,
                                              const int grd_z) {
  static_assert(WI * WJ * TG == 8, "8 waves");
  static_assert((NTX == 3) || (NTX == 1 && NTY == 1), "1 tap or 3/9 taps");
  constexpr int HALO = (NTX == 3) ? 1 : 0;
  constexpr int RBL = BI * 2, RBR = BJ * 2;             // bytes per pixel row
  constexpr int CPRL = RBL / 16, CPRR = RBR / 16;       // 16-byte chunks per row
  constexpr int RPPL = 1024 / RBL, RPPR = 1024 / RBR;   // pixel rows per 1 KiB DMA piece
  constexpr int NLP = 64 / RPPL;                        // L pieces per stage
  // R tile capacity in pixels (128 x 64 tile: strips of <= 32 columns; 64 x 128: 4 x 16 strips only)
  constexpr int RPX = (NTY == 1) ? 72 : (BI == 128 ? 136 : (BJ == 128 ? 108 : 200));
  constexpr int NRP = RPX / RPPR;
  constexpr int PPW = 5;                                // pieces per wave per stage (40 slots)
  static_assert(NLP + NRP <= 8 * PPW && NLP % 8 == 0, "stage layout");
  constexpr int STAGE = 8 * PPW * 1024;
  constexpr int WTI = BI / WI, WTJ = BJ / WJ, TI = WTI / 32, TJ = WTJ / 32;
  constexpr int NTAP = NTX * NTY;               // taps of the workgroup
  constexpr int NTW = (NTAP + TG - 1) / TG;     // taps per wave (at most)
  static_assert(TJ == 1, "one 32-channel R tile per wave");
  __shared__ __attribute__((aligned(16))) char smem[4 * STAGE];  // all 160 KiB: two slots of two K-steps

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Nine taps over four tap groups (the 64 x 64 tile): 3 + 2 + 2 + 2, and the two waves of SIMD s (waves s, s + 4)
  // take groups {0, 1} or {2, 3}: 5 / 5 / 4 / 4 tap units per SIMD.  (Three taps each for groups 0-2 and an idle
  // fourth group put 6 / 6 / 3 / 3 on the SIMDs: the matrix pipes of two of them ran half empty.)
  constexpr bool UNEVEN = (NTX * NTY == 9 && TG == 4 && WI * WJ == 2);
  const int tg = UNEVEN ? ((wave >> 1) & 1) * 2 + (wave >> 2) : wave / (WI * WJ);
  const int wij = UNEVEN ? (wave & 1) : wave % (WI * WJ);
  const int wi = wij / WJ, wj = wij % WJ;
  const int l31 = lane & 31, lh = lane >> 5;
  const int tap0 = UNEVEN ? (tg == 0 ? 0 : 1 + 2 * tg) : tg * NTW;                        // first tap of this wave
  const int ntw_me = UNEVEN ? (tg == 0 ? 3 : 2) : (NTAP - tap0 < NTW ? NTAP - tap0 : NTW);   // and how many (wave-uniform)
  // work item = (channel tile, kernel row ty, pixel split z).  The three kernel rows of one
  // (tile, z) read the same dy tile and overlapping x rows: give them workgroup ids that differ by
  // multiples of 8 so that they share an XCD (one L2) and run at about the same time (speed only).
  int bx = blk_x, by = blk_y, bz = blk_z;
  if (NTY == 1 && NTX == 3) {
    const int U = grd_x * grd_z;           // (tile, z) pairs
    if ((U & 7) == 0 && !(UZ_KFLAGS(a) & 4)) {
      const int id = blk_x + grd_x * (blk_y + 3 * blk_z);
      const int xcd = id & 7, j = id >> 3;
      by = j % 3;
      const int u = (j / 3) * 8 + xcd;
      bx = u % grd_x;
      bz = u / grd_x;
    }
  }
  const int ti0 = (bx / a.tiles_j) * BI, tj0 = (bx % a.tiles_j) * BJ;
  const int ty_blk = (NTY == 1 && NTX == 3) ? by : 0;
  const int ty_blk_g = (NTX == 1 && a.gather != 0) ? by : 0;   // gather mode: the tap of this workgroup
  const long long pb = (NTX == 1 && a.gather == 0) ? by : 0;   // batched one-tap problems: the problem of this workgroup
  const int u_beg = bz * a.upb;
  const int u_end = (u_beg + a.upb < a.units) ? u_beg + a.upb : a.units;
  const int nu = u_end - u_beg;
  const long long pb1 = (int)pb / a.nb2, pb2 = (int)pb - (int)pb1 * a.nb2;
  const __amdgpu_buffer_rsrc_t lr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(static_cast<const char*>(a.L)) + pb1 * a.lb + pb2 * a.lb2, 0, a.lbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(static_cast<const char*>(a.R)) + pb1 * a.rb + pb2 * a.rb2, 0, a.rbytes, 0x00020000);

  const int KW = a.KW, PWR = KW + 2 * HALO;        // R tile row length in pixels
  const int RR = a.KR + ((NTY == 3) ? 2 : 0);      // R tile rows
  const int roff = (NTX == 1) ? 0 : ((NTY == 3) ? -1 : ty_blk - 1);  // image row of R tile row 0 - h

  // ---- per-lane constants of this wave's five DMA pieces ---------------------------------------
  int p_rrel[PPW], p_crel[PPW], p_delta[PPW], p_coff[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int q = wave + 8 * i;
    if (q < NLP) {
      const int k = q * RPPL + lane / CPRL, pc = lane % CPRL;
      const int r = k >> a.kw_log2, c = k & (KW - 1);
      const int sw = (CPRL == 16) ? (k & 3) : ((k >> 1) & 1);
      const int lchunk = (((pc >> 2) ^ sw) << 2) + (pc & 3);   // logical 16-byte chunk = 8 channels
      p_rrel[i] = (ti0 + lchunk * 8 < a.Ci) ? r : -(1 << 20);  // channel tail -> zero fill
      p_crel[i] = c;
      p_delta[i] = r * a.W + c;
      p_coff[i] = (ti0 * 2) + lchunk * 16;
    } else if (q < NLP + NRP) {
      const int t = (q - NLP) * RPPR + lane / CPRR, pc = lane % CPRR;
      const int trow = t / PWR, tcol = t - trow * PWR;
      const int sw = (CPRR == 16) ? (t & 3) : ((t >> 1) & 1);
      const int lchunk = (((pc >> 2) ^ sw) << 2) + (pc & 3);
      p_rrel[i] = (trow < RR && tj0 + lchunk * 8 < a.Cj) ? trow + roff : -(1 << 20);
      p_crel[i] = tcol - HALO;
      p_delta[i] = (trow + roff) * a.W + tcol - HALO;
      p_coff[i] = (tj0 * 2) + lchunk * 16;
    } else {
      p_rrel[i] = -(1 << 20);
      p_crel[i] = 0;
      p_delta[i] = 0;
      p_coff[i] = 0;
    }
  }

  const int upi = (a.H * a.W) >> 6;  // K-steps per image
  const int upr = a.W / KW;          // K-steps per row block (1 unless W > 64)
  // position of K-step u: image, first row, first column
  auto locate = [&](int u, int& img, int& h, int& w0) __attribute__((always_inline)) {
    img = u / upi;
    const int rem = u - img * upi;
    const int rb = rem / upr;
    h = rb * a.KR;
    w0 = (rem - rb * upr) * KW;
  };
  // DMA piece I (of this wave's PPW) of the K-step at (image IMG, row H0, column W0) into STAGE.  Whether a piece
  // belongs to the dy or to the x tile is decided at compile time: written as a run-time `q < NLP` branch, hipcc
  // (ROCm 7.2) merges the two buffer_load ... lds calls and picks the descriptor by a computed offset into the lambda
  // capture block, which then -- with the kernel arguments and every per-lane table behind it -- stays in scratch.
#define UZ_WG_ISSUE_PIECE(I, STAGE_, IMG, H0, W0)                                                                   \
  do {                                                                                                              \
    constexpr int i_ = (I);                                                                                         \
    const int img_ = (IMG), h_ = (H0), w0_ = (W0);                                                                  \
    char* sbase_ = smem + (STAGE_) * STAGE;                                                                         \
    const int q_ = wave + 8 * i_;                                                                                   \
    const bool ok_ = (unsigned)(h_ + p_rrel[i_]) < (unsigned)a.H && (unsigned)(w0_ + p_crel[i_]) < (unsigned)a.W;   \
    const unsigned pix_ = (unsigned)((img_ * a.H + h_) * a.W + w0_ + p_delta[i_]);                                  \
    if constexpr (8 * i_ < NLP) { /* NLP % 8 == 0: a piece index is all-L or all-R for every wave */                \
      const unsigned off_ = ok_ ? pix_ * (unsigned)(a.ldl * 2) + p_coff[i_] : OOB;                                  \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(lr, (lds_ptr_t)(sbase_ + q_ * 1024), 16, off_, 0, 0, 0);            \
    } else {                                                                                                        \
      unsigned rp_ = pix_;                                                                                          \
      bool r_in_ = true;                                                                                            \
      if constexpr (MODE == 1) {                                                                                           \
        const int hh = h_ + p_rrel[i_], ww = w0_ + p_crel[i_];                                                      \
        rp_ = (unsigned)((img_ * (a.H >> 1) + (hh >> 1)) * (a.W >> 1) + (ww >> 1));                                 \
      } else if constexpr (MODE == 2) {                                                                             \
        const int hh = h_ + p_rrel[i_], ww = w0_ + p_crel[i_];                                                      \
        rp_ = (unsigned)((img_ * a.Hr + 2 * hh + (ty_blk_g >> 1)) * a.Wr + 2 * ww + (ty_blk_g & 1));                \
      } else if constexpr (MODE == 3) {                                                                             \
        const int gy = (ty_blk_g * 11) >> 5, gx = ty_blk_g - 3 * gy;                                                \
        const int rh = 2 * (h_ + p_rrel[i_]) + gy - 1, rw = 2 * (w0_ + p_crel[i_]) + gx - 1;                        \
        r_in_ = (unsigned)rh < (unsigned)a.Hr && (unsigned)rw < (unsigned)a.Wr;                                     \
        rp_ = (unsigned)((img_ * a.Hr + rh) * a.Wr + rw);                                                           \
      } else if constexpr (MODE == 4) {                                                                             \
        const int gy = (ty_blk_g * 11) >> 5, gx = ty_blk_g - 3 * gy;                                                \
        const int rh = h_ + p_rrel[i_] + (gy - 1) * a.dil, rw = w0_ + p_crel[i_] + (gx - 1) * a.dil;                \
        r_in_ = (unsigned)rh < (unsigned)a.H && (unsigned)rw < (unsigned)a.W;                                       \
        rp_ = (unsigned)((img_ * a.H + rh) * a.W + rw);                                                             \
      }                                                                                                             \
      const unsigned off_ = (ok_ && r_in_) ? rp_ * (unsigned)(a.ldr * 2) + p_coff[i_] : OOB;                        \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rr, (lds_ptr_t)(sbase_ + q_ * 1024), 16, off_, 0, 0, 0);            \
    }                                                                                                               \
  } while (0)
  auto issue = [&](int stage, int u) {
    int img, h, w0;
    locate(u, img, h, w0);
    static_assert(PPW == 5, "explicit piece list");
    UZ_WG_ISSUE_PIECE(0, stage, img, h, w0);
    UZ_WG_ISSUE_PIECE(1, stage, img, h, w0);
    UZ_WG_ISSUE_PIECE(2, stage, img, h, w0);
    UZ_WG_ISSUE_PIECE(3, stage, img, h, w0);
    UZ_WG_ISSUE_PIECE(4, stage, img, h, w0);
  };

  f32x16 acc[TI][NTW];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;

  // transposed-read lane roles: 16-lane group g, pixel row q4 and column quad p4 inside the 4x16 block
  const int g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int lk = 8 * (g >> 1) + q4;           // pixel row of this lane inside a 16-pixel sub-step
  const int lcol = 16 * (g & 1) + 4 * p4;     // channel inside a 32-channel MFMA tile
  typedef __attribute__((address_space(3))) char* lds_char_ptr;
  const unsigned smem_u = (unsigned)(size_t)(lds_char_ptr)smem;

  // byte address of this lane's transposed-read block: pixel row `row`, channel `col`
  auto tr_addr = [&](unsigned tile, int row, int col, int rowbytes, bool four_granules) -> unsigned {
    const int sw = four_granules ? (row & 3) : ((row >> 1) & 1);
    return tile + row * rowbytes + ((((col >> 5) ^ sw)) << 6) + (col & 31) * 2;
  };
  // (row offset, column shift) of the wave's tt-th tap inside the R tile
  auto tap_shift = [&](int tt, int r, int c0) -> int {
    const int tap = (tap0 + tt < NTAP) ? tap0 + tt : NTAP - 1;   // (slots past the wave's last tap read a valid address, unused)
    const int tyo = (NTY == 3) ? (tap * 11) >> 5 : 0;          // tap / 3 for tap in 0..8
    const int tx = (NTX == 3) ? tap - 3 * ((tap * 11) >> 5) : 0;
    return (r + tyo) * PWR + c0 + tx;
  };

  // Read addresses do not depend on the K-step: the dy block of sub-step ks is a constant 16 * RBL bytes further on
  // (immediate offset), and the x block of (sub-step, tap) sits at a per-lane offset worked out once here -- the
  // swizzle key is the pixel row, which moves with the tap shift and the runtime row length.  That leaves one
  // v_add per transposed-read pair in the loop (it was 22 VALU instructions per sub-step, beside 6 MFMAs).
  unsigned aoff[TI], boff[4][NTW];
#pragma unroll
  for (int i = 0; i < TI; ++i) aoff[i] = tr_addr(0u, lk, wi * WTI + i * 32 + lcol, RBL, CPRL == 16);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int k0 = ks * 16;
    const int r = k0 >> a.kw_log2, c0 = k0 & (KW - 1);
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt)
      boff[ks][tt] = NLP * 1024 + tr_addr(0u, tap_shift(tt, r, c0) + lk, wj * WTJ + lcol, RBR, CPRR == 16);
  }

  // The fragments of sub-step k + 1 (its dy pair(s) and its first x pair) are requested before the last MFMAs of
  // sub-step k, across the two K-steps of a double-step: a wave meets a bare LDS latency once per double-step, not
  // once per sub-step.  Register sets alternate with the sub-step parity.
  constexpr int NB = NTW == 1 ? 2 : NTW;   // x-fragment registers: pair (ks, tt) lives in slot (ks * NTW + tt) % NB
  bf16x4 alo[2][TI], ahi[2][TI], blo[NB], bhi[NB];
  auto fetch_first = [&](auto KS, unsigned sL) __attribute__((always_inline)) {
    constexpr int ks = decltype(KS)::value;
#pragma unroll
    for (int i = 0; i < TI; ++i) tr_pair<4 * RBL, ks * 16 * RBL>(alo[ks & 1][i], ahi[ks & 1][i], sL + aoff[i]);
    tr_pair<4 * RBR>(blo[(ks * NTW) % NB], bhi[(ks * NTW) % NB], sL + boff[ks][0]);
  };
  // The 2 x PPW DMA pieces a wave owes to the NEXT double-step go out UZ_WG_PPS at a time, slot k before sub-step k
  // of the current one (pieces 0 .. PPW - 1: K-step A, the rest: K-step B).  The second wave of a SIMD (waves 4-7)
  // trails the first by UZ_WG_LATE sub-steps, so the two never sit in an issue burst together.  Measured on UNet
  // layer shapes (B = 16, same box): two pieces per slot 96.5 us on 256 -> 256 @ 64 x 64, five per slot 87-92 us
  // whatever the lag (lag 2: 89.0, lag 0: 91.7, all ten at once + lag 4: 89.7) -- the pieces must leave EARLY, the
  // double-step waits on their arrival, not on their issue slots.
  int imgA = 0, hA = 0, wA = 0, imgB = 0, hB = 0, wB = 0;
  bool haveA = false, haveB = false, late = false;
  int nxt_stage = 0;
#define UZ_WG_PIECE(J)                                                     \
  if constexpr ((J) >= 0 && (J) < PPW) {                                   \
    if (haveA) UZ_WG_ISSUE_PIECE((J), nxt_stage, imgA, hA, wA);                         \
  } else if constexpr ((J) >= PPW && (J) < 2 * PPW) {                      \
    if (haveB) UZ_WG_ISSUE_PIECE((J) - PPW, nxt_stage + 1, imgB, hB, wB);                \
  }
  auto dma_slot = [&](auto N) __attribute__((always_inline)) {   // N: sub-step of the double-step about to start (0 .. 7)
    constexpr int n = decltype(N)::value;
#define UZ_WG_SLOT(K)                                                                \
    UZ_WG_PIECE(UZ_WG_PPS * (K))                                                       \
    if constexpr (UZ_WG_PPS > 1) { UZ_WG_PIECE(UZ_WG_PPS * (K) + 1) }                  \
    if constexpr (UZ_WG_PPS > 2) { UZ_WG_PIECE(UZ_WG_PPS * (K) + 2) }                  \
    if constexpr (UZ_WG_PPS > 3) { UZ_WG_PIECE(UZ_WG_PPS * (K) + 3) }                  \
    if constexpr (UZ_WG_PPS > 4) { UZ_WG_PIECE(UZ_WG_PPS * (K) + 4) }                  \
    if constexpr (UZ_WG_PPS > 5) { UZ_WG_PIECE(UZ_WG_PPS * (K) + 5) }                  \
    if constexpr (UZ_WG_PPS > 6) { UZ_WG_PIECE(UZ_WG_PPS * (K) + 6) }                  \
    if constexpr (UZ_WG_PPS > 7) { UZ_WG_PIECE(UZ_WG_PPS * (K) + 7) }                  \
    if constexpr (UZ_WG_PPS > 8) { UZ_WG_PIECE(UZ_WG_PPS * (K) + 8) }                  \
    if constexpr (UZ_WG_PPS > 9) { UZ_WG_PIECE(UZ_WG_PPS * (K) + 9) }
    if (late) {   // wave-uniform
      UZ_WG_SLOT(n - UZ_WG_LATE)
    } else {
      UZ_WG_SLOT(n)
    }
#undef UZ_WG_SLOT
  };
#undef UZ_WG_PIECE
#undef UZ_WG_ISSUE_PIECE

  // sub-step ks of the stage at sL; `pf`: request sub-step (ks + 1) & 3 of the stage at sLn meanwhile
  auto kstep = [&](auto KS, unsigned sL, unsigned sLn, bool pf) __attribute__((always_inline)) {
    constexpr int ks = decltype(KS)::value, cur = ks & 1;
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt) {
      if (tt + 1 < NTW) {
        tr_pair<4 * RBR>(blo[(ks * NTW + tt + 1) % NB], bhi[(ks * NTW + tt + 1) % NB], sL + boff[ks][tt + 1]);
        wait_lgkm<2>();  // everything but the pair just issued has returned (LDS returns in order)
      } else if (pf) {
        fetch_first(IntC<(ks + 1) & 3>{}, sLn);
        wait_lgkm<2 * TI + 2>();
      } else {
        wait_lgkm<0>();
      }
      if (tt == 0) {
#pragma unroll
        for (int i = 0; i < TI; ++i) {
          pin(alo[cur][i]);
          pin(ahi[cur][i]);
        }
      }
      const int slot = (ks * NTW + tt) % NB;   // folds: ks, tt and NB are compile-time under the unroll
      pin(blo[slot]);
      pin(bhi[slot]);
      const bf16x8 bfr = __builtin_shufflevector(blo[slot], bhi[slot], 0, 1, 2, 3, 4, 5, 6, 7);
      if ((NTAP % TG == 0 && !UNEVEN) || tt < ntw_me) {  // wave-uniform
#pragma unroll
        for (int i = 0; i < TI; ++i) {
          const bf16x8 afr = __builtin_shufflevector(alo[cur][i], ahi[cur][i], 0, 1, 2, 3, 4, 5, 6, 7);
          acc[i][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr, acc[i][tt], 0, 0, 0);
        }
      }
    }
  };
  // the four sub-steps of one K-step (HALF 0 / 1 of the double-step); `more`: another K-step (stage at sLn)
  // follows without a barrier
  auto compute = [&](auto HALF, unsigned sL, unsigned sLn, bool more, bool mm) __attribute__((always_inline)) {
    constexpr int n0 = decltype(HALF)::value * 4;
    dma_slot(IntC<n0 + 0>{});
    if (mm) kstep(IntC<0>{}, sL, sL, true);
    dma_slot(IntC<n0 + 1>{});
    if (mm) kstep(IntC<1>{}, sL, sL, true);
    dma_slot(IntC<n0 + 2>{});
    if (mm) kstep(IntC<2>{}, sL, sL, true);
    dma_slot(IntC<n0 + 3>{});
    if (mm) kstep(IntC<3>{}, sL, sLn, more);
  };

  if (nu > 0) issue(0, u_beg);
  if (nu > 1) issue(1, u_beg + 1);
  {
    // two K-steps (2 x 64 pixels) per barrier: the pair for double-step d + 1 streams in while d computes
    const int nd = (nu + 1) >> 1;
    late = wave >= 4 && !(UZ_KFLAGS(a) & 16);
    const bool dma = !(UZ_KFLAGS(a) & 32), mm = !(UZ_KFLAGS(a) & 64);   // measurement only: one half of the loop
#pragma unroll 1
    for (int d = 0; d < nd; ++d) {
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      nxt_stage = ((d + 1) & 1) * 2;
      const unsigned s0 = smem_u + (d & 1) * 2 * STAGE, s1 = s0 + STAGE;
      const bool two = 2 * d + 1 < nu;
      haveA = dma && 2 * d + 2 < nu;
      haveB = dma && 2 * d + 3 < nu;
      if (haveA) locate(u_beg + 2 * d + 2, imgA, hA, wA);
      if (haveB) locate(u_beg + 2 * d + 3, imgB, hB, wB);
      if (mm) fetch_first(IntC<0>{}, s0);
      compute(IntC<0>{}, s0, s1, two, mm);
      compute(IntC<1>{}, s1, s1, false, mm && two);
    }
  }

  // ---- partial slab [split][tap][Ci][Cj] ----------------------------------------------------------
#pragma unroll
  for (int tt = 0; tt < NTW; ++tt) {
    const int tw = tap0 + tt;
    if (tt >= ntw_me) continue;  // wave-uniform
    const int tap = (NTX == 1) ? ty_blk_g : ((NTY == 3) ? tw : ty_blk * 3 + tw);
    float* slab = a.slab + pb * a.sb + ((size_t)bz * (NTX == 1 ? (a.gather == 1 ? 4 : (a.gather >= 2 ? 9 : 1)) : 9) + tap) * (size_t)a.Ci * a.Cj;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      const int cj = tj0 + wj * WTJ + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ti0 + wi * WTI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (ci < a.Ci && cj < a.Cj) slab[(size_t)ci * a.Cj + cj] = acc[i][tt][r];
      }
    }
  }
}

template <int BI, int BJ, int NTY, int NTX, int WI, int WJ, int TG, int MODE>
__global__ __launch_bounds__(512, 1) void wgrad3x3_kernel(const Wg2Args a) {
  wgrad3x3_body<BI, BJ, NTY, NTX, WI, WJ, TG, MODE>(a, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.z);
}

// Several one-tap problems (the nn.Linear weight gradients of a backward range, uz_wgrad_multi) in one launch: workgroup
// w belongs to problem i with first[i] <= w < first[i + 1] and is its (tile, split) pair number w - first[i], tile fastest.
// The problems' arguments travel in the kernel-argument segment (<= 4 KB: UZ_WG_MULTI problems per launch).
constexpr int UZ_WG_MULTI = 24;
struct WgMulti {
  int n;
  int first[UZ_WG_MULTI + 1];
  int tiles[UZ_WG_MULTI];
  Wg2Args p[UZ_WG_MULTI];
};
static_assert(sizeof(WgMulti) <= 4096, "kernel arguments");
template <int BI, int BJ, int WI, int WJ, int TG>
__global__ __launch_bounds__(512, 1) void wgrad3x3_multi_kernel(const WgMulti m) {
  int i = 0;
  {   // first[] is increasing: binary search (a linear walk is one dependent scalar load from the argument segment per problem)
    int lo = 0, hi = m.n;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if ((int)blockIdx.x >= m.first[mid]) lo = mid;
      else hi = mid;
    }
    i = lo;
  }
  const int tiles = m.tiles[i], n = m.first[i + 1] - m.first[i];
  int local = blockIdx.x - m.first[i];
  // The tiles of one pixel range read the same dy / x pixels: put them on one XCD (workgroup ids 8 apart share an L2) and
  // next to each other in time.  The problem's ids of one residue mod 8 form a column; columns are filled one after the
  // other with (pixel range, tile) pairs, tile fastest -- a permutation of the problem's own workgroups.  (The hardware's
  // XCD is blockIdx.x mod 8 = (first[i] + local) mod 8: first[i] need not be a multiple of 8 -- it rotates WHICH XCD a
  // column lands on, not the fact that equal `local & 7` means equal XCD.)
  {
    const int c = local & 7, j = local >> 3, q = n >> 3, rem = n & 7;
    local = c * q + (c < rem ? c : rem) + j;
  }
  wgrad3x3_body<BI, BJ, 1, 1, WI, WJ, TG, 0>(m.p[i], local % tiles, 0, local / tiles, tiles, 1);
}

}  // namespace

int uz_wgrad3x3_plan(const uz_wgrad_desc* d, UzWgrad2Plan* p, int batch) {
  p->v9 = 0;
  p->bi = 0;
  if (batch == 1 && uz_wgrad9_plan(d, p)) return 1;   // nine taps per workgroup, row walk (uz_wgrad9.hip)
  if (batch == 1 && uz_wgrad_g4_plan(d, p)) return 1;   // 2 x 2 gather, four taps per workgroup (uz_wgrad_g4.hip)
  const bool up = d->taps_mode == UZ_TAPS_CONV_UP2;
  const bool s2 = d->taps_mode == UZ_TAPS_CONV_S2 && d->ntaps == 9;
  // dilated 3x3 (u2net's RSU4F, dirate 2 / 4 / 8 on 32 x 32 and 16 x 16 maps): nine one-tap problems whose x pixel is
  // displaced by (ty - 1, tx - 1) * dil -- the addressing of the stride-2 gather without the stride.  (They ran on the
  // first-generation kernel: 6 x 64.7 + 9 x 35.7 us per u2net step.)
  const bool dil9 = d->taps_mode == UZ_TAPS_CONV && d->ntaps == 9 && d->dil > 1 && !(uz_tune_flags() & 0x200);
  const bool gather = (d->taps_mode == UZ_TAPS_GATHER2X2 && d->ntaps == 4 && !(uz_tune_flags() & 0x4000000)) || s2 || dil9;
  if (d->dtype != UZ_BF16 || !(d->taps_mode == UZ_TAPS_CONV || up || gather)) return 0;
  if (!((d->ntaps == 9 && d->dil == 1) || (d->ntaps == 1 && !up) || gather)) return 0;
  p->one_tap = d->ntaps == 1 || gather;   // gather: four (nine) one-tap problems, blockIdx.y = tap
  p->gather = dil9 ? 3 : (s2 ? 2 : (gather ? 1 : 0));
  if (d->Ci % 8 != 0 || d->Cj % 8 != 0) return 0;
  int W = d->W, H = d->H;
  long long nimg = d->N;
  if (p->one_tap && !gather) {
    // no neighbourhood: the pixels are one flat list of N*H*W tokens, walked as rows of 64 whatever the map's
    // shape (7 x 7, 14 x 14, 56 x 56 token maps of swin_unet_v2 at 224 x 224); the ragged last row reads
    // past the buffers' bounds and is zero-filled by the DMA
    const long long P = (long long)d->N * d->H * d->W;
    if ((P + 63) / 64 >= (1LL << 24)) return 0;
    W = 64;
    H = (int)((P + 63) / 64);
    nimg = 1;
  }
  if (!(W == 16 || W == 32 || (W >= 64 && W % 64 == 0))) return 0;
  p->kw = W < 64 ? W : 64;
  // Nine taps per workgroup on a 128 x 64 (dy x x) or 64 x 128 channel tile: the x tile (strips of 4 x 16 pixels +
  // halo) serves nine taps instead of three -- 3.2 KB of LDS-DMA per MFLOP instead of 5.2 -- at 160 accumulator
  // registers per wave.  Measured against the 128 x 128 three-tap tile / the 64 x 64 nine-tap tile (B = 16, same box):
  // 64 <-> 128 channels @ 256 x 256 201-208 -> 161-167 us; 1024 -> 512 @ 32 x 32 159 -> 146; 1024 -> 1024 @ 16 x 16
  // 93 -> 90; but 5-10 % SLOWER on the 128 ... 512-channel layers at 32 x 32 ... 128 x 128, which keep three taps.
  {
    const int f = uz_tune_flags();   // ablation build: 0x100000 / 0x400000 force a form where it applies, 0x800000 neither
    const bool ok9 = !p->one_tap && !gather && !(f & 0x800000);
    p->wide9 = 0;
    const bool many_pixels = (long long)d->N * d->H * d->W >= (1LL << 19);   // (64 -> 128 @ 128 x 128, B = 16: 59.5 -> 62.0 us)
    if (ok9 && d->Ci % 128 == 0 && d->Cj % 64 == 0 &&
        ((d->Cj == 64 && many_pixels) || (d->Cj >= 1024 && d->Ci >= 512) || (f & 0x100000)))
      p->wide9 = 1;
    else if (ok9 && d->Ci == 64 && d->Cj % 128 == 0) p->wide9 = 2;
  }
  if (p->wide9 && W >= 32) p->kw = ((uz_tune_flags() & 0x200000) && p->wide9 == 1) ? 32 : 16;   // (2 x 32 strips: no gain)
  if (!p->wide9 && !p->one_tap && W >= 64 && !(d->Ci % 128 == 0 && d->Cj % 128 == 0)) {
    // the 9-tap 64 x 64 kernel keeps all three tap rows in one R tile of (kr + 2) x (kw + 2) pixels per 64-pixel
    // unit: a 1 x 64 strip re-reads x 3.1 times, 2 x 32 2.1 times, 4 x 16 1.7 times
    // (measured on 64 -> 64 @ 256 x 256: 135.7 / 132.1 / 129.6 us -- the kernel is bound elsewhere)
    const int f = (uz_tune_flags() >> 24) & 3;   // 0: default (16), 1: 64, 2: 32, 3: 16
    p->kw = f == 1 ? 64 : f == 2 ? 32 : 16;
  }
  p->kr = 64 / p->kw;
  if (H % p->kr != 0) return 0;
  p->H = H;
  p->W = W;
  const long long lbytes = ((long long)d->N * d->H * d->W - 1) * d->ldl * 2 + (long long)d->Ci * 2;
  const long long rbytes = ((long long)d->N * d->Hr * d->Wr - 1) * d->ldr * 2 + (long long)d->Cj * 2;
  if (lbytes >= (1LL << 31) || rbytes >= (1LL << 31)) return 0;
  p->big = (d->Ci % 128 == 0 && d->Cj % 128 == 0 && !p->wide9) ? 1 : 0;
  if (p->one_tap && !p->big) {
    // one-tap problems (nn.Linear weight gradients: 288 x 96, 96 x 96 ...) are bound by re-reading the
    // operands, once per tile of the OTHER operand: take the 128 x 128 tile (channel tails are zero-filled)
    // whenever that lowers Ci * tiles_j + Cj * tiles_i
    const long long t64 = (long long)d->Ci * ((d->Cj + 63) / 64) + (long long)d->Cj * ((d->Ci + 63) / 64);
    const long long t128 = (long long)d->Ci * ((d->Cj + 127) / 128) + (long long)d->Cj * ((d->Ci + 127) / 128);
    if (t128 < t64) p->big = 1;
  }
  const int b = p->big ? 128 : 64;
  p->tiles_i = (d->Ci + (p->wide9 == 1 ? 128 : b) - 1) / (p->wide9 == 1 ? 128 : b);
  p->tiles_j = (d->Cj + (p->wide9 == 2 ? 128 : b) - 1) / (p->wide9 == 2 ? 128 : b);
  p->kg = 1;
  p->units = (int)(nimg * H * W / 64);
  if (batch > 1 && !(p->one_tap && !gather)) return 0;
  const long long base = (long long)p->tiles_i * p->tiles_j * ((p->big && !p->one_tap) ? 3 : (gather ? d->ntaps : 1)) * batch;
  // one workgroup per CU (160 KB LDS each): aim for a single full round of <= 256 workgroups
  long long split = base >= UZ_NUM_CU ? 1 : UZ_NUM_CU / base;
  long long max_split = p->units / 8 > 0 ? p->units / 8 : 1;
  if (max_split > 256) max_split = 256;
  {
    const char* e = uz_ablate_env("UZ_WG_SPLIT");   // experiment hook: cap of the pixel split of one-tap problems
    if (e && p->one_tap && atoi(e) > 0 && max_split > atoi(e)) max_split = atoi(e);
  }
  if (split > max_split) split = max_split;
  if (split < 1) split = 1;
  p->upb = (int)((p->units + split - 1) / split);
  p->split = (p->units + p->upb - 1) / p->upb;
  if (p->big && !p->one_tap) {
    // make (tiles * split) a multiple of 8 when a nearby split allows it (XCD-aware id remap)
    const long long tiles = (long long)p->tiles_i * p->tiles_j;
    for (int ds = 0; ds < 8; ++ds) {
      const long long s2 = split - ds;
      if (s2 < 1) break;
      const int upb2 = (int)((p->units + s2 - 1) / s2);
      const int sp2 = (p->units + upb2 - 1) / upb2;
      if ((tiles * sp2) % 8 == 0) {
        p->upb = upb2;
        p->split = sp2;
        break;
      }
    }
  }
  p->nslabs = p->split * p->kg;
  return 1;
}

static void wg2_fill(const uz_wgrad_desc* d, const UzWgrad2Plan& p, const void* L, const void* R, float* slab, Wg2Args* ap) {
  Wg2Args& a = *ap;
  a.lb = a.rb = a.lb2 = a.rb2 = 0;
  a.nb2 = 1;
  a.sb = (long long)p.nslabs * d->ntaps * d->Ci * d->Cj;
  a.L = L;
  a.R = R;
  a.slab = slab;
  a.lbytes = (unsigned)(((long long)d->N * d->H * d->W - 1) * d->ldl * 2 + (long long)d->Ci * 2);
  a.rbytes = (unsigned)(((long long)d->N * d->Hr * d->Wr - 1) * d->ldr * 2 + (long long)d->Cj * 2);
  a.r_up = d->taps_mode == UZ_TAPS_CONV_UP2 ? 1 : 0;
  a.gather = p.gather;
  a.dil = d->dil;
  a.Hr = d->Hr;
  a.Wr = d->Wr;
  a.flags = uz_tune_flags();
  a.N = d->N;
  a.H = p.H;
  a.W = p.W;
  a.Ci = d->Ci;
  a.ldl = d->ldl;
  a.Cj = d->Cj;
  a.ldr = d->ldr;
  a.KW = p.kw;
  a.kw_log2 = p.kw == 64 ? 6 : (p.kw == 32 ? 5 : 4);
  a.KR = p.kr;
  a.units = p.units;
  a.upb = p.upb;
  a.tiles_j = p.tiles_j;
}

// One launch for up to UZ_WG_MULTI one-tap problems of one tile shape (uz_wgrad_multi): see wgrad3x3_multi_kernel
int uz_wgrad3x3_multi_max() { return UZ_WG_MULTI; }
int uz_wgrad3x3_multi_launch(const UzWgradMultiItem* items, int n, hipStream_t s) {
  UZ_REQUIRE(n >= 1 && n <= UZ_WG_MULTI, "uz_wgrad_multi: %d problems in one launch", n);
  WgMulti m;
  m.n = n;
  int wg = 0;
  const int big = items[0].p.big;
  for (int i = 0; i < n; ++i) {
    const UzWgradMultiItem& it = items[i];
    UZ_REQUIRE(it.p.one_tap && !it.p.gather && !it.p.v9 && it.p.big == big && it.p.kg == 1,
               "uz_wgrad_multi: one-tap problems of one tile shape only");
    wg2_fill(it.d, it.p, it.L, it.R, it.slab, &m.p[i]);
    m.first[i] = wg;
    m.tiles[i] = it.p.tiles_i * it.p.tiles_j;
    wg += m.tiles[i] * it.p.split;
  }
  m.first[n] = wg;
  if (big) hipLaunchKernelGGL((wgrad3x3_multi_kernel<128, 128, 2, 4, 1>), dim3(wg), dim3(512), 0, s, m);
  else hipLaunchKernelGGL((wgrad3x3_multi_kernel<64, 64, 2, 2, 2>), dim3(wg), dim3(512), 0, s, m);
  UZ_LAUNCH_CHECK("uz_wgrad_multi");
  return UZ_OK;
}

int uz_wgrad3x3_launch(const uz_wgrad_desc* d, const UzWgrad2Plan& p, const void* L, const void* R,
                       float* slab, hipStream_t s, int batch, long long lb_bytes, long long rb_bytes, long long slab_stride,
                       int batch2, long long lb2_bytes, long long rb2_bytes, const UzXf* xf) {
  if (p.v9) {
    UZ_REQUIRE(batch == 1, "uz_wgrad(3x3): the row-walk / four-tap kernels take one problem");
    UZ_REQUIRE(xf == nullptr || p.v9 == 1, "uz_wgrad_xf: the row-walk kernel only");
    return p.v9 == 2 ? uz_wgrad_g4_launch(d, p, L, R, slab, s) : uz_wgrad9_launch(d, p, L, R, slab, s, xf);
  }
  UZ_REQUIRE(xf == nullptr, "uz_wgrad_xf: the row-walk kernel only");
  Wg2Args a;
  wg2_fill(d, p, L, R, slab, &a);
  a.lb = lb_bytes;
  a.rb = rb_bytes;
  a.nb2 = batch2 > 1 ? batch2 : 1;
  a.lb2 = lb2_bytes;
  a.rb2 = rb2_bytes;
  a.sb = slab_stride ? slab_stride : (long long)p.nslabs * d->ntaps * d->Ci * d->Cj;
  UZ_REQUIRE(batch == 1 || (p.one_tap && !p.gather && batch <= 65535), "uz_wgrad(3x3): only one-tap problems are batched");
  dim3 block(512);
  const int mode = a.r_up ? 1 : (p.gather == 1 ? 2 : (p.gather == 2 ? 3 : (p.gather == 3 ? 4 : 0)));
#define UZ_WG_LAUNCH(MODE_, ...) \
  hipLaunchKernelGGL((wgrad3x3_kernel<__VA_ARGS__, MODE_>), grid, block, 0, s, a)
  if (p.one_tap) {
    UZ_REQUIRE(mode != 1, "uz_wgrad(3x3): one-tap problems have no upsampled form");
    dim3 grid(p.tiles_i * p.tiles_j, p.gather >= 2 ? 9 : (p.gather ? 4 : batch), p.split);
    if (p.big) {
      if (mode == 0) UZ_WG_LAUNCH(0, 128, 128, 1, 1, 2, 4, 1);
      else if (mode == 2) UZ_WG_LAUNCH(2, 128, 128, 1, 1, 2, 4, 1);
      else if (mode == 3) UZ_WG_LAUNCH(3, 128, 128, 1, 1, 2, 4, 1);
      else UZ_WG_LAUNCH(4, 128, 128, 1, 1, 2, 4, 1);
    } else {   // 4 waves idle: memory-bound
      if (mode == 0) UZ_WG_LAUNCH(0, 64, 64, 1, 1, 2, 2, 2);
      else if (mode == 2) UZ_WG_LAUNCH(2, 64, 64, 1, 1, 2, 2, 2);
      else if (mode == 3) UZ_WG_LAUNCH(3, 64, 64, 1, 1, 2, 2, 2);
      else UZ_WG_LAUNCH(4, 64, 64, 1, 1, 2, 2, 2);
    }
  } else if (p.big) {
    UZ_REQUIRE(mode <= 1, "uz_wgrad(3x3): gathers are one-tap problems");
    dim3 grid(p.tiles_i * p.tiles_j, 3, p.split);
    if (mode == 0) UZ_WG_LAUNCH(0, 128, 128, 1, 3, 2, 4, 1);
    else UZ_WG_LAUNCH(1, 128, 128, 1, 3, 2, 4, 1);
  } else {
    UZ_REQUIRE(mode <= 1, "uz_wgrad(3x3): gathers are one-tap problems");
    dim3 grid(p.tiles_i * p.tiles_j, 1, p.split);
    // The kernel is bound by its transposed LDS reads (a 32 x 32 wave tile reads 1.2 KB per MFMA, every
    // B fragment used once): 64-row wave tiles share each B fragment between two MFMAs (0.83 KB per MFMA) even
    // though 4 tap groups leave 3 of 12 tap slots empty.  64->64 @256x256: 141.9 -> 129.6 us, 128->64: 229.8 -> 213.4.
    if (p.wide9 == 1) {
      if (mode == 0) UZ_WG_LAUNCH(0, 128, 64, 3, 3, 2, 2, 2);
      else UZ_WG_LAUNCH(1, 128, 64, 3, 3, 2, 2, 2);
    } else if (p.wide9 == 2) {
      if (mode == 0) UZ_WG_LAUNCH(0, 64, 128, 3, 3, 1, 4, 2);
      else UZ_WG_LAUNCH(1, 64, 128, 3, 3, 1, 4, 2);
    } else if (mode == 0) UZ_WG_LAUNCH(0, 64, 64, 3, 3, 1, 2, 4);
    else UZ_WG_LAUNCH(1, 64, 64, 3, 3, 1, 2, 4);
  }
#undef UZ_WG_LAUNCH
  UZ_LAUNCH_CHECK("uz_wgrad(3x3)");
  return UZ_OK;
}
