// Weight gradient of a 3x3 convolution (stride 1, dilation 1), bf16, gfx950 (MI355X) -- "row walk", round 4:
//     dW[co][ci][ty][tx] = sum_p dy[p][co] * x[p + (ty-1, tx-1)][ci]
// (autograd weight gradient of nn.Conv2d k3 p1: reference unet_zoo/models/common_layers.py:28,31,47,52, entered from
// loss.backward(), unet_zoo/utils/training_loop.py:119).
//
// Why a second kernel beside uz_wgrad3x3.hip.  That kernel's 128 x 128 tile owns ONE kernel row: every pixel of dy and
// of x enters LDS once per tap row (174 MB read per launch against 70-130 MB algorithmic, profiles/r03_pmc_traffic.json),
// its ring holds one double-step (<= 1.5 us) of prefetch, and a 64 x 32 wave tile re-reads the x fragment of every tap.
// Here a workgroup owns ALL NINE taps of a (BI dy-channels) x (64 x-channels) tile and WALKS DOWN the image:
//   * a step = 64 output pixels = G rows x KW columns of one column strip (KW = min(W, 64), G = 64 / KW).  The x rows
//     live in a ring of row slots ((KW + 2) pixels + pad, 64 channels): step s needs rows h-1 .. h+G, of which only the
//     G new ones are fetched -- every x pixel enters LDS once (plus the strip's two halo columns), every dy pixel once;
//     LDS fill per step 24.4 KB for 9.4 MFLOP (128 x 64 tile) instead of 34 KB for 6.3 MFLOP.
//   * dy tiles and x rows are requested NSL - 2 steps (2-4 us of MFMA time) ahead by LDS-DMA; ONE s_barrier per step, in
//     the MIDDLE of the step: it publishes the next step's tiles and frees the previous step's slots, so the fragment
//     reads run across step boundaries without a bubble.
//   * a wave owns a 32 x 32 channel tile x 9 taps (144 accumulator registers).  The three kernel columns of a tap row
//     come out of ONE pair of transposed reads plus one pair shifted by two pixels: tx = 0 and tx = 2 are the two pairs as
//     read, tx = 1 is four v_alignbit_b32 between them (a lane holds 8 consecutive pixels of its channel) -- 4 x-fragment
//     reads per 3 MFMAs instead of 6.
//   * workgroups that share a pixel range (all channel tiles of one split) get ids that are equal mod 8: one XCD, one
//     L2 -- the tiles' common dy / x bytes come from HBM once (speed only).
// Output: one fp32 slab [split][tap][Ci][Cj] per pixel split, summed in fixed order by uz_wgrad's reduce kernel
// (bitwise reproducible).
#include "uz_common.h"

// The in-kernel measurement switches (no DMA / no MFMAs / no slab / no fragment reads: profiles/r04_wgrad9_skeleton.txt)
// exist only with -DUZ_W9_SKEL: each costs a scalar branch per unit of the main loop, which the plain ablation build
// (make ABLATE=1, used for same-box A/B of the PLANS) must not carry either.
#ifndef UZ_W9X_SKEL
#define UZ_W9X_SKEL 0   // measurement builds of the XF form: 1 no arithmetic, 2 no LDS reads / writes either (the walk only)
#endif
#ifdef UZ_W9_SKEL
#define UZ_W9_FLAGS(a) ((a).flags)
#else
#define UZ_W9_FLAGS(a) 0
#endif

namespace {

struct Wg9Args {
  const void* L;
  const void* R;
  float* slab;
  unsigned lbytes, rbytes;
  int H, W, Ci, ldl, Cj, ldr;
  int Hr, Wr;            // pixel grid of R (nearest x2 upsampling: H / 2, W / 2)
  int nstrips, hsteps;   // column strips per image (W / KW), steps per (image, strip) (H / G)
  int units, upb;        // steps in the whole tensor, steps per pixel split
  int tiles_j, ntiles, split;
  int flags;
  // XF (template parameter): R holds the RAW output of the convolution in front of this layer; the loader waves turn every
  // x row into relu(fma(R, xf_scale[c], xf_shift[c])) (rounded to bf16 as uz_bn_relu_apply stores it) inside the LDS ring
  // before they publish it -- the weight gradient of a DoubleConv's second convolution without the normalised tensor
  const float* xf_scale;
  const float* xf_shift;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(3))) char* lds_char_ptr;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int V> struct IntC { static constexpr int value = V; };
constexpr unsigned OOB = 0x80000000u;   // beyond any descriptor's num_records (tensors < 2 GiB)

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt range");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
// (transposed reads from inline asm: for the builtin hipcc inserts s_waitcnt vmcnt(0) while LDS-DMA is in flight)
template <int OFF, int OFF0>
__device__ __forceinline__ void tr_pair(bf16x4& lo, bf16x4& hi, unsigned lds_addr) {
  static_assert(OFF0 >= 0 && OFF0 + OFF < 65536, "ds offset");
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
               : "=&v"(lo), "=&v"(hi)
               : "v"(lds_addr), "i"(OFF0), "i"(OFF0 + OFF));
}
__device__ __forceinline__ void pin(bf16x4& v) { asm volatile("" : "+v"(v)); }
// (a function: with a vector element as the direct operand of __builtin_bit_cast hipcc 7.2 reads element 0, uz_conv3x3_pp.hip)
__device__ __forceinline__ float u2f(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ void lds_rd16(u32x4& dst, unsigned lds_addr) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(lds_addr));
}
__device__ __forceinline__ void lds_wr16(unsigned lds_addr, const u32x4& v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr), "v"(v) : "memory");
}

template <int BI, int KW, int NW_ = 4, int KG_ = 1> struct W9 {
  static_assert((BI == 128 || BI == 64) && (KW == 64 || KW == 32 || KW == 16), "tile configuration");
  static constexpr int NW = NW_;                     // waves: KG pixel groups x NW / (2 KG) (dy) x 2 (x)
  static constexpr int KG = KG_;                     // pixel groups: group k owns the sub-steps 4 k / KG .. of every step
  static constexpr int TI = BI * KG / (16 * NW);     // 32-channel dy tiles per wave
  static constexpr int NU = 12 / KG;                 // units (sub-step, tap row) per wave and step
  static_assert(TI >= 1 && (NW == 4 || NW == 8) && (KG == 1 || KG == 2) && TI * 16 * NW == BI * KG, "wave grid");
  static constexpr int G = 64 / KW;                  // image rows per step
  static constexpr int RBL = BI * 2;                 // bytes per dy pixel row in LDS
  static constexpr int CPRL = RBL / 16;              // 16-byte chunks per dy pixel
  static constexpr int RPPL = 1024 / RBL;            // dy pixels per 1 KB DMA piece
  static constexpr int NLP = 64 / RPPL;              // dy pieces per step
  static constexpr int RPX = KW == 64 ? 72 : (KW == 32 ? 40 : 24);   // pixels of an x row slot (KW + 2, padded to 8)
  static constexpr int RPIECES = RPX / 8;            // 1 KB pieces per x row (128 B per pixel)
  static constexpr int QR = (KW + 1) / 8;            // the piece that holds the right halo column
  static constexpr int ROWB = RPX * 128;
  static constexpr int LSTAGE = 64 * RBL;
#ifndef UZ_W9_DEEP
#define UZ_W9_DEEP 0   // 1: a seventh dy stage on the 64-wide tile -- measured same-box against six: no difference (the ring is not the limit)
#endif
  static constexpr int NSL = BI == 128 ? (KW == 64 ? 4 : 5) : (KW == 16 ? 5 : (UZ_W9_DEEP ? 7 : 6));   // dy stages = steps in the ring
  static constexpr int NSR = NSL * G + 4;            // x row slots: G per step + 2 shared + 2 for one segment start
  static constexpr int KL = NLP / NW;                // dy pieces per wave and step
  static constexpr int MR = (RPIECES + NW - 1) / NW; // x pieces per wave and row (piece q = wave + 4 m; the last m: some waves)
  static constexpr int PMIN = KL + G * (RPIECES / NW);   // pieces EVERY wave requests per steady step
  static constexpr int L_OFF = 0, R_OFF = NSL * LSTAGE, RING = R_OFF + NSR * ROWB;
  static constexpr int XCH = KG == 2 ? (NW / 2) * 9 * 16 * 256 : 0;   // the pixel groups' accumulators meet in LDS after the loop
  static constexpr int SMEM = RING > XCH ? RING : XCH;
  static_assert(NLP % NW == 0, "dy pieces split evenly over the waves");
  static_assert(SMEM <= 160 * 1024, "LDS budget");
  static_assert((NSL - 2) * PMIN <= 63, "vmcnt range");
  static_assert(RPX >= KW + 2 && QR < RPIECES, "row slot");
};

// BI dy-channels x 64 x-channels x nine taps per workgroup; KW-wide column strips; MODE 1: x is read through nearest x2
// upsampling (UpConvBlock, attention_unet.py:16-29).
// Four waves, one per SIMD, each a (BI / 2) x 32 channel tile x 9 taps (288 / 144 accumulator registers).  The first
// version ran 8 waves of 32 x 32 tiles with the ring bookkeeping recomputed per DMA piece: ~400 scalar / vector / LDS
// instructions per wave and step beside 36 MFMAs.  An instruction costs its wave >= 4 cycles of issue, so the MFMAs
// alone (everything else switched off) ran at 54 cycles each and the empty loop took as long as the MFMAs should
// (profiles/r04_wgrad9_skeleton.txt).  Now: ~240 instructions beside 72 MFMAs (3.3 per MFMA gap), bookkeeping per step
// and row instead of per piece, fragment addresses per step instead of per read.
// LD = 1: four LOADER waves (4 .. 7) beside four compute waves (one of each per SIMD).  The loaders do nothing but the
// ring: bookkeeping, LDS-DMA requests, the vmcnt wait that publishes a step; the compute waves' instruction stream is then
// fragment reads, permutes and MFMAs only (~3.7 instructions per MFMA gap, which one wave per SIMD hides, cf.
// MI355X_MICROARCH.md "one wave per SIMD ... <= 5 ... hidden per gap").  Both kinds meet at the one s_barrier per step.
template <int BI, int KW, int MODE, int NWV = 4, int KGV = 1, int LD = 0, bool XF = false>
__global__ __launch_bounds__(64 * (NWV + 4 * LD), 1) void wgrad9_kernel(const Wg9Args a) {
  typedef W9<BI, KW, NWV, KGV> C;
  static_assert(LD == 0 || (NWV == 4 && KGV == 1), "loader waves go with four compute waves");
  static_assert(!XF || LD == 1, "the input transform is the loader waves' work");
  constexpr int KG = C::KG, NU = C::NU;
  constexpr int NW = C::NW, TI = C::TI, G = C::G, RBL = C::RBL, CPRL = C::CPRL, RPPL = C::RPPL, NLP = C::NLP;
  constexpr int RPIECES = C::RPIECES, QR = C::QR, ROWB = C::ROWB, LSTAGE = C::LSTAGE, NSL = C::NSL, NSR = C::NSR;
  constexpr int KL = C::KL, MR = C::MR, PMIN = C::PMIN, L_OFF = C::L_OFF, R_OFF = C::R_OFF;
  constexpr unsigned ROW_OOB = 0x80000000u;    // a whole row / step out of range: every lane beyond num_records
  constexpr unsigned LANE_OOB = 0x60000000u;   // a halo lane at an image edge: out of range with or without ROW_OOB on top
                                               // (tensors < 0x60000000 bytes, uz_wgrad9_plan)
  __shared__ __attribute__((aligned(1024))) char smem[C::SMEM];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = LD != 0 && wave_id >= 4;
  const int wave = LD != 0 ? (wave_id & 3) : wave_id;   // index among the waves of its kind (DMA piece tables / wave tile)
  // wave tile: dy channels 32 TI wi .., x channels 32 wj ..; pixel group kgrp (KG = 2: the 64 x 64 tile on eight waves --
  // waves w and w + 4, which share a SIMD, own the same channel tile and the two halves of every step's pixels; with four
  // waves, one per SIMD, the ~200 scalar / vector / LDS instructions of a step beside 36 MFMAs left the matrix pipe idle a
  // third of the time)
  const int kgrp = KG == 2 ? wave / (NW / 2) : 0, wloc = KG == 2 ? wave % (NW / 2) : wave;
  const int wi = wloc >> 1, wj = wloc & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  // work item = (channel tile, pixel split z); the tiles of one z share an XCD when the split is a multiple of 8
  int tile, z;
  {
    const int id = blockIdx.x;
    if ((a.split & 7) == 0 && !(UZ_W9_FLAGS(a) & 4)) {
      const int k = id >> 3, zh = k / a.ntiles;
      tile = k - zh * a.ntiles;
      z = zh * 8 + (id & 7);
    } else {
      z = id / a.ntiles;
      tile = id - z * a.ntiles;
    }
  }
  const int ti0 = (tile / a.tiles_j) * BI, tj0 = (tile % a.tiles_j) * 64;
  const int u_beg = z * a.upb;
  const int nu = (u_beg + a.upb < a.units ? u_beg + a.upb : a.units) - u_beg;
  const __amdgpu_buffer_rsrc_t lr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.L), 0, a.lbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.R), 0, a.rbytes, 0x00020000);
  const unsigned smem_u = (unsigned)(size_t)(lds_char_ptr)smem;

  // ---- DMA piece tables: per lane, the byte offset of its 16 bytes relative to the step's first dy pixel / to the x
  // pixel at the strip's first column.  64-byte granules of a pixel row are XOR-swizzled with the pixel index, on this
  // side and on the read side, so that the four pixel rows of a transposed read fall into distinct banks.
  // Lanes that are never valid (channel tails) carry ROW_OOB: out of range on top of a valid base; on top of an
  // out-of-range base they wrap and fetch bytes that only reach accumulator rows / columns nobody stores.
  unsigned ltab[KL];          // dy piece wave + 4 k
  unsigned xtab[MR], xtabe[MR];   // x piece q = wave + 4 m of a row; xtabe: its halo lane out of range (strip at the image edge)
  int xflag[MR];                  // XF: bit 0 this lane's bytes exist at all (channel tail, slot padding), bit 1 a halo column
#pragma unroll
  for (int k = 0; k < KL; ++k) {
    const int i = wave + NW * k;
    const int kpx = i * RPPL + lane / CPRL, pc = lane % CPRL;
    const int r = kpx / KW, c = kpx % KW;
    const int sw = CPRL == 16 ? (kpx & 3) : ((kpx >> 1) & 1);
    const int lchunk = (((pc >> 2) ^ sw) << 2) + (pc & 3);
    ltab[k] = (ti0 + lchunk * 8 < a.Ci) ? (unsigned)((r * a.W + c) * a.ldl * 2 + ti0 * 2 + lchunk * 16) : ROW_OOB;
  }
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    const int q = wave + NW * m;
    const int t = q * 8 + (lane >> 3), pc = lane & 7;   // slot pixel t <-> image column w0 + t - 1
    const int sw = (t >> 1) & 1;
    const int lchunk = (((pc >> 2) ^ sw) << 2) + (pc & 3);
    const int col = t - 1;
    const int scol = MODE == 1 ? (col >> 1) : col;      // (arithmetic shift: column -1 stays -1)
    const bool ok = tj0 + lchunk * 8 < a.Cj && t <= KW + 1;
    xtab[m] = ok ? (unsigned)(scol * a.ldr * 2 + tj0 * 2 + lchunk * 16) : ROW_OOB;
    xtabe[m] = (ok && (t == 0 || t == KW + 1)) ? LANE_OOB : xtab[m];
    xflag[m] = (ok ? 1 : 0) | ((t == 0 || t == KW + 1) ? 2 : 0);
  }

  // ---- the issue stream: position of the next step to request ---------------------------------------------------------
  int c_img, c_strip, c_hb;
  {
    const int per_img = a.nstrips * a.hsteps;
    c_img = u_beg / per_img;
    const int rem = u_beg - c_img * per_img;
    c_strip = rem / a.hsteps;
    c_hb = rem - c_strip * a.hsteps;
  }
  int m_hb = c_hb;            // the compute stream's row block (the same walk, NSL - 1 steps behind)
  int i_t = 0, i_vb = 0, i_ls = 0;   // next step to request: index, ring slot of its row j = 0, dy stage
  const bool dma_on = !(UZ_W9_FLAGS(a) & 32);   // measurement only: no requests after the prologue
  const unsigned ldl2 = (unsigned)(a.ldl * 2), ldr2 = (unsigned)(a.ldr * 2);

  // the x pieces of row J (0 .. G + 1) of the step being requested
#define UZ_W9_XROW(J)                                                                                                \
  do {                                                                                                               \
    const int row_ = h - 1 + (J);                                                                                    \
    const bool rv_ = valid && (unsigned)row_ < (unsigned)a.H;                                                        \
    const unsigned rb_ = MODE == 1 ? (((unsigned)c_img * a.Hr + (unsigned)(row_ >> 1)) * a.Wr + (unsigned)(w0 >> 1)) * ldr2   \
                                   : (pix0 + (unsigned)(((J) - 1) * a.W)) * ldr2;                                    \
    const unsigned rbase_ = rv_ ? rb_ : ROW_OOB;                                                                     \
    int slot_ = i_vb + (J);                                                                                          \
    if (slot_ >= NSR) slot_ -= NSR;                                                                                  \
    char* dst_ = smem + R_OFF + slot_ * ROWB + wave * 1024;                                                          \
    _Pragma("unroll") for (int m = 0; m < MR; ++m) {                                                                 \
      const int q_ = wave + NW * m;                                                                                  \
      const bool edge_ = (q_ == 0 && at_l) || (q_ == QR && at_r);   /* wave-uniform */                               \
      const unsigned voff_ = (edge_ ? xtabe[m] : xtab[m]) + rbase_;                                                  \
      if (NW * m + NW <= RPIECES || q_ < RPIECES)                                                                    \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rr, (lds_ptr_t)(dst_ + m * (NW * 1024)), 16, voff_, 0, 0, 0);       \
    }                                                                                                                \
  } while (0)

  auto issue_batch = [&]() __attribute__((always_inline)) {
    if (!dma_on && i_t >= NSL - 1) {   // (measurement build only)
      ++i_t;
      return;
    }
    const bool valid = i_t < nu;
    const bool start = c_hb == 0 || i_t == 0;
    const int h = c_hb * G, w0 = c_strip * KW;
    const bool at_l = w0 == 0, at_r = w0 + KW == a.W;
    const unsigned pix0 = ((unsigned)c_img * a.H + (unsigned)h) * a.W + (unsigned)w0;
    const unsigned lbase = valid ? pix0 * ldl2 : ROW_OOB;
    char* ldst = smem + L_OFF + i_ls * LSTAGE + wave * 1024;
#pragma unroll
    for (int k = 0; k < KL; ++k) {
      // (the sum in a variable of its own: written as the builtin's argument, hipcc 7.2's HOST pass drops the whole kernel
      // instantiation without a diagnostic and the library fails to load with an undefined kernel stub)
      const unsigned voff = ltab[k] + lbase;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(lr, (lds_ptr_t)(ldst + k * (NW * 1024)), 16, voff, 0, 0, 0);
    }
#pragma unroll
    for (int j = 2; j < G + 2; ++j) UZ_W9_XROW(j);
    if (start) {   // wave-uniform: the first step of an (image, strip) or of this workgroup also fetches rows h-1 and h
      UZ_W9_XROW(0);
      UZ_W9_XROW(1);
    }
    // advance the walk: rows fastest, then strips, then images
    if (++c_hb == a.hsteps) {
      c_hb = 0;
      if (++c_strip == a.nstrips) {
        c_strip = 0;
        ++c_img;
      }
    }
    i_vb += G + (c_hb == 0 ? 2 : 0);   // a segment start shares no row with its predecessor
    if (i_vb >= NSR) i_vb -= NSR;
    ++i_t;
    i_ls = (i_ls + 1 == NSL) ? 0 : i_ls + 1;
  };

  if (LD != 0 && loader) {
    // ---- a loader wave: request NSL - 1 steps ahead, publish a step when it has landed, one barrier per step ------------
    // XF: between "landed" and "published" the wave turns the NEW x rows of that step, piece by piece as it requested them
    // (its own vmcnt wait orders the LDS-DMA in front of its own reads), into relu(fma(x, scale, shift)) in place.  A
    // lane's 16 bytes are one 8-channel chunk of one pixel, and which chunk depends on the lane alone (the swizzle key of
    // slot pixel 8 q + (lane >> 3) is bit 4 of the lane): its 8 + 8 table values are kernel constants in registers.
    // Lanes whose request was out of range (zero padding above / below / beside the image, channel tails) keep their zeros.
    float xsc[8], xsh[8];
    int x_strip = c_strip, x_hb = c_hb, x_t = 0, x_vb = 0;   // the transform stream: the same walk as the requests
    if constexpr (XF) {
      const int sw = (lane >> 4) & 1, pc = lane & 7;
      const int lchunk = (((pc >> 2) ^ sw) << 2) + (pc & 3);
      const int ch = tj0 + lchunk * 8 < a.Cj ? tj0 + lchunk * 8 : 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        xsc[e] = a.xf_scale[ch + e];
        xsh[e] = a.xf_shift[ch + e];
      }
      // (used here: the wait for these loads then sits in front of the first LDS-DMA request, not among them)
#pragma unroll
      for (int e = 0; e < 8; ++e) asm volatile("" ::"v"(xsc[e]), "v"(xsh[e]));
    }
    auto xf_step = [&]() __attribute__((always_inline)) {
      if constexpr (XF) {
        const bool valid = x_t < nu;
        const bool start = x_hb == 0 || x_t == 0;
        const int h = x_hb * G, w0 = x_strip * KW;
        const bool at_l = w0 == 0, at_r = w0 + KW == a.W;
        // rows J = 2 .. G + 1 (and 0, 1 at a segment start), pieces m of this wave: at most (G + 2) MR
#pragma unroll
        for (int jj = 0; jj < G + 2; ++jj) {
          const int J = jj < G ? jj + 2 : jj - G;   // the G new rows, then the two more of a segment start
          if (jj >= G && !start) continue;
          const bool rv = valid && (unsigned)(h - 1 + J) < (unsigned)a.H;
          int slot = x_vb + J;
          if (slot >= NSR) slot -= NSR;
          const unsigned base = smem_u + (unsigned)(R_OFF + slot * ROWB + wave * 1024) + (unsigned)(lane << 4);
          u32x4 v[MR];
#pragma unroll
          for (int m = 0; m < MR; ++m)
            if (!(UZ_W9X_SKEL & 2) && (NW * m + NW <= RPIECES || wave + NW * m < RPIECES)) lds_rd16(v[m], base + m * (NW * 1024));
          if constexpr (MR == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0])::"memory");
          else if constexpr (MR == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1])::"memory");
          else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2])::"memory");
          static_assert(MR <= 3, "x pieces per wave and row");
#pragma unroll
          for (int m = 0; m < MR; ++m) {
            const int q = wave + NW * m;
            if (NW * m + NW <= RPIECES || q < RPIECES) {
              const bool edge = (q == 0 && at_l) || (q == QR && at_r);
              // (not a comparison of the offsets: the left halo column of an inner strip is a NEGATIVE offset from the strip's
              // first pixel, a huge unsigned that is perfectly in range once the row base is added)
              const bool ok = rv && (xflag[m] & 1) && !(edge && (xflag[m] & 2));
              typedef float f32x2 __attribute__((ext_vector_type(2)));
              typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
              typedef short s16x2 __attribute__((ext_vector_type(2)));
              u32x4 o = v[m];
#pragma unroll
              for (int i = 0; i < 4 && !(UZ_W9X_SKEL & 3); ++i) {
                const float x0 = u2f(v[m][i] << 16), x1 = u2f(v[m][i] & 0xffff0000u);
                const bf16x2 b = __builtin_convertvector(f32x2{fmaf(x0, xsc[2 * i], xsh[2 * i]), fmaf(x1, xsc[2 * i + 1], xsh[2 * i + 1])}, bf16x2);
                o[i] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, b), s16x2{0, 0}));
              }
              if (ok && !(UZ_W9X_SKEL & 2)) lds_wr16(base + m * (NW * 1024), o);
            }
          }
        }
        if (++x_hb == a.hsteps) {
          x_hb = 0;
          if (++x_strip == a.nstrips) x_strip = 0;
        }
        x_vb += G + (x_hb == 0 ? 2 : 0);
        if (x_vb >= NSR) x_vb -= NSR;
        ++x_t;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    };
#pragma unroll
    for (int b = 0; b < NSL - 1; ++b) issue_batch();
    wait_vmcnt<(NSL - 2) * PMIN>();
    xf_step();                          // step 0
    __builtin_amdgcn_s_barrier();
#pragma unroll 1
    for (int s = 0; s < nu; ++s) {
      wait_vmcnt<(NSL - 3) * PMIN>();   // step s + 1 has landed
      xf_step();                        // ... and is what the convolution in front of this layer fed its successor
      __builtin_amdgcn_s_barrier();     // ... for every wave; step s - 1 is read by nobody any more
      issue_batch();                    // step s + NSL - 1 into its slots
    }
    wait_vmcnt<0>();
    return;
  }

  f32x16 acc[TI][9];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;

  // ---- fragment read addresses: per lane constants + the step's stage / row slot; sub-step and pixel shifts are immediates
  // transposed-read roles: 16-lane group g, pixel row q4 and column quad p4 of the 4 x 16 block
  const int g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int lk = 8 * (g >> 1) + q4;           // pixel of this lane inside a 16-pixel sub-step
  const int lcol = 16 * (g & 1) + 4 * p4;     // channel inside a 32-channel tile
  // a pixel group's sub-steps start 64 / KG pixels into the step: KW = 64: half a row further (coff), KW <= 32: roff rows down
  constexpr int SPG = 4 / KG;                                // sub-steps per pixel group
  constexpr int RW = (KG == 2 ? (G >= 4 ? G / 2 : 1) : G) + 2;   // x rows a wave reads per step
  const int roff = (KG == 2 && G >= 2) ? kgrp * (G / 2) : 0;
  const int coff = (KG == 2 && G == 1) ? kgrp * 32 : 0;
  unsigned aoff[TI];
#pragma unroll
  for (int i = 0; i < TI; ++i)
    aoff[i] = (unsigned)(kgrp * (SPG * 16 * RBL) + lk * RBL + (((wi * TI + i) ^ (CPRL == 16 ? q4 : (q4 >> 1))) << 6) + lcol * 2);
  const unsigned boffV = (unsigned)((coff + lk) * 128 + ((wj ^ ((q4 >> 1) & 1)) << 6) + lcol * 2);             // pixels lk + {0..3, 4..7}
  const unsigned boffW = (unsigned)((coff + lk + 2) * 128 + ((wj ^ (((q4 + 2) >> 1) & 1)) << 6) + lcol * 2);   // the same two pixels on
  unsigned va[TI], vv[RW], vw[RW];   // this step's dy stage and the x rows this wave reads, as this lane reads them
  auto step_addr = [&](int vb, int ls) __attribute__((always_inline)) {
    const unsigned lb = smem_u + L_OFF + ls * LSTAGE;
#pragma unroll
    for (int i = 0; i < TI; ++i) va[i] = lb + aoff[i];
#pragma unroll
    for (int j = 0; j < RW; ++j) {
      int slot = vb + roff + j;
      if (slot >= NSR) slot -= NSR;
      const unsigned rb = smem_u + R_OFF + slot * ROWB;
      vv[j] = rb + boffV;
      vw[j] = rb + boffW;
    }
  };

  // unit U = (sub-step ks = U / 3, tap row ty = U % 3): 3 TI MFMAs per wave.  Its fragments are requested two units ahead.
  bf16x4 alo[2][TI] = {}, ahi[2][TI] = {}, vlo[3] = {}, vhi[3] = {}, wlo[3] = {}, whi[3] = {};
  auto fetch = [&](auto UC) __attribute__((always_inline)) {
    constexpr int U = decltype(UC)::value, ks = U / 3, ty = U % 3, r = (16 * ks) / KW, c0 = (16 * ks) % KW, set = U % 3;
    if (UZ_W9_FLAGS(a) & 256) return;   // measurement only: no fragment reads
    if constexpr (ty == 0) {
#pragma unroll
      for (int i = 0; i < TI; ++i) tr_pair<4 * RBL, ks * 16 * RBL>(alo[ks & 1][i], ahi[ks & 1][i], va[i]);
    }
    tr_pair<4 * 128, c0 * 128>(vlo[set], vhi[set], vv[r + ty]);
    tr_pair<4 * 128, c0 * 128>(wlo[set], whi[set], vw[r + ty]);
  };
  auto compute = [&](auto UC) __attribute__((always_inline)) {
    constexpr int U = decltype(UC)::value, ks = U / 3, ty = U % 3, set = U % 3;
    if constexpr (ty == 0) {
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        pin(alo[ks & 1][i]);
        pin(ahi[ks & 1][i]);
      }
    }
    pin(vlo[set]);
    pin(vhi[set]);
    pin(wlo[set]);
    pin(whi[set]);
    const bf16x8 f0 = __builtin_shufflevector(vlo[set], vhi[set], 0, 1, 2, 3, 4, 5, 6, 7);   // pixels 0..7
    const bf16x8 f2 = __builtin_shufflevector(wlo[set], whi[set], 0, 1, 2, 3, 4, 5, 6, 7);   // pixels 2..9
    const u32x2 xa = __builtin_bit_cast(u32x2, vlo[set]), xb = __builtin_bit_cast(u32x2, vhi[set]);
    const u32x2 wb = __builtin_bit_cast(u32x2, whi[set]);
    u32x4 s;   // pixels 1..8: every dword one pixel further
    s.x = __builtin_amdgcn_alignbit(xa.y, xa.x, 16);
    s.y = __builtin_amdgcn_alignbit(xb.x, xa.y, 16);
    s.z = __builtin_amdgcn_alignbit(xb.y, xb.x, 16);
    s.w = __builtin_amdgcn_alignbit(wb.y, xb.y, 16);
    const bf16x8 f1 = __builtin_bit_cast(bf16x8, s);
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      const bf16x8 afr = __builtin_shufflevector(alo[ks & 1][i], ahi[ks & 1][i], 0, 1, 2, 3, 4, 5, 6, 7);
      acc[i][ty * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, f0, acc[i][ty * 3 + 0], 0, 0, 0);
      acc[i][ty * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, f1, acc[i][ty * 3 + 1], 0, 0, 0);
      acc[i][ty * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, f2, acc[i][ty * 3 + 2], 0, 0, 0);
    }
  };
  // reads of unit U + 2 go out, then wait for everything older than unit U + 1's (LDS returns in order), then the MFMAs of U
#define UZ_W9_NREADS(U) (((U) % 3 == 0) ? 4 + 2 * TI : 4)
#define UZ_W9_UNIT(U)                                                  \
  do {                                                                 \
    fetch(IntC<(U) + 2>{});                                            \
    wait_lgkm<UZ_W9_NREADS((U) + 1) + UZ_W9_NREADS((U) + 2)>();        \
    if (mm) compute(IntC<(U)>{});                                      \
    __builtin_amdgcn_sched_barrier(0);                                 \
  } while (0)

  // ---- prologue: the first NSL - 1 steps -------------------------------------------------------------------------------
  if constexpr (LD == 0) {
#pragma unroll
    for (int b = 0; b < NSL - 1; ++b) issue_batch();
    wait_vmcnt<(NSL - 2) * PMIN>();   // the first step has landed (what may stay in flight: the steady pieces of the younger steps)
  }
  __builtin_amdgcn_s_barrier();

  const bool mm = !(UZ_W9_FLAGS(a) & 64);   // measurement only: no MFMAs
  const bool stagger_off = (UZ_W9_FLAGS(a) & 512) != 0;   // measurement only: both waves of a SIMD request together
  int m_vb = 0, m_ls = 0;
  step_addr(0, 0);
  fetch(IntC<0>{});
  fetch(IntC<1>{});
#pragma unroll 1
  for (int s = 0; s < nu; ++s) {
    if constexpr (NU == 12) {
      UZ_W9_UNIT(0);
      UZ_W9_UNIT(1);
      UZ_W9_UNIT(2);
      UZ_W9_UNIT(3);
      UZ_W9_UNIT(4);
      UZ_W9_UNIT(5);
    } else {
      UZ_W9_UNIT(0);
      UZ_W9_UNIT(1);
      UZ_W9_UNIT(2);
    }
    // mid-step: step s + 1 has landed for every wave, step s - 1 is read by nobody any more -> request step s + NSL - 1
    if constexpr (LD == 0) wait_vmcnt<(NSL - 3) * PMIN>();
    __builtin_amdgcn_s_barrier();
    // the two waves of a SIMD (w, w + 4) request at different times: while one does its bookkeeping and DMA issue (~100
    // instructions that issue no MFMA) the other one's MFMAs keep the matrix pipe busy
    if (LD == 0 && (NW == 4 || wave < 4 || stagger_off)) issue_batch();
    if constexpr (NU == 12) {
      UZ_W9_UNIT(6);
      UZ_W9_UNIT(7);
      UZ_W9_UNIT(8);
      if (LD == 0 && NW == 8 && wave >= 4 && !stagger_off) issue_batch();
      UZ_W9_UNIT(9);
    } else {
      if (LD == 0 && NW == 8 && wave >= 4 && !stagger_off) issue_batch();
      UZ_W9_UNIT(3);
    }
    // where the next step lives (the last read of this step's addresses was the request of unit NU - 1)
    if (++m_hb == a.hsteps) m_hb = 0;
    m_vb += G + (m_hb == 0 ? 2 : 0);
    if (m_vb >= NSR) m_vb -= NSR;
    m_ls = (m_ls + 1 == NSL) ? 0 : m_ls + 1;
    step_addr(m_vb, m_ls);
    // the last two units request units 0 and 1 of the next step (the MFMAs stay outside the branches: inside, hipcc keeps
    // a second copy of the accumulator registers)
    const bool more = s + 1 < nu;
    if (more) {
      fetch(IntC<0>{});
      wait_lgkm<4 + 4 + 2 * TI>();
    } else {
      wait_lgkm<4>();
    }
    if (mm) compute(IntC<NU - 2>{});
    __builtin_amdgcn_sched_barrier(0);
    if (more) {
      fetch(IntC<1>{});
      wait_lgkm<4 + 2 * TI + 4>();
    } else {
      wait_lgkm<0>();
    }
    if (mm) compute(IntC<NU - 1>{});
    __builtin_amdgcn_sched_barrier(0);
  }
#undef UZ_W9_UNIT
#undef UZ_W9_NREADS
#undef UZ_W9_XROW
  if constexpr (LD == 0) wait_vmcnt<0>();   // (the requests for steps past the end: zero fills nobody reads)

  if constexpr (KG == 2) {
    // the second pixel group hands its accumulators to its partner (same channel tile, same lane roles) through LDS:
    // [wave tile][tap][4 rows][lane][4 floats] -- a lane's 16 bytes, 1 KB per wave-instruction, no bank conflicts
    static_assert(TI == 1, "one accumulator tile per wave");
    __builtin_amdgcn_s_barrier();   // every wave has left the ring
    float* xch = reinterpret_cast<float*>(smem) + (size_t)wloc * (9 * 16 * 64) + lane * 4;
    if (kgrp == 1) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
          *reinterpret_cast<f32x4*>(xch + (t * 4 + r4) * 256) =
              f32x4{acc[0][t][4 * r4], acc[0][t][4 * r4 + 1], acc[0][t][4 * r4 + 2], acc[0][t][4 * r4 + 3]};
    }
    __syncthreads();
    if (kgrp == 1) return;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(xch + (t * 4 + r4) * 256);
        acc[0][t][4 * r4] += o.x;
        acc[0][t][4 * r4 + 1] += o.y;
        acc[0][t][4 * r4 + 2] += o.z;
        acc[0][t][4 * r4 + 3] += o.w;
      }
  }
  // ---- partial slab [split][tap][Ci][Cj] -------------------------------------------------------------------------------
  if (UZ_W9_FLAGS(a) & 128) return;   // measurement only: no slab
  const int cj = tj0 + wj * 32 + l31;
  const int ci0 = ti0 + wi * 32 * TI + 4 * lh;
  float* slab0 = a.slab + (size_t)z * 9 * (size_t)a.Ci * a.Cj + cj;
  if (ti0 + BI <= a.Ci) {   // (wave-uniform) whole dy tile inside: one lane mask for all stores
    if (cj < a.Cj) {
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            slab0[((size_t)t * a.Ci + ci0 + i * 32 + (r & 3) + 8 * (r >> 2)) * a.Cj] = acc[i][t][r];
    }
  } else {
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ci = ci0 + i * 32 + (r & 3) + 8 * (r >> 2);
          if (ci < a.Ci && cj < a.Cj) slab0[((size_t)t * a.Ci + ci) * a.Cj] = acc[i][t][r];
        }
  }
}

template <int BI, int KW> struct W9Nsl { static constexpr int value = W9<BI, KW>::NSL; };

}  // namespace

// plan: returns 1 and fills p (v9 = 1) when this kernel takes the descriptor
int uz_wgrad9_plan(const uz_wgrad_desc* d, UzWgrad2Plan* p) {
  const int f = uz_tune_flags();
  if (f & 0x8000000) return 0;   // ablation build: the round-3 kernels
  const bool up = d->taps_mode == UZ_TAPS_CONV_UP2;
  if (d->dtype != UZ_BF16 || !(d->taps_mode == UZ_TAPS_CONV || up) || d->ntaps != 9 || d->dil != 1) return 0;
  if (d->Ci % 8 != 0 || d->Cj % 8 != 0) return 0;
  const int W = d->W, H = d->H;
  if (!(W == 16 || W == 32 || (W >= 64 && W % 64 == 0))) return 0;
  const int kw = W < 64 ? W : 64, g = 64 / kw;
  if (H % g != 0) return 0;
  // 64 x 64 tiles on four compute + four loader waves everywhere; the 128 x 64 tile on eight do-everything waves (this
  // round's first form) stays behind an ablation switch.  Same box, B = 16 unet layers, uz_wgrad = kernel + reduction
  // (profiles/r04_kbench_wgrad_rowwalk_vs_r03.txt): 64 x 64 + loaders 1300 us over the fourteen layers, 128 x 64 (with the
  // round-3 kernels where those won) 1369, round 3 1462 -- and half the slab bytes (37.7 MB per launch).
  const int bi = (d->Ci > 64 && (f & 0x10000000)) ? 128 : 64;
  const int nsl = bi == 128 ? (kw == 64 ? W9Nsl<128, 64>::value : (kw == 32 ? W9Nsl<128, 32>::value : W9Nsl<128, 16>::value))
                            : (kw == 64 ? W9Nsl<64, 64>::value : (kw == 32 ? W9Nsl<64, 32>::value : W9Nsl<64, 16>::value));
  const int hsteps = H / g;
  if (hsteps < nsl - 1) return 0;   // at most one segment start among the steps of the ring
  const long long lbytes = ((long long)d->N * d->H * d->W - 1) * d->ldl * 2 + (long long)d->Ci * 2;
  const long long rbytes = ((long long)d->N * d->Hr * d->Wr - 1) * d->ldr * 2 + (long long)d->Cj * 2;
  if (lbytes >= 0x60000000LL || rbytes >= 0x60000000LL) return 0;   // (the kernel's out-of-range marks)
  const long long units = (long long)d->N * (W / kw) * hsteps;
  if (units >= (1LL << 30)) return 0;
  p->v9 = 1;
  p->bi = bi;
  p->kw = kw;
  p->kr = g;
  p->H = H;
  p->W = W;
  p->big = 0;
  p->one_tap = 0;
  p->gather = 0;
  p->wide9 = 0;
  p->kg = 1;
  p->tiles_i = (d->Ci + bi - 1) / bi;
  p->tiles_j = (d->Cj + 63) / 64;
  p->units = (int)units;
  const long long ntiles = (long long)p->tiles_i * p->tiles_j;
  // one workgroup per CU: a single round of <= UZ_NUM_CU workgroups, each with at least 8 steps
  long long split = ntiles >= UZ_NUM_CU ? 1 : UZ_NUM_CU / ntiles;
  const long long max_split = units / 8 > 0 ? units / 8 : 1;
  if (split > max_split) split = max_split;
  if (split > 8) split -= split % 8;   // a multiple of 8: the tiles of a split share an XCD
  if (split < 1) split = 1;
  p->upb = (int)((units + split - 1) / split);
  p->split = (int)((units + p->upb - 1) / p->upb);
  p->nslabs = p->split;
  return 1;
}

const char* uz_wgrad9_name(const UzWgrad2Plan& p) {
  return p.bi == 128 ? "wgrad9_bf16_128x64_rowwalk" : "wgrad9_bf16_64x64_rowwalk";
}

int uz_wgrad9_launch(const uz_wgrad_desc* d, const UzWgrad2Plan& p, const void* L, const void* R, float* slab, hipStream_t s,
                     const UzXf* xf) {
  Wg9Args a;
  a.xf_scale = xf ? xf->scale : nullptr;
  a.xf_shift = xf ? xf->shift : nullptr;
  if (xf) UZ_REQUIRE(p.bi == 64 && xf->scale && xf->shift && !(uz_tune_flags() & 0x22000000),
                     "uz_wgrad_xf: the input transform is the loader-wave form's (64 x 64 tiles)");
  a.L = L;
  a.R = R;
  a.slab = slab;
  a.lbytes = (unsigned)(((long long)d->N * d->H * d->W - 1) * d->ldl * 2 + (long long)d->Ci * 2);
  a.rbytes = (unsigned)(((long long)d->N * d->Hr * d->Wr - 1) * d->ldr * 2 + (long long)d->Cj * 2);
  a.H = p.H;
  a.W = p.W;
  a.Ci = d->Ci;
  a.ldl = d->ldl;
  a.Cj = d->Cj;
  a.ldr = d->ldr;
  a.Hr = d->Hr;
  a.Wr = d->Wr;
  a.nstrips = p.W / p.kw;
  a.hsteps = p.H / p.kr;
  a.units = p.units;
  a.upb = p.upb;
  a.tiles_j = p.tiles_j;
  a.ntiles = p.tiles_i * p.tiles_j;
  a.split = p.split;
  a.flags = uz_tune_flags();
  const bool up = d->taps_mode == UZ_TAPS_CONV_UP2;
  const dim3 grid((unsigned)(a.ntiles * p.split));
#define UZ_W9_LAUNCH2(BI_, KW_, NW_, KG_)                                                                  \
  do {                                                                                                     \
    if (up) hipLaunchKernelGGL((wgrad9_kernel<BI_, KW_, 1, NW_, KG_>), grid, dim3(64 * NW_), 0, s, a);     \
    else hipLaunchKernelGGL((wgrad9_kernel<BI_, KW_, 0, NW_, KG_>), grid, dim3(64 * NW_), 0, s, a);        \
  } while (0)
#define UZ_W9_LAUNCH1(BI_, KW_, NW_) UZ_W9_LAUNCH2(BI_, KW_, NW_, 1)
#define UZ_W9_LAUNCHL(BI_, KW_)                                                                            \
  do {                                                                                                     \
    if (xf && up) hipLaunchKernelGGL((wgrad9_kernel<BI_, KW_, 1, 4, 1, 1, true>), grid, dim3(512), 0, s, a);     \
    else if (xf) hipLaunchKernelGGL((wgrad9_kernel<BI_, KW_, 0, 4, 1, 1, true>), grid, dim3(512), 0, s, a);      \
    else if (up) hipLaunchKernelGGL((wgrad9_kernel<BI_, KW_, 1, 4, 1, 1>), grid, dim3(512), 0, s, a);      \
    else hipLaunchKernelGGL((wgrad9_kernel<BI_, KW_, 0, 4, 1, 1>), grid, dim3(512), 0, s, a);              \
  } while (0)
#ifdef UZ_ABLATE
  // the measurement build (make ABLATE=1) keeps the forms this round's plan was chosen against: the 128 x 64 tile on eight
  // do-everything waves (0x10000000) or on four waves of 64 x 32 (+ 0x20000000), the 64 x 64 tile on four do-everything
  // waves (0x20000000) or on eight waves in two pixel groups (0x2000000).  The shipped library does not instantiate them
  // (round-4 review: 36 kernels, 2 MB of the .so, that no plan of the shipped build can select).
  const bool w4 = (a.flags & 0x20000000) != 0;
  if (p.bi == 128) {
    if (p.kw == 64) { if (w4) UZ_W9_LAUNCH1(128, 64, 4); else UZ_W9_LAUNCH1(128, 64, 8); }
    else if (p.kw == 32) { if (w4) UZ_W9_LAUNCH1(128, 32, 4); else UZ_W9_LAUNCH1(128, 32, 8); }
    else { if (w4) UZ_W9_LAUNCH1(128, 16, 4); else UZ_W9_LAUNCH1(128, 16, 8); }
  } else {
    const int form = (a.flags & 0x20000000) ? 1 : ((a.flags & 0x2000000) ? 2 : 0);
    if (p.kw == 64) { if (form == 1) UZ_W9_LAUNCH1(64, 64, 4); else if (form == 2) UZ_W9_LAUNCH2(64, 64, 8, 2); else UZ_W9_LAUNCHL(64, 64); }
    else if (p.kw == 32) { if (form == 1) UZ_W9_LAUNCH1(64, 32, 4); else if (form == 2) UZ_W9_LAUNCH2(64, 32, 8, 2); else UZ_W9_LAUNCHL(64, 32); }
    else { if (form == 1) UZ_W9_LAUNCH1(64, 16, 4); else if (form == 2) UZ_W9_LAUNCH2(64, 16, 8, 2); else UZ_W9_LAUNCHL(64, 16); }
  }
#else
  UZ_REQUIRE(p.bi == 64, "uz_wgrad9: the 128 x 64 tile exists in the measurement build only");
  if (p.kw == 64) UZ_W9_LAUNCHL(64, 64);
  else if (p.kw == 32) UZ_W9_LAUNCHL(64, 32);
  else UZ_W9_LAUNCHL(64, 16);
#endif
#undef UZ_W9_LAUNCH1
#undef UZ_W9_LAUNCH2
#undef UZ_W9_LAUNCHL
  UZ_LAUNCH_CHECK("uz_wgrad(row walk)");
  return UZ_OK;
}
