// Direct 3x3 convolution, third generation ("ping-pong"), bf16, gfx950 (MI355X), NHWC.
//
// Stands in for nn.Conv2d(k=3, padding=1) forward and its input gradient (reference:
// unet_zoo/models/common_layers.py:28,31,47,52,71; autograd a19) wherever the input channels come in multiples of 32.
// Same arithmetic as uz_conv3x3.hip (bf16 operands, fp32 accumulation on the matrix cores, result rounded once to bf16,
// BatchNorm sums of the STORED values), a different schedule:
//
//  * K unit = (tap, 32 input channels).  The halo patch of a 32-channel slab is PW x PHP rows of 64 bytes, column-major
//    with an odd column height PHP: a tap, a patch row and a 16-column step are immediate offsets of the ds_read, the
//    swizzle key depends on the patch column only.  Two patch buffers; a ring of weight tiles [BN][64 B] per unit.
//  * v_mfma_f32_16x16x32_bf16 with the weights as the row operand: a lane owns 4 consecutive channels of one pixel per
//    accumulator.  A 16-pixel tile is 16 consecutive patch columns of one patch row; the four 16-byte chunks of an LDS row
//    are rotated by 2 * (column >> 2), which keeps every ds_read_b128 lane group conflict-free at every tap shift.
//    (Measured against the 32x32x16 stream on the same tiles, same box, bit-identical results: 1186 -> 1255 TFLOP/s over
//    the twelve unet layers of the first configuration -- the chip holds a higher clock on the stream that moves half the
//    accumulator bytes per flop.  MFMAs alone on real data, no fragment reads, no DMA: 1500 TFLOP/s = the ceiling of this
//    structure at the clock the chip holds.)
//  * PING-PONG: the two waves that share a SIMD (w, w + 4) never compute at the same time.  Between two s_barriers one
//    group issues the MFMAs of a phase while the other reads the fragments of its next phase from LDS and issues its
//    LDS-DMA pieces, then the roles swap.  (Lockstep, one barrier per phase: 8 % slower.  DMA issue moved into the MFMA
//    stream: 7 % slower.  s_setprio, and where the lgkmcnt wait sits: nothing.)
//  * The unit stream runs across the tiles of a workgroup: the last slab of a tile requests the first patch and the first
//    weight tiles of the next one.  The epilogue is wave-local (no barrier): 32-pixel rounds through a private 4.5 KB
//    staging strip, 16-byte buffer stores of full 128-byte lines, statistics from the stored values; both groups run it
//    in the same barrier interval.  (Tried on the 64-accumulator configurations and dropped: outputs packed to bf16 at
//    the tile's end, the two store rounds run at the head of the next tile's first two read phases -- bit-identical,
//    but 64 -> 64 @ 256 x 256 86 -> 100 us, 128 -> 64 137 -> 151: a round is longer than the 768-cycle compute phase it
//    was meant to hide under, so it only moved the exposed time and added the packing.)
//
// Tile configurations (template parameters TH x TW pixels, WM x WN waves, UPP units per phase):
//   16 x 32, 4 x 2, 1   512 pixels x 128 channels, wave 128 x 64: the layers with >= 128 output channels and enough
//                       pixels for one tile per CU (6 fragment reads per 16 MFMA cycles x 32, 24.5 KB of DMA per K = 64)
//   16 x 32, 8 x 1, 3   512 pixels x 64 channels, wave 64 x 64, a phase = one tap row: 64 output channels
//    8 x 32, 4 x 2, 3   256 pixels x 128 channels, wave 64 x 64: 32 x 32 ... maps, where 512-pixel tiles leave CUs idle
//   16 x 16, 4 x 2, 3   256 pixels (one 16 x 16 map) x 128 channels
#include "uz_common.h"

namespace {

struct PpArgs {
  const void* x;
  const void* w;
  void* y;
  const float* bias;
  float* stats;
  unsigned xbytes, wbytes, ybytes;
  int N, H, W, Cin, ldx, Nout, ldy, K;
  int th_n, tw_n, ntiles;
  int ups;
  const void* bn_y;
  const float* bn_scale;
  const float* bn_shift;
  const float* bn_mean;
  const float* bn_invstd;
  int ld_bny;
  // split-K (template parameter SPLIT): blockIdx.z owns the channel slabs [z * cps, (z + 1) * cps) and writes its fp32
  // partial tile to part[z][pixel][channel]; igemm_split_reduce_kernel adds them in fixed order (+ bias, statistics)
  float* part;
  int cps;
  // XF (template parameter): x holds the RAW output of the preceding convolution; its BatchNorm + ReLU
  // a = relu(fma(x, xf_scale[c], xf_shift[c])) (rounded to bf16 as uz_bn_relu_apply would store it) is applied to the halo
  // patch inside LDS, so the normalised activation never exists in HBM (DoubleConv's middle tensor, common_layers.py:28-33)
  const float* xf_scale;
  const float* xf_shift;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(3))) char* lds_char_ptr;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned OOB = 0x80000000u;   // beyond any descriptor's num_records (tensors < 2 GiB)
constexpr int RB = 64;                  // bytes per LDS row = 32 channels
constexpr int KU = 32;                  // K per unit
constexpr int STG_ROW = 144;            // staging row: 64 channels + 16 bytes
constexpr int STG_W = 32 * STG_ROW;     // a wave's staging strip: 32 pixels (also parks its 64 x 64 B of running sums)

// XM (the XF instantiations): another deal of the halo pieces to the waves, see piece() below
template <int TH_, int TW_, int WM_, int WN_, int UPP_, bool XM_ = false>
struct PpCfg {
  static constexpr int TH = TH_, TW = TW_, WM = WM_, WN = WN_, UPP = UPP_;
  static constexpr bool XM = XM_;
  static_assert(WM * WN == 8 && (TW == 32 || TW == 16) && TH % WM == 0 && 9 % UPP == 0, "tile configuration");
  static constexpr int PH = TH + 2, PW = TW + 2, PHP = PH | 1;
  static constexpr int PROWS = PW * PHP;
  static constexpr int APIECES = (PROWS * RB + 1023) / 1024;   // 1 KB pieces (16 rows) of a patch
  static constexpr int A_BYTES = APIECES * 1024;
  static constexpr int APW = XM ? 8 : (APIECES + 7) / 8;       // pieces per wave and slab (beyond APIECES: dummies)
  // Halo piece k of wave w.  Plain: w + 8 k.  XM (input transform, three phases per slab): the arithmetic of a piece
  // requested in phase q rides in the MFMA gaps of the requesting wave's compute phase q + 1 -- but group 1's LAST compute
  // phase runs beside group 0's first reads of the next slab, so group 1 must own no piece of the second batch.  Group 1
  // (waves 4-7): pieces 0 .. 15, all in the first batch (k < 4; k >= 4 dummies); group 0: 16 .. 27 in the first batch
  // (k < 3, k = 3 a dummy), 28 ... in the second (k = 4 .. 7).  Four requests per wave in each of the first two phases:
  // two dummies per wave and slab more than the plain deal, and no transform left in any read phase.
  static constexpr int piece(int w, int k) {
    if (!XM) return w + 8 * k;
    if (w >= 4) return k < 4 ? (w - 4) + 4 * k : APIECES + 64;
    return k < 3 ? 16 + w + 4 * k : (k == 3 ? APIECES + 64 : 28 + w + 4 * (k - 4));
  }
  static_assert(!XM || (9 / UPP_ == 3 && APIECES > 28 && APIECES <= 44), "the XM deal is made for three phases and 29 .. 44 pieces");
  static constexpr int BN = 64 * WN;
  static constexpr int B_UNIT = BN * RB;                       // weight tile of a unit
  static constexpr int BPU = B_UNIT / 1024;                    // pieces per unit (8 or 4)
  static constexpr int NPH = 9 / UPP;                          // phases per slab
  static constexpr int BPP = UPP * BPU;                        // weight pieces per phase
  static constexpr int NBJ = (BPP + 7) / 8;                    // ... per wave (the last round may cover waves 0-3 only)
  static constexpr int NSLOT = UPP == 1 ? 5 : 3;               // weight ring, in phases
  static constexpr int DPH = NSLOT - 1;                        // weight tiles are requested DPH phases ahead
  static constexpr int ROWS_W = TH / WM;                       // patch rows of a wave
  static constexpr int HALVES = TW / 16;
  static constexpr int PT = ROWS_W * HALVES;                   // 16-pixel tiles of a wave
  static constexpr int CT = 4;                                 // 16-channel tiles of a wave
  static_assert(PT % 2 == 0, "epilogue rounds are 32 pixels");
  static constexpr int NSTORE = 2 * PT;                        // epilogue stores per wave and tile
  static constexpr int OFF_B = 2 * A_BYTES;
  static constexpr int OFF_STG = OFF_B + NSLOT * UPP * B_UNIT;
  static constexpr int OFF_SCR = OFF_STG + 8 * STG_W;          // 1 KB that swallows the dummy pieces
  static constexpr int OFF_BIAS = OFF_SCR + 1024;
  static constexpr int SMEM_BYTES = OFF_BIAS + BN * 4;
  static_assert(SMEM_BYTES <= 160 * 1024, "LDS budget");
  static_assert(UPP * (PT + CT) * 4 + PT * CT * 4 <= 200, "register budget: fragments + accumulators");
  // XF instantiations: what is left of the 160 KB holds the (scale, shift) table of the input channels, 64 bytes per
  // 8-channel chunk [scale x 8 | shift x 8] -- one base address and four ds_read_b128 per transformed piece
  static constexpr int OFF_XF = SMEM_BYTES;
  static constexpr int XF_CHUNKS = (160 * 1024 - SMEM_BYTES) / 64;
  static constexpr int XF_CH = XF_CHUNKS * 8;                  // input channels the table can hold
  // a halo piece requested in phase q is transformed in place at the end of read phase q + XD by the wave that requested it
  // (its own vmcnt wait orders the LDS-DMA in front of its own reads; the phase's barrier publishes the result)
  static constexpr int XD = 1;
  // vmcnt(N) before that transform: everything up to phase p - XD has landed, the requests of the XD youngest phases may fly
  static constexpr int wait_xf(int p, int grp) {
    int n = 0;
    for (int k = p - XD + 1; k <= p; ++k) n += nb(grp) + na(k);
    return n;
  }
  static constexpr bool xf_here(int p) { return p - XD >= 0 && na(p - XD) > 0; }

  // halo pieces a wave requests in phase p (for the next slab): none in a slab's last phase when a slab has few phases
  // (they must have landed when that phase ends)
  static constexpr int na(int p) {
    if (XM) return p < 2 ? 4 : 0;
    if (NPH == 9) return p < APW ? 1 : 0;
    const int first = (APW + NPH - 2) / (NPH - 1);
    int left = APW;
    for (int k = 0; k < NPH - 1; ++k) {
      const int n = left < first ? left : first;
      if (k == p) return n;
      left -= n;
    }
    return 0;
  }
  static constexpr int na_before(int p) {   // halo pieces requested in phases < p of the same slab
    int n = 0;
    for (int k = 0; k < p; ++k) n += na(k);
    return n;
  }
  static constexpr int nb(int grp) {   // weight pieces a wave of group grp requests per phase
    int n = 0;
    for (int j = 0; j < NBJ; ++j)
      if (grp * 4 + 8 * j < BPP) ++n;   // (piece index = wave + 8 j; the waves of a group have the same count)
    return n;
  }
  // vmcnt(N) at the end of the read phase p.  A wave requests per phase [its weight pieces of phase p + DPH][its halo pieces
  // na(p) of the next slab].  Before the barrier that precedes any read of phase p + 1, the wave's own pieces of phase
  // p + 1 (requested in phase p + 1 - DPH) must have landed: what may stay in flight is everything younger.  Before a
  // slab's first phase the whole patch of that slab must be in as well.
  static constexpr int wait_normal(int p, int grp) {
    int n = na((p + 1 - DPH + 9 * NPH) % NPH);
    for (int k = p + 2 - DPH; k <= p; ++k) n += nb(grp) + na((k + 9 * NPH) % NPH);
    if (p == NPH - 1) {
      int last = 0;
      for (int k = 0; k < NPH; ++k)
        if (na(k) > 0) last = k;
      const int m = (NPH - 1 - last) * nb(grp);
      if (m < n) n = m;
    }
    return n;
  }
  // last slab of a workgroup's last tile: no halo pieces, no weight pieces beyond the last phase
  static constexpr int wait_nonext(int p, int grp) {
    int n = 0;
    if (p + 1 - DPH < 0) n += na((p + 1 - DPH + 9 * NPH) % NPH);   // (a phase of the previous slab: that one was normal)
    for (int k = p + 2 - DPH; k <= p; ++k) {
      if (k < 0) n += nb(grp) + na((k + 9 * NPH) % NPH);
      else if (k + DPH < NPH) n += nb(grp);
    }
    return n;
  }
};
typedef PpCfg<16, 32, 4, 2, 1> Cfg512;
static_assert(Cfg512::wait_normal(0, 0) == 4 && Cfg512::wait_normal(3, 1) == 7 && Cfg512::wait_normal(8, 0) == 3, "wait table");
static_assert(Cfg512::wait_nonext(0, 0) == 3 && Cfg512::wait_nonext(5, 0) == 2 && Cfg512::wait_nonext(7, 1) == 0, "wait table");
static_assert(Cfg512::SMEM_BYTES == 163328 && Cfg512::NSTORE == 16, "512 x 128 configuration");

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, unsigned voff, unsigned soff) {
  // one wave-instruction: lane i writes LDS bytes [base + 16 i, +16) with the 16 bytes at voff + soff (zeros when out of
  // range; masked lanes carry voff = OOB, which is out of range whether or not the scalar offset takes part in the check).
  // The wave-uniform part of the address (tap, channel slab) travels in the scalar offset: added to the lane offsets it
  // would be hoisted out of the tile loop as one more register per tap.
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)lds_wave_base, 16, voff, soff, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt range");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int OFF> __device__ __forceinline__ void lds_read16(f32x4& dst, unsigned lds_addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "n"(OFF));
}
__device__ __forceinline__ void pin16(f32x4& v) { asm volatile("" : "+v"(v)); }
template <int V> struct IntC { static constexpr int value = V; };

#ifndef UZ_XF_SKEL
#define UZ_XF_SKEL 0   // measurement builds of the XF form: 1 no arithmetic (the staged bytes go back as they are), 2 no staging
                      // reads and no write-back either (the deal of the pieces and the table only)
#endif
#ifndef UZ_PP_SKEL
#define UZ_PP_SKEL 0   // measurement builds: 1 no fragment reads, 2 no MFMAs, 4 no DMA after the prologue, 8 no epilogue,
                      // 16 fragment reads in a tile's first phase only, 128 both groups in lockstep (timing only)
#endif

// (a function, not __builtin_bit_cast(float, vec[i]) in place: with a vector element as its direct operand hipcc 7.2 reads
// element 0 whatever the subscript -- found when every channel was scaled by its chunk's first scale)
__device__ __forceinline__ float u2f(unsigned u) { return __builtin_bit_cast(float, u); }
// one v_fma_f32, opaque to the SLP vectoriser: left alone it pairs neighbouring fmas into v_pk_fma_f32, which beside MFMAs
// costs more issue cycles than the two scalar instructions it replaces (MI355X_MICROARCH.md, "price of one filler")
__device__ __forceinline__ float fma1(float a, float b, float c) {
  float r;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
template <int OFF> __device__ __forceinline__ void lds_read16u(u32x4& dst, unsigned lds_addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "n"(OFF));
}
template <int OFF> __device__ __forceinline__ void lds_write4u(unsigned lds_addr, unsigned v) {
  asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(lds_addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ void lds_write16u(unsigned lds_addr, const u32x4& v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr), "v"(v) : "memory");
}

template <typename C, bool BNRED, bool SPLIT = false, bool XF = false>
__global__ __launch_bounds__(512, 2) void conv3x3_pp_kernel(const PpArgs a) {
  static_assert(!(BNRED && SPLIT), "the split-K form has no fused epilogue");
  static_assert(!(BNRED && XF), "the fused BatchNorm-backward sums belong to input gradients, the input transform to forwards");
  typedef bf16_t T;
  constexpr int ES = 2, VEC = 8;
  constexpr int TH = C::TH, TW = C::TW, PH = C::PH, PHP = C::PHP, PROWS = C::PROWS, APIECES = C::APIECES, A_BYTES = C::A_BYTES;
  constexpr int APW = C::APW, BN = C::BN, B_UNIT = C::B_UNIT, BPU = C::BPU, NPH = C::NPH, BPP = C::BPP, NBJ = C::NBJ;
  constexpr int NSLOT = C::NSLOT, DPH = C::DPH, UPP = C::UPP, ROWS_W = C::ROWS_W, HALVES = C::HALVES, PT = C::PT, CT = C::CT;
  constexpr int OFF_B = C::OFF_B, OFF_STG = C::OFF_STG, OFF_SCR = C::OFF_SCR, OFF_BIAS = C::OFF_BIAS, NSTORE = C::NSTORE;
  constexpr int SLOT_BYTES = UPP * B_UNIT;
  __shared__ __attribute__((aligned(1024))) char smem[C::SMEM_BYTES + (XF ? C::XF_CHUNKS * 64 : 0)];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                                  // ping-pong group: waves w and w + 4 share a SIMD
  const int wm = C::WN == 2 ? wave >> 1 : wave, wn = C::WN == 2 ? wave & 1 : 0;   // wave tile: patch rows ROWS_W wm .., channels 64 wn ..
  const int l15 = lane & 15, lq = lane >> 4;
  const int n0 = blockIdx.y * BN;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.wbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.ybytes, 0x00020000);
  const unsigned smem_u = (unsigned)(size_t)(lds_char_ptr)smem;

  // ---- fragment read addresses -------------------------------------------------------------------------------------
  // patch pixel (pi, pj) lives in LDS row pj * PHP + pi; weight rows are output channels.  A row's four 16-byte chunks are
  // rotated by 2 * (col >> 2), col = patch column / channel, on the DMA source side and on the read alike.  A lane reads
  // pixel column (lane & 15) + 16 h + tx, K chunk lane >> 4: one base per tx; patch row and column half h are immediate
  // offsets (16 columns further the rotation is the same).
  auto swz = [](int chunk, int col) { return (chunk + 2 * (col >> 2)) & 3; };
  auto unswz = [](int phys, int col) { return (phys - 2 * (col >> 2)) & 3; };
  unsigned a_base[3], b_base;
#pragma unroll
  for (int tx = 0; tx < 3; ++tx) {
    const int pj = l15 + tx;
    a_base[tx] = smem_u + (pj * PHP + ROWS_W * wm) * RB + (swz(lq, pj) << 4);
  }
  {
    const int brow = wn * 64 + l15;
    b_base = smem_u + OFF_B + brow * RB + (swz(lq, brow) << 4);
  }
  // ---- LDS-DMA source offsets --------------------------------------------------------------------------------------
  // weight piece wave + 8 j of a phase: unit (wave + 8 j) / BPU, rows 16 ((wave + 8 j) % BPU) + (lane >> 2): the row part is
  // the same for every j (BPU divides 8); this lane fetches the chunk that belongs at (lane & 3)
  unsigned bvoff;
  {
    const int brow = (wave % BPU) * 16 + (lane >> 2);
    bvoff = (n0 + brow < a.Nout) ? ((unsigned)(n0 + brow) * (unsigned)a.K * ES + (unswz(lane & 3, brow) << 4)) : OOB;
  }
  unsigned avoff[APW];   // halo pieces wave + 8 k of the tile whose patches are being requested (channel slab 0)
  auto compute_avoff = [&](int im, int hh0, int ww0) {
    // (an opaque copy of the lane id: everything below is tile-invariant up to the last two lines, and hoisted out of the
    // tile loop it would hold a dozen registers through the main loop -- the kernel then spills its DMA offsets)
    int ln = lane;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int k = 0; k < APW; ++k) {
      const int r = C::piece(wave, k) * 16 + (ln >> 2);
      const int pj = r / PHP, pi = r - pj * PHP;
      const int hh = hh0 - 1 + pi, ww = ww0 - 1 + pj;
      const bool ok = r < PROWS && pi < PH && (unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)a.W;
      const unsigned pix = a.ups ? (unsigned)((im * (a.H >> 1) + (hh >> 1)) * (a.W >> 1) + (ww >> 1))
                                 : (unsigned)((im * a.H + hh) * a.W + ww);
      avoff[k] = ok ? (pix * (unsigned)a.ldx + (unswz(ln & 3, pj) << 3)) * ES : OOB;
    }
  };
  auto issue_a = [&](auto kc, int buf, int cslab) {
    constexpr int k = decltype(kc)::value;
    if constexpr (k < APW) {
      const int piece = C::piece(wave, k);
      char* dst = piece < APIECES ? smem + buf * A_BYTES + piece * 1024 : smem + OFF_SCR;
      dma16(xr, dst, avoff[k], (unsigned)(cslab * KU * ES));
    }
  };
  // weight pieces of phase ph (units UPP ph .. UPP ph + UPP - 1 of slab cslab) into ring slot `slot`
  auto issue_b = [&](int slot, int cslab, int ph) {
#pragma unroll
    for (int j = 0; j < NBJ; ++j) {
      const int idx = wave + 8 * j;
      if (idx < BPP) {   // wave-uniform
        const int tap = UPP * ph + idx / BPU;
        dma16(wr, smem + OFF_B + slot * SLOT_BYTES + idx * 1024, bvoff, (unsigned)((tap * a.Cin + cslab * KU) * ES));
      }
    }
  };
  auto decode = [&](int tile, int& im, int& hh0, int& ww0) {
    const int per = a.th_n * a.tw_n;
    im = tile / per;
    const int rem = tile - im * per;
    const int ti = rem / a.tw_n;
    hh0 = ti * TH;
    ww0 = (rem - ti * a.tw_n) * TW;
  };

  // bias table (fp32, the channels of this workgroup): the accumulators' initial value
  float* const sBias = reinterpret_cast<float*>(smem + OFF_BIAS);
  if (tid < BN) sBias[tid] = (!BNRED && !SPLIT && a.bias != nullptr && n0 + tid < a.Nout) ? a.bias[n0 + tid] : 0.f;
  {   // running BatchNorm sums of this lane (channel chunk lane & 7 of the read-back phase): zero, in the staging strip
    f32x4* sp = reinterpret_cast<f32x4*>(smem + OFF_STG + wave * STG_W + lane * 64);
    sp[0] = sp[1] = sp[2] = sp[3] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int ncb = a.Cin / KU;
  // the slabs of this workgroup: all of them, or (split-K: 16 x 16 bottleneck maps whose tiles fill a quarter of the chip)
  // the z-th range of cps slabs
  const int cbeg = SPLIT ? (int)blockIdx.z * a.cps : 0;
  const int cend = SPLIT ? (cbeg + a.cps < ncb ? cbeg + a.cps : ncb) : ncb;
  if constexpr (XF) {   // (scale, shift) of this workgroup's input channels [32 cbeg, 32 cend): chunk q = [scale x 8 | shift x 8]
    float* const sXf = reinterpret_cast<float*>(smem + C::OFF_XF);
    for (int i = tid; i < (cend - cbeg) * KU; i += 512) {
      const int ch = cbeg * KU + i;
      sXf[(i >> 3) * 16 + (i & 7)] = a.xf_scale[ch];
      sXf[(i >> 3) * 16 + 8 + (i & 7)] = a.xf_shift[ch];
    }
  }
  // XF: halo piece wave + 8 k of slab cs, landed in patch buffer buf, becomes relu(fma(x, scale, shift)) in place, by the wave
  // that requested it (its own vmcnt wait orders the LDS-DMA in front of its own reads).  Of its piece's row (lane >> 2) a
  // lane takes LOGICAL chunk lane & 3 -- physical chunk swz(lane & 3, patch column) -- so its 8 + 8 table values are the
  // same for every piece of a slab and live in registers (xt).  A row whose request was out of range (zero padding, rows
  // beyond the patch) keeps its zeros.  All LDS accesses from inline asm: beside LDS-DMA in flight hipcc would put
  // s_waitcnt vmcnt(0) in front of its own.
  u32x4 xd[4] = {}, xt[4] = {};   // staged raw bytes of up to four pieces; [scale 0..3 | 4..7 | shift 0..3 | 4..7]
  unsigned xad[4] = {};           // where they go back to
  auto xf_addr = [&](auto kc, int buf) __attribute__((always_inline)) -> unsigned {
    constexpr int k = decltype(kc)::value;
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int pc_ = C::piece(wave, k);
    const int piece = pc_ < APIECES ? pc_ : 0;   // (a dummy: any address inside the patch, its bytes are never written back)
    const int r = piece * 16 + (ln >> 2);
    const int pj = r / PHP;
    return smem_u + (unsigned)(buf * A_BYTES + piece * 1024) + (unsigned)((ln >> 2) << 6) + (unsigned)(swz(ln & 3, pj) << 4);
  };
  auto xf_table = [&](int cs) __attribute__((always_inline)) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const unsigned taddr = smem_u + (unsigned)C::OFF_XF + (unsigned)((((cs - cbeg) << 2) + (ln & 3)) << 6);
    lds_read16u<0>(xt[0], taddr);
    lds_read16u<16>(xt[1], taddr);
    lds_read16u<32>(xt[2], taddr);
    lds_read16u<48>(xt[3], taddr);
  };
  auto xf_math = [&](const u32x4& v) __attribute__((always_inline)) -> u32x4 {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float sc0 = u2f(xt[i >> 1][2 * (i & 1)]), sc1 = u2f(xt[i >> 1][2 * (i & 1) + 1]);
      const float sh0 = u2f(xt[2 + (i >> 1)][2 * (i & 1)]), sh1 = u2f(xt[2 + (i >> 1)][2 * (i & 1) + 1]);
      const float x0 = u2f(v[i] << 16), x1 = u2f(v[i] & 0xffff0000u);
      // fma in fp32, round to nearest even (v_cvt_pk_bf16_f32), then the ReLU on the packed pair: a negative bf16 is a
      // negative int16, so max(., 0) as integers is max(., +0.0) -- the bits uz_bn_relu_apply stores (a NaN stays a NaN)
      const bf16x2 b = __builtin_convertvector(f32x2{fmaf(x0, sc0, sh0), fmaf(x1, sc1, sh1)}, bf16x2);
      o[i] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, b), s16x2{0, 0}));
    }
    return o;
  };
  // whether piece wave + 8 k exists (the rest are the dummies that keep the request counts equal): wave-uniform
  auto xf_real = [&](int k) { return C::piece(wave, k) < APIECES; };
  // one piece at once (prologue; group 1 in a slab's last read phase): read, wait, transform, write
  auto xf_now = [&](auto kc, int buf) __attribute__((always_inline)) {
    constexpr int k = decltype(kc)::value;
    if constexpr (XF && k < APW) {
      if (xf_real(k)) {
        const unsigned ad = xf_addr(kc, buf);
        u32x4 v;
        lds_read16u<0>(v, ad);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v), "+v"(xt[0]), "+v"(xt[1]), "+v"(xt[2]), "+v"(xt[3])::"memory");
        const u32x4 o = xf_math(v);
        if (avoff[k] != OOB) lds_write16u(ad, o);   // rows outside the image keep the zeros the LDS-DMA put there
      }
    }
  };
  // acc[pt][ct]: 16-pixel tile pt = HALVES * (patch row of the wave) + column half, 16-channel tile ct
  f32x4 acc[PT][CT];
  f32x4 fa[UPP][PT], fb[UPP][CT];

  int apar = 0;    // patch buffer of the current slab
  int bslot = 0;   // weight slot of the current phase
  int img = 0, h0 = 0, w0 = 0, nim = 0, nh0 = 0, nw0 = 0;

  // ---- prologue: first patch, first DPH weight phases -------------------------------------------------------------------
  static_assert(DPH <= NPH, "the weight prefetch spans at most one slab boundary");
  if ((int)blockIdx.x < a.ntiles) {
    decode(blockIdx.x, img, h0, w0);
    compute_avoff(img, h0, w0);
    issue_a(IntC<0>(), 0, cbeg); issue_a(IntC<1>(), 0, cbeg); issue_a(IntC<2>(), 0, cbeg);
    issue_a(IntC<3>(), 0, cbeg); issue_a(IntC<4>(), 0, cbeg); issue_a(IntC<5>(), 0, cbeg);
    issue_a(IntC<6>(), 0, cbeg); issue_a(IntC<7>(), 0, cbeg);
    static_assert(APW <= 8, "prologue issues eight halo pieces per wave");
#pragma unroll
    for (int u = 0; u < DPH; ++u) issue_b(u, cbeg, u);
  }
  wait_vmcnt<0>();
  __syncthreads();
  if constexpr (XF) {   // the first patch (the table is visible since the barrier above)
    if ((int)blockIdx.x < a.ntiles) {
      xf_table(cbeg);
      xf_now(IntC<0>(), 0); xf_now(IntC<1>(), 0); xf_now(IntC<2>(), 0);
      xf_now(IntC<3>(), 0); xf_now(IntC<4>(), 0); xf_now(IntC<5>(), 0);
      xf_now(IntC<6>(), 0); xf_now(IntC<7>(), 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // ---- one phase ---------------------------------------------------------------------------------------------------------
  // c: slab of this tile, first: c == 0 (the previous tile's stores are among the young requests), last: c == ncb - 1,
  // has_next: another tile follows.  INIT: first phase of a tile (the MFMAs start from the bias).
  auto phase = [&](auto pc, auto initc, int c, bool first, bool last, bool has_next) __attribute__((always_inline)) {
    constexpr int p = decltype(pc)::value;
    constexpr bool INIT = decltype(initc)::value != 0;
    if (grp == 1 && !(UZ_PP_SKEL & 128)) __builtin_amdgcn_s_barrier();
    // ---- read phase (the other group computes) ----
    f32x4 cinit[CT];
    if constexpr (INIT) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) cinit[ct] = *reinterpret_cast<const f32x4*>(sBias + wn * 64 + ct * 16 + 4 * lq);
    }
    if (!(UZ_PP_SKEL & 1) && (!(UZ_PP_SKEL & 16) || INIT)) {
      const unsigned aoff = (unsigned)(apar * A_BYTES), boff = (unsigned)(bslot * SLOT_BYTES);
      const unsigned vb = b_base + boff;
      auto read_unit = [&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k < UPP) {
          constexpr int t = UPP * p + k, ty = t / 3, tx = t - 3 * ty;
          const unsigned va = a_base[tx] + aoff;
          auto rd_a = [&](auto ptc) __attribute__((always_inline)) {
            constexpr int pt = decltype(ptc)::value;
            if constexpr (pt < PT) lds_read16<(pt / HALVES + ty) * RB + (pt % HALVES) * 16 * PHP * RB>(fa[k][pt], va);
          };
          rd_a(IntC<0>()); rd_a(IntC<1>());
          lds_read16<k * B_UNIT + 0 * 16 * RB>(fb[k][0], vb);
          lds_read16<k * B_UNIT + 1 * 16 * RB>(fb[k][1], vb);
          lds_read16<k * B_UNIT + 2 * 16 * RB>(fb[k][2], vb);
          lds_read16<k * B_UNIT + 3 * 16 * RB>(fb[k][3], vb);
          rd_a(IntC<2>()); rd_a(IntC<3>()); rd_a(IntC<4>()); rd_a(IntC<5>()); rd_a(IntC<6>()); rd_a(IntC<7>());
        }
      };
      read_unit(IntC<0>()); read_unit(IntC<1>()); read_unit(IntC<2>());
    }
    const bool nonext = last && !has_next;
    if constexpr (XF && C::xf_here(p)) {
      // The pieces this wave requested XD phases ago are STAGED here (one ds_read each), between the fragment reads and
      // this phase's requests, so that they return while those are issued; their arithmetic rides in the gaps of this
      // wave's own MFMAs below.  They must have landed: nothing of this phase is in flight yet, so that is vmcnt(0) --
      // which asks of the weight pieces of phase p + 1 only what the end of this read phase asks anyway (and of the
      // previous tile's stores that they are done, one and a half phases after they were issued).
      // NO branch around the reads (not even for a dummy piece or the last slab of the last tile, which stage bytes nobody
      // writes back): to hipcc an asm load's register is written when the statement ends, and at the join of a branch it
      // COPIED the staged registers before the data had arrived -- one tile in a few hundred came out untransformed.
      constexpr int q = p - C::XD, k0 = C::na_before(q), kn = C::na(q);
      wait_vmcnt<0>();
      if constexpr (q == 0 || C::na(q > 0 ? q - 1 : 0) == 0) xf_table(nonext ? cbeg : (last ? cbeg : c + 1));   // the slab's first pieces: its table
      auto stage = [&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i < kn) {
          const unsigned ad = xf_addr(IntC<k0 + i>(), apar ^ 1);
          if (!(UZ_XF_SKEL & 2)) lds_read16u<0>(xd[i], ad);
          // where the result goes: back in place -- or, for a lane whose row is zero padding (it keeps the zeros the
          // LDS-DMA put there), a dummy piece, the slab after the last: into the 1 KB that swallows the dummy pieces.
          // An address instead of a branch: the write-back sits in the MFMA stream, where an exec-masked block cost more
          // than the arithmetic.
          xad[i] = (!nonext && xf_real(k0 + i) && avoff[k0 + i] != OOB) ? ad : smem_u + (unsigned)OFF_SCR + (unsigned)(lane << 4);
        }
      };
      stage(IntC<0>()); stage(IntC<1>()); stage(IntC<2>()); stage(IntC<3>());
    }
    if (!(UZ_PP_SKEL & 4)) {
      // weight pieces of phase p + DPH
      constexpr int pn = (p + DPH) % NPH;
      constexpr bool wrap = p + DPH >= NPH;
      int sl = bslot + DPH;
      sl = sl >= NSLOT ? sl - NSLOT : sl;
      if (!wrap) issue_b(sl, c, pn);
      else if (!last) issue_b(sl, c + 1, pn);
      else if (has_next) issue_b(sl, cbeg, pn);
      // halo pieces of the next slab (of this tile, or slab 0 of the next tile: avoff then holds that tile's offsets)
      if (!nonext) {
        constexpr int k0 = C::na_before(p), kn = C::na(p);
        const int cs = last ? cbeg : c + 1;
        if constexpr (kn > 0) issue_a(IntC<k0>(), apar ^ 1, cs);
        if constexpr (kn > 1) issue_a(IntC<k0 + 1>(), apar ^ 1, cs);
        if constexpr (kn > 2) issue_a(IntC<k0 + 2>(), apar ^ 1, cs);
        if constexpr (kn > 3) issue_a(IntC<k0 + 3>(), apar ^ 1, cs);
        static_assert(kn <= 4, "at most four halo pieces per wave and phase");
      }
    }
    {
      // XF: the halo pieces this wave requested XD phases ago are transformed below: they must have landed (and with them
      // everything older, the previous tile's stores included)
      constexpr bool TX = XF && C::xf_here(p);
      auto waits = [&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value;
        constexpr int NW0 = C::wait_normal(p, g), NN = C::wait_nonext(p, g), NX = C::wait_xf(p, g);
        constexpr int NW = (TX && NX < NW0) ? NX : NW0;
        if (nonext) {
          if (p <= DPH - 2 && first) wait_vmcnt<NN + NSTORE>();
          else wait_vmcnt<NN>();
        } else {
          if (p <= DPH - 2 && first && !(TX && NX < NW0 + NSTORE)) wait_vmcnt<NW0 + NSTORE>();
          else wait_vmcnt<NW>();
        }
      };
      if (C::nb(0) == C::nb(1) || grp == 0) waits(IntC<0>());
      else waits(IntC<1>());
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- compute phase ----
#pragma unroll
    for (int k = 0; k < UPP; ++k) {
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) pin16(fa[k][pt]);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) pin16(fb[k][ct]);
    }
    constexpr bool TXC = XF && C::xf_here(p);
    u32x4 xo[4];
    if constexpr (TXC) {
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(xd[i]));
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(xt[i]));
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!(UZ_PP_SKEL & 2)) {
      __builtin_amdgcn_s_setprio(1);
      if constexpr (!TXC) {
#pragma unroll
        for (int k = 0; k < UPP; ++k)
#pragma unroll
          for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
              const bf16x8 wv = *reinterpret_cast<const bf16x8*>(&fb[k][ct]);
              const bf16x8 xv = *reinterpret_cast<const bf16x8*>(&fa[k][pt]);
              acc[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, xv, (INIT && k == 0) ? cinit[ct] : acc[pt][ct], 0, 0, 0);
            }
      } else {
        // The staged pieces' arithmetic, two vector instructions behind every MFMA (an MFMA holds the SIMD's vector issue
        // for 8 of its 16 cycles: two per gap are nearly free, MI355X_MICROARCH.md) -- placed by hand, one micro-step
        // (unpack | fma | round + relu of one dword = two channels) per gap with a scheduling barrier behind it: left to
        // sched_group_barrier the scheduler put most of it behind the last MFMA.
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        constexpr int q = p - C::XD, k0 = C::na_before(q), kn = C::na(q);
        constexpr int NMF = UPP * PT * CT;
        static_assert(3 * 4 * kn <= NMF, "one micro-step per MFMA gap");
        float t0[16], t1[16];
#pragma unroll
        for (int g = 0; g < NMF; ++g) {
          const int k = g / (PT * CT), pt = (g / CT) % PT, ct = g % CT;
          const bf16x8 wv = *reinterpret_cast<const bf16x8*>(&fb[k][ct]);
          const bf16x8 xv = *reinterpret_cast<const bf16x8*>(&fa[k][pt]);
          acc[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, xv, (INIT && k == 0) ? cinit[ct] : acc[pt][ct], 0, 0, 0);
          if ((UZ_XF_SKEL & 1) && g < kn) xo[g] = xd[g];
          if (!(UZ_XF_SKEL & 3) && g < 3 * 4 * kn) {
            const int u = g / 3, st = g % 3, pc = u >> 2, i = u & 3;
            if (st == 0) {
              t0[u] = u2f(xd[pc][i] << 16);
              t1[u] = u2f(xd[pc][i] & 0xffff0000u);
            } else if (st == 1) {
              t0[u] = fma1(t0[u], u2f(xt[i >> 1][2 * (i & 1)]), u2f(xt[2 + (i >> 1)][2 * (i & 1)]));
              t1[u] = fma1(t1[u], u2f(xt[i >> 1][2 * (i & 1) + 1]), u2f(xt[2 + (i >> 1)][2 * (i & 1) + 1]));
            } else {
              const bf16x2 bb = __builtin_convertvector(f32x2{t0[u], t1[u]}, bf16x2);
              xo[pc][i] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, bb), s16x2{0, 0}));
            }
          }
          // a finished piece goes back at once (one 16-byte write: as four dword writes, bank-conflicted four ways, the launch
          // took 115 us instead of 111), overlapping the MFMAs that remain
          if (!(UZ_XF_SKEL & 2) && !(UZ_XF_SKEL & 4) && g >= 12 && g % 12 == 0 && g / 12 - 1 < kn) lds_write16u(xad[g / 12 - 1], xo[g / 12 - 1]);
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (12 * kn >= NMF) {
          if (!(UZ_XF_SKEL & 2) && !(UZ_XF_SKEL & 4)) lds_write16u(xad[kn - 1], xo[kn - 1]);
        }
        if (UZ_XF_SKEL & 4) {   // (measurement: every write-back behind the last MFMA)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (i < kn) lds_write16u(xad[i], xo[i]);
        }
      }
      __builtin_amdgcn_s_setprio(0);
      if constexpr (TXC) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (INIT) {
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[pt][ct] = cinit[ct];
    }
    __builtin_amdgcn_sched_barrier(0);
    bslot = bslot + 1 == NSLOT ? 0 : bslot + 1;
    // the barrier after a compute phase is the partner's barrier before its next read phase: group 0 executes it here,
    // group 1 at the top of the next phase; after a tile's last phase group 0 still executes it (group 1's matching one
    // opens the next tile), so both groups meet the same number of barriers per tile
    if (grp == 0 && !(UZ_PP_SKEL & 128)) __builtin_amdgcn_s_barrier();
  };

  // ---- wave-local epilogue ---------------------------------------------------------------------------------------------
  auto epilogue = [&](int im, int hh0, int ww0) {
    if (UZ_PP_SKEL & 8) {
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) asm volatile("" ::"v"(acc[pt][0]), "v"(acc[pt][1]), "v"(acc[pt][2]), "v"(acc[pt][3]));
      return;
    }
    if constexpr (SPLIT) {
      // raw fp32 partial tile: a lane holds 4 consecutive channels of pixel (patch row pt / HALVES, column 16 (pt % HALVES)
      // + lane & 15) per accumulator -> one 16-byte store each; everything has left before the next tile's counted waits
      int ls = lane;
      asm volatile("" : "+v"(ls));
      float* const pz = a.part + (size_t)blockIdx.z * ((size_t)a.N * a.H * a.W) * a.Nout;
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        const int hh = hh0 + ROWS_W * wm + pt / HALVES, ww = ww0 + (pt % HALVES) * 16 + (ls & 15);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const int ch = n0 + wn * 64 + 16 * ct + 4 * (ls >> 4);
          if (hh < a.H && ww < a.W && ch < a.Nout)
            *reinterpret_cast<f32x4*>(pz + ((size_t)(im * a.H + hh) * a.W + ww) * a.Nout + ch) = acc[pt][ct];
        }
      }
      wait_vmcnt<0>();
      return;
    }
    char* const stg = smem + OFF_STG + wave * STG_W;
    int ln = lane;   // opaque: the address arithmetic of the epilogue must not live in registers through the main loop
    asm volatile("" : "+v"(ln));
    const int e15 = ln & 15, eq = ln >> 4;
    const int cc = ln & 7;
    const int nch = n0 + wn * 64 + cc * VEC;
    // this lane's running sums live in the staging strip between epilogues (16 registers the main loop needs)
    float sq1[VEC], sq2[VEC];
    {
      const f32x4* sp = reinterpret_cast<const f32x4*>(stg + ln * 64);
      const f32x4 s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sq1[e] = s0[e];
        sq1[4 + e] = s1[e];
        sq2[e] = s2[e];
        sq2[4 + e] = s3[e];
      }
    }
    float bsc[VEC], bsh[VEC], bmu[VEC], bis[VEC];
    if constexpr (BNRED) {
      const int ch0 = nch < a.Nout ? nch : 0;
#pragma unroll
      for (int e = 0; e < VEC; e += 4) {
        *reinterpret_cast<f32x4*>(&bsc[e]) = *reinterpret_cast<const f32x4*>(a.bn_scale + ch0 + e);
        *reinterpret_cast<f32x4*>(&bsh[e]) = *reinterpret_cast<const f32x4*>(a.bn_shift + ch0 + e);
        *reinterpret_cast<f32x4*>(&bmu[e]) = *reinterpret_cast<const f32x4*>(a.bn_mean + ch0 + e);
        *reinterpret_cast<f32x4*>(&bis[e]) = *reinterpret_cast<const f32x4*>(a.bn_invstd + ch0 + e);
      }
    }
    // a round = two 16-pixel tiles = 32 pixels: one patch row (TW 32) or two (TW 16)
#pragma unroll
    for (int r = 0; r < PT / 2; ++r) {
      // pixel q = 0 .. 31 of the round
      auto pix_h = [&](int q) { return hh0 + ROWS_W * wm + (TW == 32 ? r : 2 * r + (q >> 4)); };
      auto pix_w = [&](int q) { return ww0 + (TW == 32 ? q : (q & 15)); };
      Vec16<T> yb[4];
      if constexpr (BNRED) {
        const T* __restrict__ by = static_cast<const T*>(a.bn_y);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int q = (ln >> 3) + 8 * k;
          const int hc = min(pix_h(q), a.H - 1), wc = min(pix_w(q), a.W - 1);   // clamped: unused outside the image
          yb[k] = ld16(by + ((size_t)(im * a.H + hc) * a.W + wc) * a.ld_bny + (nch < a.Nout ? nch : 0));
        }
      }
      // stage: a lane holds 4 consecutive channels of one pixel per accumulator -> one 8-byte LDS write
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          bf16x4 pk;
#pragma unroll
          for (int e = 0; e < 4; ++e) pk[e] = (bf16_t)acc[2 * r + h][ct][e];
          *reinterpret_cast<bf16x4*>(stg + (16 * h + e15) * STG_ROW + (16 * ct + 4 * eq) * ES) = pk;
        }
      // read back: lane = (pixel (lane >> 3) + 8 k, channel chunk lane & 7); LDS operations of one wave execute in order
      Vec16<T> vb[4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        vb[k] = *reinterpret_cast<const Vec16<T>*>(stg + ((ln >> 3) + 8 * k) * STG_ROW + cc * 16);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int q = (ln >> 3) + 8 * k;
        const int hh = pix_h(q), ww = pix_w(q);
        const bool inside = hh < a.H && ww < a.W && nch < a.Nout;
        // buffer stores executed by every lane (outside the image: out of range): exactly NSTORE vector-memory
        // operations per wave and tile, which the counted waits of the next tile's first phases allow for
        const unsigned off = inside ? (unsigned)((((im * a.H + hh) * a.W + ww) * a.ldy + nch) * ES) : OOB;
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(&vb[k]), yr, off, 0, 0);
        if (inside) {
          if constexpr (BNRED) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
              const float yv = (float)yb[k].v[e];
              const float dz = fmaf(yv, bsc[e], bsh[e]) > 0.f ? (float)vb[k].v[e] : 0.f;
              sq1[e] += dz;
              sq2[e] += dz * ((yv - bmu[e]) * bis[e]);
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
    }
    {
      f32x4* sp = reinterpret_cast<f32x4*>(stg + ln * 64);
      sp[0] = f32x4{sq1[0], sq1[1], sq1[2], sq1[3]};
      sp[1] = f32x4{sq1[4], sq1[5], sq1[6], sq1[7]};
      sp[2] = f32x4{sq2[0], sq2[1], sq2[2], sq2[3]};
      sp[3] = f32x4{sq2[4], sq2[5], sq2[6], sq2[7]};
    }
  };

  // ---- the tiles of this workgroup ---------------------------------------------------------------------------------------
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int next = tile + gridDim.x;
    const bool has_next = next < a.ntiles;
    if (has_next) decode(next, nim, nh0, nw0);
#pragma unroll 1
    for (int c = cbeg; c < cend; ++c) {
      const bool first = c == cbeg, last = c == cend - 1;
      // from the last slab on, the halo requests are those of the next tile's first patch
      if (last && has_next) compute_avoff(nim, nh0, nw0);
      if (first) phase(IntC<0>(), IntC<1>(), c, first, last, has_next);
      else phase(IntC<0>(), IntC<0>(), c, first, last, has_next);
      phase(IntC<1>(), IntC<0>(), c, first, last, has_next);
      phase(IntC<2>(), IntC<0>(), c, first, last, has_next);
      if constexpr (NPH == 9) {
        phase(IntC<3>(), IntC<0>(), c, first, last, has_next);
        phase(IntC<4>(), IntC<0>(), c, first, last, has_next);
        phase(IntC<5>(), IntC<0>(), c, first, last, has_next);
        phase(IntC<6>(), IntC<0>(), c, first, last, has_next);
        phase(IntC<7>(), IntC<0>(), c, first, last, has_next);
        phase(IntC<8>(), IntC<0>(), c, first, last, has_next);
      }
      static_assert(NPH == 9 || NPH == 3, "phases per slab");
      apar ^= 1;
    }
    if constexpr (BNRED) wait_vmcnt<0>();   // (the epilogue's own loads would make the compiler wait for everything anyway)
    epilogue(img, h0, w0);
    img = nim;
    h0 = nh0;
    w0 = nw0;
  }

  // ---- statistics: fixed-order sum over the lanes that own a channel chunk ---------------------------------------------------
  if (!SPLIT && a.stats != nullptr) {
    wait_vmcnt<0>();
    __syncthreads();
    // thread `th` left its sums [2][VEC] at strip(th >> 6) + (th & 63) * 64
    auto red = [&](int th, int idx) { return reinterpret_cast<const float*>(smem + OFF_STG + (th >> 6) * STG_W + (th & 63) * 64)[idx]; };
    if (tid < 2 * BN) {
      const int which = tid / BN, ch = tid - which * BN;
      const int cwn = ch >> 6, cc = (ch & 63) >> 3, e = ch & 7;
      float t = 0.f;
      for (int m = 0; m < C::WM; ++m)
        for (int l = 0; l < 8; ++l) {
          const int w = C::WN == 2 ? 2 * m + cwn : m;
          t += red((w << 6) + l * 8 + cc, which * VEC + e);
        }
      if (n0 + ch < a.Nout) a.stats[((size_t)blockIdx.x * 2 + which) * a.Nout + n0 + ch] = t;
    }
  }
}

typedef PpCfg<16, 32, 8, 1, 3> Cfg512x64;
typedef PpCfg<16, 32, 8, 1, 3, true> Cfg512x64X;   // the XF instantiation's deal of the halo pieces
typedef PpCfg<8, 32, 4, 2, 3> Cfg256;
typedef PpCfg<16, 16, 4, 2, 3> Cfg256w16;
typedef PpCfg<8, 16, 4, 2, 3> Cfg128w16;

}  // namespace

// ---- host side --------------------------------------------------------------------------------------------------------
// returns 1 and fills the plan when a ping-pong configuration takes this descriptor
int uz_pp_plan(const uz_conv_desc* d, UzPpPlan* p) {
  const bool up = d->taps_mode == UZ_TAPS_CONV_UP2;
  if (d->dtype != UZ_BF16) return 0;
  if (!(d->taps_mode == UZ_TAPS_CONV || up) || d->ntaps != 9 || d->dil != 1 || d->store_mode != UZ_STORE_PLAIN) return 0;
  if (up && ((d->H & 1) || (d->W & 1) || d->Hin * 2 != d->H || d->Win * 2 != d->W)) return 0;
  if (d->Cin % KU != 0 || d->Nout % 8 != 0 || d->ldy % 8 != 0 || d->ldx % 8 != 0) return 0;
  const long long xbytes = ((long long)d->N * d->Hin * d->Win - 1) * d->ldx * 2 + (long long)d->Cin * 2;
  const long long wbytes = (long long)d->Nout * 9 * d->Cin * 2;
  const long long ybytes = ((long long)d->N * d->H * d->W - 1) * d->ldy * 2 + (long long)d->Nout * 2;
  if (xbytes >= (1LL << 31) || wbytes >= (1LL << 31) || ybytes >= (1LL << 31)) return 0;
  if (uz_tune_flags() & 0x2000000) return 0;   // ablation build: keep the second-generation kernels
  auto tiles = [&](int th, int tw) { return (long long)d->N * ((d->H + th - 1) / th) * ((d->W + tw - 1) / tw); };
  auto rounds = [&](long long t) { return (t + UZ_NUM_CU - 1) / UZ_NUM_CU; };
  int cfg = -1;
  if (d->Nout >= 128) {
    const int tn = (d->Nout + 127) / 128;
    if (d->W >= 32 && d->H >= 8) {
      // 512-pixel tiles when they still fill the chip: a round of them counted at 2 x 0.8 of a 256-pixel round (twice
      // the pixels, ~20 % more throughput: fewer DMA bytes and fragment reads per MFMA)
      const long long r512 = rounds(tiles(16, 32) * tn), r256 = rounds(tiles(8, 32) * tn);
      cfg = (d->H >= 16 && (r512 * 16 < r256 * 10 || (uz_tune_flags() & 0x1000000))) ? UZ_PP_512 : UZ_PP_256;
      if (cfg == UZ_PP_256 && (uz_tune_flags() & 0x8000000)) cfg = -1;   // ablation build: 256-pixel tiles on the old kernels
    } else if (d->W >= 9 && d->W <= 16 && d->H >= 8) {
      cfg = UZ_PP_256W16;
      if (uz_tune_flags() & 0x8000000) cfg = -1;
    }
  } else if (d->Nout > 32 && d->W >= 32 && d->H >= 16) {
    // 64 output channels: 512-pixel tiles; small problems stay on the second-generation kernels
    if (tiles(16, 32) >= UZ_NUM_CU / 2 && !(uz_tune_flags() & 0x10000000)) cfg = UZ_PP_512X64;
  }
  if (cfg < 0) return 0;
  // 128-pixel tiles (8 x 16) where 256-pixel ones leave at least half of the CUs without a tile and the split-K plan below
  // (a quarter) does not apply: unet's 16 x 16 maps at B = 16, 512 -> 1024 50.1 -> 41.8 us, 1024 -> 1024 92.2 -> 76.2 us
  // (bit-identical: an output's K order does not depend on its tile; profiles/r05_pp128_probe.txt).  Twice the weight
  // fragments per MFMA -- which idle CUs pay for, full ones would not.
  if ((cfg == UZ_PP_256 || cfg == UZ_PP_256W16) && !(uz_tune_flags() & 0x4)) {
    const long long tot = (cfg == UZ_PP_256 ? tiles(8, 32) : tiles(16, 16)) * ((d->Nout + 127) / 128);
    const bool splits = tot * 4 <= UZ_NUM_CU_HW && d->Cin / KU >= 16;
    if (tot * 2 <= UZ_NUM_CU_HW && !splits) cfg = UZ_PP_128W16;
  }
  const int th = (cfg == UZ_PP_256 || cfg == UZ_PP_128W16) ? 8 : 16, tw = (cfg == UZ_PP_256W16 || cfg == UZ_PP_128W16) ? 16 : 32, bn = cfg == UZ_PP_512X64 ? 64 : 128;
  p->cfg = cfg;
  p->bn = bn;
  p->th_n = (d->H + th - 1) / th;
  p->tw_n = (d->W + tw - 1) / tw;
  p->ntiles = d->N * p->th_n * p->tw_n;
  p->tiles_n = (d->Nout + bn - 1) / bn;
  int cap = UZ_NUM_CU / p->tiles_n;
  if (cap < 1) cap = 1;
  p->grid_m = p->ntiles < cap ? p->ntiles : cap;
  // split-K: the 16 x 16 / 32 x 32 maps of a 256 x 256 input have 32 ... 128 tiles and 16 ... 32 channel slabs of nine taps
  // each -- a quarter to a half of the chip walks a long K loop.  With a workspace (uz_conv_igemm_ws) the slabs are dealt
  // to `ksplit` workgroups per tile.  Only where at least FOUR ranges fit (tiles <= a quarter of the CUs): measured on
  // unet's 16 x 16 layers at B = 16, 1024 -> 512 (64 tiles, 4 ranges) 78.0 -> 47.1 us, but with two ranges the fp32 partial
  // tiles and the reduce pass cost what the shorter K loop saves (512 -> 1024: 43.8 -> 47.8 us, 1024 -> 1024: 80.6 -> 72.3).
  // Sized by the hardware's CU count, not by a CU reserve: the summation order, hence the rounding of the result, must
  // not depend on that setting.
  p->ksplit = 1;
  p->cps = d->Cin / KU;
  {
    const long long tot = (long long)p->ntiles * p->tiles_n;
    const int ncb = d->Cin / KU;
    if ((cfg == UZ_PP_256 || cfg == UZ_PP_256W16) && tot * 4 <= UZ_NUM_CU_HW && ncb >= 16 && d->Nout % 8 == 0 &&
        !(uz_tune_flags() & 0x40000000)) {
      long long s = UZ_NUM_CU_HW / tot;
      if (s > ncb / 4) s = ncb / 4;
      if (s > 8) s = 8;
      if (s >= 4) {
        p->cps = (int)((ncb + s - 1) / s);
        p->ksplit = (ncb + p->cps - 1) / p->cps;
      }
    }
  }
  return 1;
}

// input channels whose (scale, shift) table fits beside a configuration's LDS image (uz_conv_igemm_xf); 0: no XF form
int uz_pp_xf_channels(const UzPpPlan& p) {
  switch (p.cfg) {
    case UZ_PP_512X64: return Cfg512x64X::XF_CH;
    default: return 0;
  }
}

int uz_pp_launch(const uz_conv_desc* d, const UzPpPlan& p, const void* x, const void* w, const float* bias, void* y,
                 float* stats, hipStream_t s, const UzBnRed* br, float* part, const UzXf* xf) {
  PpArgs a;
  a.part = part;
  a.cps = p.cps;
  a.xf_scale = xf ? xf->scale : nullptr;
  a.xf_shift = xf ? xf->shift : nullptr;
  if (xf)
    UZ_REQUIRE(br == nullptr && part == nullptr && xf->scale && xf->shift && d->Cin <= uz_pp_xf_channels(p),
               "uz_conv_igemm_xf(direct3x3 ping-pong): configuration %d takes no input transform for %d channels", p.cfg, d->Cin);
  UZ_REQUIRE(part == nullptr || (br == nullptr && p.ksplit > 1 && (p.cfg == UZ_PP_256 || p.cfg == UZ_PP_256W16)),
             "uz_conv_igemm(direct3x3 ping-pong): split-K launch without a split plan");
  a.x = x;
  a.w = w;
  a.y = y;
  a.bias = bias;
  a.stats = stats;
  a.xbytes = (unsigned)(((long long)d->N * d->Hin * d->Win - 1) * d->ldx * 2 + (long long)d->Cin * 2);
  a.wbytes = (unsigned)((long long)d->Nout * 9 * d->Cin * 2);
  a.ybytes = (unsigned)(((long long)d->N * d->H * d->W - 1) * d->ldy * 2 + (long long)d->Nout * 2);
  a.N = d->N;
  a.H = d->H;
  a.W = d->W;
  a.Cin = d->Cin;
  a.ldx = d->ldx;
  a.Nout = d->Nout;
  a.ldy = d->ldy;
  a.K = 9 * d->Cin;
  a.th_n = p.th_n;
  a.tw_n = p.tw_n;
  a.ntiles = p.ntiles;
  a.ups = d->taps_mode == UZ_TAPS_CONV_UP2 ? 1 : 0;
  a.bn_y = br ? br->y : nullptr;
  a.bn_scale = br ? br->scale : nullptr;
  a.bn_shift = br ? br->shift : nullptr;
  a.bn_mean = br ? br->mean : nullptr;
  a.bn_invstd = br ? br->invstd : nullptr;
  a.ld_bny = br ? br->ldy : 0;
  if (br) UZ_REQUIRE(stats != nullptr, "uz_conv_igemm_bnred: partial rows missing");
  if (part != nullptr) {   // every tile has its own ksplit workgroups (grid_m = ntiles <= 128 here)
    dim3 gs(p.ntiles, p.tiles_n, p.ksplit);
    if (p.cfg == UZ_PP_256) hipLaunchKernelGGL((conv3x3_pp_kernel<Cfg256, false, true>), gs, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((conv3x3_pp_kernel<Cfg256w16, false, true>), gs, dim3(512), 0, s, a);
    UZ_LAUNCH_CHECK("uz_conv_igemm(direct3x3 ping-pong, split-K)");
    return UZ_OK;
  }
  dim3 grid(p.grid_m, p.tiles_n), block(512);
  if (xf) {
    hipLaunchKernelGGL((conv3x3_pp_kernel<Cfg512x64X, false, false, true>), grid, block, 0, s, a);
    UZ_LAUNCH_CHECK("uz_conv_igemm_xf(direct3x3 ping-pong)");
    return UZ_OK;
  }
#define UZ_PP_GO(CFG)                                                                          \
  do {                                                                                         \
    if (br) hipLaunchKernelGGL((conv3x3_pp_kernel<CFG, true>), grid, block, 0, s, a);          \
    else hipLaunchKernelGGL((conv3x3_pp_kernel<CFG, false>), grid, block, 0, s, a);            \
  } while (0)
  switch (p.cfg) {
    case UZ_PP_512: UZ_PP_GO(Cfg512); break;
    case UZ_PP_512X64: UZ_PP_GO(Cfg512x64); break;
    case UZ_PP_256: UZ_PP_GO(Cfg256); break;
    case UZ_PP_256W16: UZ_PP_GO(Cfg256w16); break;
    case UZ_PP_128W16: UZ_PP_GO(Cfg128w16); break;
    default: UZ_REQUIRE(false, "uz_conv_igemm(direct3x3 ping-pong): bad configuration %d", p.cfg);
  }
#undef UZ_PP_GO
  UZ_LAUNCH_CHECK("uz_conv_igemm(direct3x3 ping-pong)");
  return UZ_OK;
}
