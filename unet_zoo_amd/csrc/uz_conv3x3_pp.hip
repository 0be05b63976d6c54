// Direct 3x3 convolution, third generation ("ping-pong"), bf16, gfx950 (MI355X), NHWC.
//
// Stands in for nn.Conv2d(k=3, padding=1) forward and its input gradient on the layers with at least 128 output
// channels and enough pixels to give every CU a 512-pixel tile (reference: unet_zoo/models/common_layers.py:28,31,47,52,71;
// autograd a19).  Same arithmetic as uz_conv3x3.hip (bf16 operands, fp32 accumulation on v_mfma_f32_32x32x16_bf16,
// result rounded once to bf16, BatchNorm sums of the STORED values), a different schedule:
//
//  * Workgroup tile 512 pixels (16 x 32 patch) x 128 output channels, 8 waves as 4 (pixels) x 2 (channels), wave tile
//    128 pixels x 64 channels: 6 fragment reads per 8 MFMAs (the 64 x 64 wave tile of the second generation: 8 per 8)
//    and 24.5 KB of LDS-DMA per 64-deep K step of 256 MFMAs (there: 20.7 KB per 128 MFMAs).
//  * K unit = (tap, 32 input channels).  The halo patch of a 32-channel slab is 34 x 19 rows of 64 bytes (column-major
//    with an odd column height, so that a tap and an M tile are immediate offsets of the ds_read and the XOR swizzle key
//    depends on the patch column only); two patch buffers, five weight slots of [128][64 B].
//  * PING-PONG: the two waves that share a SIMD (w, w + 4) never compute at the same time.  Between two s_barriers one
//    group issues its 16 MFMAs of a unit while the other reads the 12 fragments of its next unit from LDS and issues its
//    LDS-DMA pieces (one weight piece per unit, one halo piece in six of nine units), then the roles swap.  The matrix
//    pipe of a SIMD sees one back-to-back MFMA stream; fragment reads, address arithmetic, DMA issue and the counted
//    s_waitcnt vmcnt(N) all sit in the other wave's half.
//  * The unit stream runs across the tiles of a workgroup: the last slab of a tile requests the first patch and the first
//    weight tiles of the next one.  The epilogue is wave-local (no barrier): 32-pixel rounds through a private 4.5 KB
//    staging strip, 16-byte buffer stores of full 128-byte lines, statistics from the stored values; both groups run it
//    in the same barrier interval (the group that finished first starts it while the other one computes its last unit).
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
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(3))) char* lds_char_ptr;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned OOB = 0x80000000u;   // beyond any descriptor's num_records (tensors < 2 GiB), also after adding a slab offset

constexpr int TH = 16, TW = 32, PH = TH + 2, PW = TW + 2, PHP = PH | 1;   // 18 x 34 halo patch, columns of 19 rows
constexpr int RB = 64;                                                   // bytes per LDS row = 32 channels
constexpr int KU = 32;                                                   // K per unit
constexpr int PROWS = PW * PHP;                                          // 646
constexpr int APIECES = (PROWS * RB + 1023) / 1024;                      // 41 pieces of 1 KB (16 rows)
constexpr int A_BYTES = APIECES * 1024;
constexpr int APW = (APIECES + 7) / 8;                                   // 6 pieces per wave and slab (the tail ones are dummies)
constexpr int BN = 128;
constexpr int B_SLOT = BN * RB;                                          // 8 KB = 8 pieces, one per wave
constexpr int NSLOT = 5, DPF = NSLOT - 1;                                // weight tiles are requested DPF units ahead
constexpr int OFF_B = 2 * A_BYTES;
constexpr int STG_ROW = 144;                                             // staging row: 64 channels + 16 bytes
constexpr int STG_W = 32 * STG_ROW;
constexpr int OFF_STG = OFF_B + NSLOT * B_SLOT;
constexpr int OFF_SCR = OFF_STG + 8 * STG_W;                             // 1 KB that swallows the dummy pieces
constexpr int OFF_BIAS = OFF_SCR + 1024;
constexpr int SMEM_BYTES = OFF_BIAS + BN * 4;
static_assert(SMEM_BYTES <= 160 * 1024, "LDS budget");
constexpr int NSTORE = 16;                                               // epilogue stores per wave and tile

// vmcnt(N) at the end of the read phase of the unit with tap t: everything this wave requested for unit + 1 has landed.
// A wave's request sequence per unit is [weight piece of unit + DPF][halo piece t of the next slab, t < APW]; the
// weight piece of unit + 1 was requested DPF - 1 units ago, so the requests that may stay in flight are the DPF - 1
// younger weight pieces and the halo pieces requested in units t - (DPF - 1) .. t; before a slab's first unit the whole
// patch of that slab must be in: nothing younger than the three weight pieces after the last halo piece.
constexpr int pp_wait_normal(int t) {
  int n = DPF - 1;
  for (int k = t - (DPF - 1); k <= t; ++k) {
    const int kk = (k + 9) % 9;
    if (kk < APW) ++n;
  }
  if (t == 8) n = 8 - APW + 1 < n ? 8 - APW + 1 : n;   // units APW .. 8 requested weight pieces only
  return n;
}
// last slab of a workgroup's last tile: no halo pieces, no weight pieces beyond the last unit
constexpr int pp_wait_nonext(int t) {
  int n = 0;
  for (int k = t - (DPF - 2); k <= t; ++k)
    if (k + DPF <= 8) ++n;     // (k < 0: a unit of the previous slab, which requested its weight piece)
  return n;
}
static_assert(pp_wait_normal(0) == 4 && pp_wait_normal(3) == 7 && pp_wait_normal(8) == 3, "wait table");
static_assert(pp_wait_nonext(0) == 3 && pp_wait_nonext(5) == 2 && pp_wait_nonext(7) == 0, "wait table");

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

#ifndef UZ_PP_SKEL
#define UZ_PP_SKEL 0   // measurement builds: 1 no fragment reads, 2 no MFMAs, 4 no DMA after the prologue, 8 no epilogue,
                      // 16 fragment reads in a tile's first unit only, 32 no s_setprio, 64 no lgkmcnt wait before the barrier
                      // (after it instead), 128 both groups in lockstep (timing only)
#endif

// S16: v_mfma_f32_16x16x32_bf16 (a unit = 32 MFMAs of 16 cycles) instead of 32x32x16 (16 of 32 cycles): same fragment
// bytes and accumulator count, half the accumulator traffic per flop; the chip holds a higher clock on that stream.
template <bool BNRED, bool S16>
__global__ __launch_bounds__(512, 2) void conv3x3_pp_kernel(const PpArgs a) {
  typedef bf16_t T;
  constexpr int ES = 2, VEC = 8;
  __shared__ __attribute__((aligned(1024))) char smem[SMEM_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                  // ping-pong group: waves w and w + 4 share a SIMD
  const int wm = wave >> 1, wn = wave & 1;    // wave tile: patch rows 4 wm .. 4 wm + 3, channels 64 wn .. 64 wn + 63
  const int l31 = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.y * BN;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.wbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.ybytes, 0x00020000);
  const unsigned smem_u = (unsigned)(size_t)(lds_char_ptr)smem;

  // ---- fragment read addresses -------------------------------------------------------------------------------------
  // patch pixel (pi, pj) lives in LDS row pj * PHP + pi; weight rows are output channels.  A row's four 16-byte chunks
  // are permuted with a key of the patch column / channel (swz()), the same on the DMA source side and on the read.
  // 32x32x16: a lane reads pixel column l31 + tx, K chunk 2 q + lh: one base per (tx, q); the patch row 4 wm + i + ty
  // is an immediate offset.  16x16x32: a lane reads pixel column (lane & 15) + 16 h + tx, K chunk lane >> 4: one base
  // per tx; patch row and column half h are immediate offsets (16 columns further the key is the same).
  // XOR with (col >> 2) & 3 is conflict-free for the 32-row reads at every tap shift; the 16-row reads (two K chunks per
  // 16-lane group) need the rotation by 2 * (col >> 2).
  auto swz = [](int chunk, int col) { return S16 ? ((chunk + 2 * (col >> 2)) & 3) : (chunk ^ ((col >> 2) & 3)); };
  auto unswz = [](int phys, int col) { return S16 ? ((phys - 2 * (col >> 2)) & 3) : (phys ^ ((col >> 2) & 3)); };
  const int l15 = lane & 15, lq = lane >> 4;
  unsigned a_base[3][2], b_base[2];
#pragma unroll
  for (int tx = 0; tx < 3; ++tx)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if constexpr (S16) {
        const int pj = l15 + tx;
        a_base[tx][q] = smem_u + (pj * PHP + 4 * wm) * RB + (swz(lq, pj) << 4);
      } else {
        const int pj = l31 + tx;
        a_base[tx][q] = smem_u + (pj * PHP + 4 * wm) * RB + (swz(2 * q + lh, pj) << 4);
      }
    }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    if constexpr (S16) {
      const int brow = wn * 64 + l15;
      b_base[q] = smem_u + OFF_B + brow * RB + (swz(lq, brow) << 4);
    } else {
      const int brow = wn * 64 + l31;
      b_base[q] = smem_u + OFF_B + brow * RB + (swz(2 * q + lh, brow) << 4);
    }
  }
  // ---- LDS-DMA source offsets --------------------------------------------------------------------------------------
  // weight piece `wave` of a slot: rows 16 wave + (lane >> 2), this lane fetches the chunk that belongs at (lane & 3)
  unsigned bvoff;
  {
    const int brow = wave * 16 + (lane >> 2);
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
      const int r = (wave + 8 * k) * 16 + (ln >> 2);
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
    const int piece = wave + 8 * k;
    char* dst = piece < APIECES ? smem + buf * A_BYTES + piece * 1024 : smem + OFF_SCR;
    dma16(xr, dst, avoff[k], (unsigned)(cslab * KU * ES));
  };
  auto issue_b = [&](int slot, int cslab, int tap) {
    dma16(wr, smem + OFF_B + slot * B_SLOT + wave * 1024, bvoff, (unsigned)((tap * a.Cin + cslab * KU) * ES));
  };
  auto decode = [&](int tile, int& im, int& hh0, int& ww0) {
    const int per = a.th_n * a.tw_n;
    im = tile / per;
    const int rem = tile - im * per;
    const int ti = rem / a.tw_n;
    hh0 = ti * TH;
    ww0 = (rem - ti * a.tw_n) * TW;
  };

  // bias table (fp32, 128 channels of this workgroup): the accumulators' initial value
  float* const sBias = reinterpret_cast<float*>(smem + OFF_BIAS);
  if (tid < BN) sBias[tid] = (!BNRED && a.bias != nullptr && n0 + tid < a.Nout) ? a.bias[n0 + tid] : 0.f;

  const int ncb = a.Cin / KU;
  // 32x32x16: acc[i][j] = patch row 4 wm + i x channels 32 j (16 registers); fa[q][i], fb[q][j]
  // 16x16x32: accs[2 i + h][ct] = patch row i, column half h x channels 16 ct (4 registers); fa[h][i], fb[ct >> 1][ct & 1]
  f32x16 acc[4][2];
  f32x4 accs[S16 ? 8 : 1][4];
  f32x4 fa[2][4], fb[2][2];
  {   // running BatchNorm sums of this lane (channel chunk lane & 7 of the read-back phase): zero, in the staging strip
    f32x4* sp = reinterpret_cast<f32x4*>(smem + OFF_STG + wave * STG_W + lane * 64);
    sp[0] = sp[1] = sp[2] = sp[3] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  int apar = 0;    // patch buffer of the current slab
  int bslot = 0;   // weight slot of the current unit
  int img = 0, h0 = 0, w0 = 0, nim = 0, nh0 = 0, nw0 = 0;

  // ---- prologue: first patch, first DPF weight tiles ------------------------------------------------------------------
  if ((int)blockIdx.x < a.ntiles) {
    decode(blockIdx.x, img, h0, w0);
    compute_avoff(img, h0, w0);
    issue_a(IntC<0>(), 0, 0); issue_a(IntC<1>(), 0, 0); issue_a(IntC<2>(), 0, 0);
    issue_a(IntC<3>(), 0, 0); issue_a(IntC<4>(), 0, 0); issue_a(IntC<5>(), 0, 0);
#pragma unroll
    for (int u = 0; u < DPF; ++u) issue_b(u, 0, u);
  }
  wait_vmcnt<0>();
  __syncthreads();

  // ---- one unit ---------------------------------------------------------------------------------------------------------
  // c: slab of this tile, first: c == 0 (the previous tile's stores are among the young requests), last: c == ncb - 1,
  // has_next: another tile follows.  INIT: first unit of a tile (the MFMAs start from the bias).
  auto unit = [&](auto tc, auto initc, int c, bool first, bool last, bool has_next) __attribute__((always_inline)) {
    constexpr int t = decltype(tc)::value;
    constexpr bool INIT = decltype(initc)::value != 0;
    constexpr int ty = t / 3, tx = t - 3 * ty;
    if (grp == 1 && !(UZ_PP_SKEL & 128)) __builtin_amdgcn_s_barrier();
    // ---- read phase (the other group computes) ----
    f32x16 cinit[2];
    f32x4 cinit4[4];
    if constexpr (INIT) {
      if constexpr (S16) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) cinit4[ct] = *reinterpret_cast<const f32x4*>(sBias + wn * 64 + ct * 16 + 4 * lq);
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(sBias + wn * 64 + j * 32 + 8 * q + 4 * lh);
#pragma unroll
            for (int e = 0; e < 4; ++e) cinit[j][4 * q + e] = b4[e];
          }
      }
    }
    if (!(UZ_PP_SKEL & 1) && (!(UZ_PP_SKEL & 16) || INIT)) {
      const unsigned aoff = (unsigned)(apar * A_BYTES), boff = (unsigned)(bslot * B_SLOT);
      if constexpr (S16) {
        const unsigned va = a_base[tx][0] + aoff, vb = b_base[0] + boff;
        lds_read16<(0 + ty) * RB>(fa[0][0], va);
        lds_read16<(0 + ty) * RB + 16 * PHP * RB>(fa[1][0], va);
        lds_read16<0>(fb[0][0], vb);
        lds_read16<16 * RB>(fb[0][1], vb);
        lds_read16<32 * RB>(fb[1][0], vb);
        lds_read16<48 * RB>(fb[1][1], vb);
        lds_read16<(1 + ty) * RB>(fa[0][1], va);
        lds_read16<(1 + ty) * RB + 16 * PHP * RB>(fa[1][1], va);
        lds_read16<(2 + ty) * RB>(fa[0][2], va);
        lds_read16<(2 + ty) * RB + 16 * PHP * RB>(fa[1][2], va);
        lds_read16<(3 + ty) * RB>(fa[0][3], va);
        lds_read16<(3 + ty) * RB + 16 * PHP * RB>(fa[1][3], va);
      } else {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const unsigned va = a_base[tx][q] + aoff, vb = b_base[q] + boff;
          lds_read16<(0 + ty) * RB>(fa[q][0], va);
          lds_read16<(1 + ty) * RB>(fa[q][1], va);
          lds_read16<(2 + ty) * RB>(fa[q][2], va);
          lds_read16<(3 + ty) * RB>(fa[q][3], va);
          lds_read16<0>(fb[q][0], vb);
          lds_read16<32 * RB>(fb[q][1], vb);
        }
      }
    }
    const bool nonext = last && !has_next;
    if (!(UZ_PP_SKEL & 4)) {
      // weight piece of unit + DPF
      constexpr int tn = (t + DPF) % 9;
      constexpr bool wrap = t + DPF >= 9;
      int sl = bslot + DPF;
      sl = sl >= NSLOT ? sl - NSLOT : sl;
      if (!wrap) issue_b(sl, c, tn);
      else if (!last) issue_b(sl, c + 1, tn);
      else if (has_next) issue_b(sl, 0, tn);
      // halo piece t of the next slab (of this tile, or slab 0 of the next tile: avoff then holds that tile's offsets)
      if constexpr (t < APW) {
        if (!nonext) issue_a(IntC<t>(), apar ^ 1, last ? 0 : c + 1);
      }
    }
    {
      constexpr int NW = pp_wait_normal(t), NN = pp_wait_nonext(t);
      if (nonext) {
        if (t <= DPF - 2 && first) wait_vmcnt<NN + NSTORE>();
        else wait_vmcnt<NN>();
      } else {
        if (t <= DPF - 2 && first) wait_vmcnt<NW + NSTORE>();
        else wait_vmcnt<NW>();
      }
    }
    if (!(UZ_PP_SKEL & 64)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (UZ_PP_SKEL & 64) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // ---- compute phase ----
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int i = 0; i < 4; ++i) pin16(fa[q][i]);
#pragma unroll
      for (int j = 0; j < 2; ++j) pin16(fb[q][j]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!(UZ_PP_SKEL & 2)) {
      if (!(UZ_PP_SKEL & 32)) __builtin_amdgcn_s_setprio(1);
      if constexpr (S16) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
              const bf16x8 wv = *reinterpret_cast<const bf16x8*>(&fb[ct >> 1][ct & 1]);
              const bf16x8 xv = *reinterpret_cast<const bf16x8*>(&fa[h][i]);
              accs[2 * i + h][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, xv, INIT ? cinit4[ct] : accs[2 * i + h][ct], 0, 0, 0);
            }
      } else {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const bf16x8 wv = *reinterpret_cast<const bf16x8*>(&fb[q][j]);
              const bf16x8 xv = *reinterpret_cast<const bf16x8*>(&fa[q][i]);
              if (INIT && q == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, xv, cinit[j], 0, 0, 0);
              else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, xv, acc[i][j], 0, 0, 0);
            }
      }
      if (!(UZ_PP_SKEL & 32)) __builtin_amdgcn_s_setprio(0);
    } else if (INIT) {
      if constexpr (S16) {
#pragma unroll
        for (int pt = 0; pt < 8; ++pt)
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) accs[pt][ct] = cinit4[ct];
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = cinit[j];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    bslot = bslot + 1 == NSLOT ? 0 : bslot + 1;
  };
  // the barrier after a compute phase is the partner's barrier before its next read phase: group 0 executes it here,
  // group 1 at the top of unit(); after a tile's last unit group 0 still executes it (group 1's matching one opens the
  // next tile), so both groups meet 18 * ncb barriers per tile
  auto tail_barrier = [&]() {
    if (grp == 0 && !(UZ_PP_SKEL & 128)) __builtin_amdgcn_s_barrier();
  };

  // ---- wave-local epilogue ---------------------------------------------------------------------------------------------
  auto epilogue = [&](int im, int hh0, int ww0) {
    if (UZ_PP_SKEL & 8) {
      if constexpr (S16) {
#pragma unroll
        for (int pt = 0; pt < 8; ++pt) asm volatile("" ::"v"(accs[pt][0]), "v"(accs[pt][1]), "v"(accs[pt][2]), "v"(accs[pt][3]));
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(acc[i][0]), "v"(acc[i][1]));
      }
      return;
    }
    char* const stg = smem + OFF_STG + wave * STG_W;
    int ln = lane;   // opaque: the address arithmetic of the epilogue must not live in registers through the main loop
    asm volatile("" : "+v"(ln));
    const int l31 = ln & 31, lh = ln >> 5, l15 = ln & 15, lq = ln >> 4;
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
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int hh = hh0 + 4 * wm + i;
      Vec16<T> yb[4];
      if constexpr (BNRED) {
        const T* __restrict__ by = static_cast<const T*>(a.bn_y);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int hc = min(hh, a.H - 1), wc = min(ww0 + (ln >> 3) + 8 * k, a.W - 1);   // clamped: unused outside the image
          yb[k] = ld16(by + ((size_t)(im * a.H + hc) * a.W + wc) * a.ld_bny + (nch < a.Nout ? nch : 0));
        }
      }
      // stage: a lane holds 4 consecutive channels of one pixel per register quad -> one 8-byte LDS write
      if constexpr (S16) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) {
            bf16x4 pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = (bf16_t)accs[2 * i + h][ct][e];
            *reinterpret_cast<bf16x4*>(stg + (16 * h + l15) * STG_ROW + (16 * ct + 4 * lq) * ES) = pk;
          }
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            bf16x4 pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = (bf16_t)acc[i][j][4 * q + e];
            *reinterpret_cast<bf16x4*>(stg + l31 * STG_ROW + (32 * j + 8 * q + 4 * lh) * ES) = pk;
          }
      }
      // read back: lane = (pixel (lane >> 3) + 8 k, channel chunk lane & 7); LDS operations of one wave execute in order
      Vec16<T> vb[4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        vb[k] = *reinterpret_cast<const Vec16<T>*>(stg + ((ln >> 3) + 8 * k) * STG_ROW + cc * 16);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int ww = ww0 + (ln >> 3) + 8 * k;
        const bool inside = hh < a.H && ww < a.W && nch < a.Nout;
        // buffer stores executed by every lane (outside the image: out of range): exactly NSTORE vector-memory
        // operations per wave and tile, which the counted waits of the next tile's first units allow for
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
    for (int c = 0; c < ncb; ++c) {
      const bool first = c == 0, last = c == ncb - 1;
      // from the last slab on, the halo requests are those of the next tile's first patch
      if (last && has_next) compute_avoff(nim, nh0, nw0);
      if (first) unit(IntC<0>(), IntC<1>(), c, first, last, has_next);
      else unit(IntC<0>(), IntC<0>(), c, first, last, has_next);
      tail_barrier();
      unit(IntC<1>(), IntC<0>(), c, first, last, has_next); tail_barrier();
      unit(IntC<2>(), IntC<0>(), c, first, last, has_next); tail_barrier();
      unit(IntC<3>(), IntC<0>(), c, first, last, has_next); tail_barrier();
      unit(IntC<4>(), IntC<0>(), c, first, last, has_next); tail_barrier();
      unit(IntC<5>(), IntC<0>(), c, first, last, has_next); tail_barrier();
      unit(IntC<6>(), IntC<0>(), c, first, last, has_next); tail_barrier();
      unit(IntC<7>(), IntC<0>(), c, first, last, has_next); tail_barrier();
      unit(IntC<8>(), IntC<0>(), c, first, last, has_next); tail_barrier();
      apar ^= 1;
    }
    if constexpr (BNRED) wait_vmcnt<0>();   // (the epilogue's own loads would make the compiler wait for everything anyway)
    epilogue(img, h0, w0);
    img = nim;
    h0 = nh0;
    w0 = nw0;
  }

  // ---- statistics: fixed-order sum over the 32 lanes (4 waves x 8 pixel groups) that own a channel chunk ---------------------
  if (a.stats != nullptr) {
    wait_vmcnt<0>();
    __syncthreads();
    // thread `th` left its sums [2][VEC] at strip(th >> 6) + (th & 63) * 64
    auto red = [&](int th, int idx) { return reinterpret_cast<const float*>(smem + OFF_STG + (th >> 6) * STG_W + (th & 63) * 64)[idx]; };
    if (tid < 2 * BN) {
      const int which = tid / BN, ch = tid - which * BN;
      const int cwn = ch >> 6, cc = (ch & 63) >> 3, e = ch & 7;
      float t = 0.f;
      for (int m = 0; m < 4; ++m)
        for (int l = 0; l < 8; ++l) {
          const int th = ((2 * m + cwn) << 6) + l * 8 + cc;
          t += red(th, which * VEC + e);
        }
      if (n0 + ch < a.Nout) a.stats[((size_t)blockIdx.x * 2 + which) * a.Nout + n0 + ch] = t;
    }
  }
}

}  // namespace

// ---- host side --------------------------------------------------------------------------------------------------------
// returns 1 and fills the plan when the ping-pong kernel takes this descriptor
int uz_pp_plan(const uz_conv_desc* d, UzPpPlan* p) {
  const bool up = d->taps_mode == UZ_TAPS_CONV_UP2;
  if (d->dtype != UZ_BF16) return 0;
  if (!(d->taps_mode == UZ_TAPS_CONV || up) || d->ntaps != 9 || d->dil != 1 || d->store_mode != UZ_STORE_PLAIN) return 0;
  if (up && ((d->H & 1) || (d->W & 1) || d->Hin * 2 != d->H || d->Win * 2 != d->W)) return 0;
  if (d->Cin % KU != 0 || d->Nout % 8 != 0 || d->ldy % 8 != 0 || d->ldx % 8 != 0) return 0;
  if (d->Nout < 128 || d->W < 32 || d->H < 16) return 0;
  const long long xbytes = ((long long)d->N * d->Hin * d->Win - 1) * d->ldx * 2 + (long long)d->Cin * 2;
  const long long wbytes = (long long)d->Nout * 9 * d->Cin * 2;
  const long long ybytes = ((long long)d->N * d->H * d->W - 1) * d->ldy * 2 + (long long)d->Nout * 2;
  if (xbytes >= (1LL << 31) || wbytes >= (1LL << 31) || ybytes >= (1LL << 31)) return 0;
  p->th_n = (d->H + TH - 1) / TH;
  p->tw_n = (d->W + TW - 1) / TW;
  p->ntiles = d->N * p->th_n * p->tw_n;
  p->tiles_n = (d->Nout + BN - 1) / BN;
  // worth it when the 512-pixel tiles still fill the chip: compare the rounds of the two tilings, the ping-pong
  // round counted at 2 x 0.8 of a second-generation round (twice the pixels, measured ~20 % more MFMA throughput)
  const long long t512 = (long long)p->ntiles * p->tiles_n;
  const int tw2 = 32, th2 = 8;
  const long long t256 = (long long)d->N * ((d->H + th2 - 1) / th2) * ((d->W + tw2 - 1) / tw2) * p->tiles_n;
  const long long r512 = (t512 + UZ_NUM_CU - 1) / UZ_NUM_CU, r256 = (t256 + UZ_NUM_CU - 1) / UZ_NUM_CU;
  if (!(uz_tune_flags() & 0x1000000) && r512 * 16 >= r256 * 10) return 0;
  if (uz_tune_flags() & 0x2000000) return 0;   // ablation build: keep the second-generation kernels
  int cap = UZ_NUM_CU / p->tiles_n;
  if (cap < 1) cap = 1;
  p->grid_m = p->ntiles < cap ? p->ntiles : cap;
  return 1;
}

int uz_pp_launch(const uz_conv_desc* d, const UzPpPlan& p, const void* x, const void* w, const float* bias, void* y,
                 float* stats, hipStream_t s, const UzBnRed* br) {
  PpArgs a;
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
  dim3 grid(p.grid_m, p.tiles_n), block(512);
#ifdef UZ_ABLATE
  if (uz_tune_flags() & 0x4000000) {   // the 32x32x16 stream
    if (br) hipLaunchKernelGGL((conv3x3_pp_kernel<true, false>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((conv3x3_pp_kernel<false, false>), grid, block, 0, s, a);
    UZ_LAUNCH_CHECK("uz_conv_igemm(direct3x3 ping-pong 32x32x16)");
    return UZ_OK;
  }
#endif
  if (br) hipLaunchKernelGGL((conv3x3_pp_kernel<true, true>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((conv3x3_pp_kernel<false, true>), grid, block, 0, s, a);
  UZ_LAUNCH_CHECK("uz_conv_igemm(direct3x3 ping-pong)");
  return UZ_OK;
}
