// Direct 3x3 convolution (stride 1, dilation 1, zero padding 1) for gfx950 (MI355X), NHWC.
//
// Stands in for nn.Conv2d(k=3, padding=1) forward and, with flipped/transposed packed weights,
// its input-gradient (reference: unet_zoo/models/common_layers.py:28,31,47,52,71; autograd a19).
//
// One 512-thread workgroup (8 waves, one per CU) owns a 2-D patch of 256 output pixels
// (8x32 or 16x16) and BN output channels.  For each 128-byte slab of input channels the
// (TH+2)x(TW+2) halo patch is brought into LDS ONCE by LDS-DMA (buffer_load ... lds: rows that
// fall outside the image are made out-of-range in the buffer descriptor, so the hardware writes the
// zero padding) and then serves all nine taps: a tap is only a different row offset in LDS.
// Weights stream through LDS in [BN][128 B] tiles, one per (slab, tap) unit: two units per raw s_barrier
// (two slots of two tiles; the pair for the next double-step is issued during the current one, behind a
// counted `s_waitcnt vmcnt(N)`) -- 8-13 % faster than one unit per barrier with a 3-slot ring, because the
// barrier + wait costs a quarter of the MFMA time of a 64-deep K step.
// When the whole [BN][9*Cin] weight matrix fits (Cin = one slab, BN = 64: the memory-bound
// 64->64 layers) it is loaded once per workgroup and stays resident while the workgroup walks its
// tiles, and the next tile's halo patch is prefetched during the current tile's 144 MFMAs per wave.
// LDS rows are 128 bytes; 16-byte chunks are XOR-swizzled with ((row>>1)&7) on the DMA SOURCE
// address and on the ds_read_b128 address (the LDS image itself stays lane-linear).
// Epilogue: bias, BatchNorm partial sums of the stored value, then the bf16 tile is transposed
// through LDS so that every lane stores 16 contiguous bytes of one pixel (full-line writes).
#include "uz_common.h"

namespace {

struct DirectArgs {
  const void* x;
  const void* w;
  void* y;
  const float* bias;
  float* stats;
  unsigned xbytes, wbytes;
  unsigned ybytes;   // size of y when < 2 GiB (buffer stores with a counted wait, see the streaming epilogue), else 0
  int N, H, W, Cin, ldx, Nout, ldy, K;
  int th_n, tw_n, ntiles;
  int flags;  // tuning/ablation switches (env UZ_TUNE, tools/kbench.py); 0 in production
  int ups;    // 1: the input lives at (H/2, W/2) and is read through nearest x2 upsampling
  // BatchNorm-backward reduction fused into the epilogue (uz_conv_igemm_bnred): this launch computes the gradient g
  // of the activation relu(bn(bn_y)); instead of the statistics of its own output the workgroup rows of `stats`
  // receive sum(dz) and sum(dz * xhat) with dz = g * [bn_scale * bn_y + bn_shift > 0], xhat = (bn_y - mean) * invstd
  const void* bn_y;
  const float* bn_scale;
  const float* bn_shift;
  const float* bn_mean;
  const float* bn_invstd;
  int ld_bny;
};

template <typename T> struct Mma2;
template <> struct Mma2<bf16_t> {
  // bf16 tiles are accumulated TRANSPOSED (weights as the row operand): accumulator column = pixel
  // (lane & 31), rows = output channels, so that a lane owns 4 CONSECUTIVE channels of one pixel per
  // register quad and the epilogue stages them with 8-byte LDS writes (see the epilogue)
  static __device__ __forceinline__ void run(const Vec16<bf16_t>& a, const Vec16<bf16_t>& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&b),
                                                *reinterpret_cast<const bf16x8*>(&a), c, 0, 0, 0);
  }
};
template <> struct Mma2<float> {
  static __device__ __forceinline__ void run(const Vec16<float>& a, const Vec16<float>& b, f32x16& c) {
#pragma unroll
    for (int t = 0; t < 4; ++t) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[t], b.v[t], c, 0, 0, 0);
  }
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, unsigned voff) {
  // one wave-instruction: lane i writes LDS bytes [base + 16 i, +16) with the 16 bytes at voff
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)lds_wave_base, 16, voff, 0, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt range");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr unsigned OOB = 0x80000000u;  // beyond any descriptor's num_records (tensors < 2 GiB; it was 2^28 until round 2: inside every tensor of 256 MiB or more, whose zero padding then read real pixels -- tests/test_ops_gpu.py::test_conv3x3_padding_of_a_tensor_beyond_256_mib)
#ifndef UZ_CONV_SKEL
#define UZ_CONV_SKEL 0   // measurement builds (-DUZ_CONV_SKEL=bits): 1 no fragment reads, 2 no MFMAs, 4 no DMA past the first double-step, 8 no epilogue
#endif

// Fragment reads of the streaming kernel's double-step are issued from inline asm with hand-counted lgkmcnt waits:
// left to the compiler every ds_read_b128 group is followed by `s_waitcnt lgkmcnt(0)` and two MFMAs, i.e. a bare LDS
// round trip per K-chunk; as asm the reads of chunk g + 1 leave before the MFMAs of chunk g.
__device__ __forceinline__ void lds_read16(f32x4& dst, unsigned lds_addr) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(lds_addr));
}
template <int N> __device__ __forceinline__ void wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
template <int V> struct IntC { static constexpr int value = V; };
__device__ __forceinline__ void pin16(f32x4& v) { asm volatile("" : "+v"(v)); }

// BNRED: the epilogue accumulates the BatchNorm-backward sums of DirectArgs::bn_* instead of output statistics (a
// separate instantiation: the extra live registers of that epilogue must not weigh on the plain kernels; no bias)
template <typename T, int TW, int BN, bool BRES, bool BNRED = false>
__global__ __launch_bounds__(512, 1) void conv3x3_direct_kernel(const DirectArgs a) {
  constexpr int VEC = ElemTraits<T>::VEC;
  constexpr int ES = (int)sizeof(T);
  constexpr int BK = 8 * VEC;
  constexpr int TH = 256 / TW, PH = TH + 2, PW = TW + 2, PROWS = PH * PW;
  constexpr int APIECES = (PROWS + 7) / 8;
  constexpr int A_BYTES = APIECES * 1024;
  constexpr int APW = (APIECES + 7) / 8;  // A pieces per wave
  constexpr int NBP = BN / 64;            // B pieces per wave per step
  constexpr int B_STAGE = BN * 128;
  constexpr int B_SLOTS = BRES ? 9 : 4;   // 3-slot ring (one tap per barrier) or 2 x 2 (two taps per barrier)
  constexpr int TN = BN / 64;             // 32-wide N tiles per wave (waves: 4 (M) x 2 (N))
  constexpr int WTN = BN / 2;
  static_assert(APW <= 9, "A patch pieces must fit the nine tap steps");
  __shared__ __attribute__((aligned(16))) char smem[2 * A_BYTES + B_SLOTS * B_STAGE];
  char* const sB = smem + 2 * A_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.y * BN;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.wbytes, 0x00020000);
  T* __restrict__ yg = static_cast<T*>(a.y);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.ybytes, 0x00020000);

  // ---- DMA pieces --------------------------------------------------------------------------
  // A piece p = wave + 8 i covers patch rows 8p .. 8p+7; this lane: row 8p + (lane>>3), physical
  // chunk (lane&7), which must hold logical chunk (lane&7) ^ swz(row).  Everything is recomputed
  // from i when a piece is issued (a handful of VALU ops) instead of living in registers.
  unsigned b_row_off[NBP];
  int b_ch[NBP];  // first channel (inside a slab) of the logical chunk this lane fetches
#pragma unroll
  for (int i = 0; i < NBP; ++i) {
    const int n = (wave + 8 * i) * 8 + (lane >> 3);
    b_row_off[i] = (n0 + n < a.Nout) ? (unsigned)(n0 + n) * (unsigned)a.K * ES : OOB;
    b_ch[i] = ((lane & 7) ^ ((n >> 1) & 7)) * VEC;
  }
  // ---- per-lane constants of the MFMA fragment reads ----------------------------------------
  // M tile mt = 2 wm + i: 32 pixels = one patch row (TW 32) or two half rows (TW 16)
  int f_pi[2], f_pj;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int mt = 2 * wm + i;
    f_pi[i] = (TW == 32) ? mt : 2 * mt + (l31 >> 4);
  }
  f_pj = (TW == 32) ? l31 : (l31 & 15);
  int b_frag_off[TN];
  int b_sw[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int brow = wn * WTN + j * 32 + l31;
    b_frag_off[j] = brow * 128;
    b_sw[j] = (brow >> 1) & 7;
  }

  if ((UZ_KFLAGS(a) & 0x20) && wave >= 4) __builtin_amdgcn_s_setprio(1);
  if ((UZ_KFLAGS(a) & 0x2000) && wave < 4) __builtin_amdgcn_s_setprio(1);
  const int ncb = (a.Cin + BK - 1) / BK;  // the last slab may be partial: channels >= Cin read as zero
  float s1[TN], s2[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) s1[j] = s2[j] = 0.f;

  int img = 0, h0 = 0, w0 = 0;
  auto decode = [&](int tile, int& im, int& hh0, int& ww0) {
    const int per = a.th_n * a.tw_n;
    im = tile / per;
    const int rem = tile - im * per;
    const int ti = rem / a.tw_n;
    hh0 = ti * TH;
    ww0 = (rem - ti * a.tw_n) * TW;
  };
  auto issue_a_piece = [&](int i, int buf, int cb, int im, int hh0, int ww0) -> bool {
    const int piece = wave + 8 * i;
    if (piece >= APIECES) return false;  // wave-uniform
    const int r = piece * 8 + (lane >> 3);
    const int pi = r / PW, pj = r - pi * PW;
    const int hh = hh0 - 1 + pi, ww = ww0 - 1 + pj;
    const int ch = cb * BK + ((lane & 7) ^ ((r >> 1) & 7)) * VEC;
    const bool ok = r < PROWS && (unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)a.W && ch < a.Cin;
    const unsigned pix = a.ups ? (unsigned)((im * (a.H >> 1) + (hh >> 1)) * (a.W >> 1) + (ww >> 1))
                               : (unsigned)((im * a.H + hh) * a.W + ww);
    const unsigned off = ok ? (pix * (unsigned)a.ldx + ch) * ES : OOB;
    dma16(xr, smem + buf * A_BYTES + piece * 1024, off);
    return true;
  };
  auto issue_b_piece = [&](int i, int slot, int cb, int tap) {
    const int ch = cb * BK + b_ch[i];
    const unsigned off = (b_row_off[i] == OOB || ch >= a.Cin) ? OOB : b_row_off[i] + (tap * a.Cin + ch) * ES;
    dma16(wr, sB + slot * B_STAGE + (wave + 8 * i) * 1024, off);
  };
  auto issue_b = [&](int slot, int cb, int tap) {
#pragma unroll
    for (int i = 0; i < NBP; ++i) issue_b_piece(i, slot, cb, tap);
  };

  f32x16 acc[2][TN];
  // ---- the double-step as one read pipeline (streaming path) ----------------------------------------------
  // chunk g = 0 .. 7 of the double-step = K-chunk g & 3 of unit g >> 2; set g & 1 of the fragment registers.
  typedef __attribute__((address_space(3))) char* lds_char_ptr;
  const unsigned smem_u = (unsigned)(size_t)(lds_char_ptr)smem;
  unsigned boffq[4][TN];   // weight fragment of K-chunk q, N tile j, relative to the slot (tap-invariant)
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int j = 0; j < TN; ++j)
      boffq[q][j] = smem_u + 2 * A_BYTES + b_frag_off[j] + (((2 * q + lh) ^ b_sw[j]) << 4);
  struct UnitAddr { unsigned arow[2]; int asw[2]; unsigned bbase; };
  auto unit_addr = [&](int tap, int abuf, int bslot) __attribute__((always_inline)) {
    UnitAddr u;
    const int ty = (tap * 11) >> 5, tx = tap - 3 * ty;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int prow = (f_pi[i] + ty) * PW + f_pj + tx;
      u.arow[i] = smem_u + abuf * A_BYTES + prow * 128;
      u.asw[i] = (prow >> 1) & 7;
    }
    u.bbase = bslot * B_STAGE;
    return u;
  };
  constexpr int RPC = 2 + TN;   // LDS reads per K-chunk
  // ---- unit-sized segments ---------------------------------------------------------------------------------------
  // A wave requests the sixteen fragments of a whole unit (64 VGPRs, inline-asm ds_read_b128 with hand-counted
  // lgkmcnt waits), then issues its sixteen MFMAs, K-chunk q as soon as its four fragments are in.
  // Measured with compile-time variants of this kernel on 256 -> 256 @ 64 x 64 (B = 16; -DUZ_CONV_SKEL): skeleton
  // (no reads, MFMAs, DMA) 15 us, + fragment reads 38 us, + MFMAs instead 53 us, reads AND MFMAs 68-73 us, + the
  // LDS-DMA stream 82-84 us: read time adds to MFMA time almost in full.  Tried against that, all within +-5 % of each
  // other: the compiler's own read -> wait -> 2 MFMA groups (round 1), reads of chunk g + 1 ahead of the MFMAs of chunk g,
  // reads of chunk g + 2 between the MFMAs of chunk g, these unit-long segments (best by 2-4 % on the 128- and
  // 256-channel layers), s_setprio for either half of the waves, accumulators in AGPRs (30-60 % slower: copies in the
  // epilogue and a 128-VGPR budget).
  f32x4 ua[4][2], ub[4][TN];
  auto read_unit = [&](const UnitAddr& u) __attribute__((always_inline)) {
    if (UZ_CONV_SKEL & 1) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int lc = 2 * q + lh;
#pragma unroll
      for (int i = 0; i < 2; ++i) lds_read16(ua[q][i], u.arow[i] + ((lc ^ u.asw[i]) << 4));
#pragma unroll
      for (int j = 0; j < TN; ++j) lds_read16(ub[q][j], boffq[q][j] + u.bbase);
    }
  };
  auto mma_unit = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q == 0) wait_lgkm<3 * RPC>();
      else if (q == 1) wait_lgkm<2 * RPC>();
      else if (q == 2) wait_lgkm<RPC>();
      else wait_lgkm<0>();
#pragma unroll
      for (int i = 0; i < 2; ++i) pin16(ua[q][i]);
#pragma unroll
      for (int j = 0; j < TN; ++j) pin16(ub[q][j]);
      if (UZ_CONV_SKEL & 2) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          Mma2<T>::run(*reinterpret_cast<const Vec16<T>*>(&ua[q][i]), *reinterpret_cast<const Vec16<T>*>(&ub[q][j]), acc[i][j]);
    }
  };

  // Resident-weight tiles have only two MFMAs per fragment set (wave tile 64x32), too few to
  // cover an LDS round trip: walk the 36 (tap, K-chunk) groups of a tile as ONE software pipeline,
  // group g+1's three ds_read_b128 issued ahead of group g's MFMAs (two register sets; the order is
  // pinned with sched_group_barrier because the scheduler otherwise folds the sets back into one).
  auto compute_tile_pipelined = [&](int abuf) {
    const char* sA = smem + abuf * A_BYTES;
    Vec16<T> af[2][2], bf[2][TN];
    auto load_group = [&](int gidx, int set) {
      const int tap = gidx >> 2, q = gidx & 3;
      const int ty = tap / 3, tx = tap - 3 * ty;
      const int lc = 2 * q + lh;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int prow = (f_pi[i] + ty) * PW + f_pj + tx;
        af[set][i] = *reinterpret_cast<const Vec16<T>*>(sA + prow * 128 + ((lc ^ ((prow >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bf[set][j] = *reinterpret_cast<const Vec16<T>*>(sB + tap * B_STAGE + b_frag_off[j] + ((lc ^ b_sw[j]) << 4));
    };
    load_group(0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 2 + TN, 0);
#pragma unroll
    for (int gidx = 0; gidx < 36; ++gidx) {
      if (gidx + 1 < 36) {
        load_group(gidx + 1, (gidx + 1) & 1);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 + TN, 0);   // DS reads of the next group
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) Mma2<T>::run(af[gidx & 1][i], bf[gidx & 1][j], acc[i][j]);
      __builtin_amdgcn_sched_group_barrier(0x008, (sizeof(T) == 2 ? 2 : 8) * TN, 0);  // this group's MFMAs
    }
  };

  if constexpr (BRES) {
    // resident weights: 9 tap tiles, loaded once
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) issue_b(tap, 0, tap);
  }
  int first = blockIdx.x;
  if (BRES && first < a.ntiles) {
    decode(first, img, h0, w0);
#pragma unroll
    for (int i = 0; i < APW; ++i) issue_a_piece(i, 0, 0, img, h0, w0);
    wait_vmcnt<0>();  // resident weights + first patch
  }

  float bv[TN];  // fp32 path: bias of this lane's output channel (loaded once, not per tile)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * WTN + j * 32 + l31;
    bv[j] = (a.bias != nullptr && n < a.Nout) ? a.bias[n] : 0.f;
  }
  // bf16 path (transposed accumulators): accumulator register r of N tile j is channel
  // wn*WTN + 32 j + (r & 3) + 8 (r >> 2) + 4 lh
  float bq[TN][16];
  if constexpr (sizeof(T) == 2 && !BNRED) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * WTN + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        bq[j][r] = (a.bias != nullptr && n < a.Nout) ? a.bias[n] : 0.f;
      }
  }
  // read-back phase of the bf16 epilogue: thread `tid` always handles the 16-byte channel chunk
  // tid % CPR of 256*CPR/512 pixels, so its statistics accumulate in registers across tiles
  float sq1[VEC], sq2[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) sq1[e] = sq2[e] = 0.f;

  int it = 0;
  int par = 0, dpar = 0;   // streaming path: halo buffer of slab 0 / slot pair of double-step 0 of the current tile
  bool pre = false;        // ... whose first patch and weight pair the previous tile has already requested
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x, ++it) {
    decode(tile, img, h0, w0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int cbuf;       // A buffer that holds the C staging area afterwards
    int cpair = 0;  // streaming path: slot pair that holds the staging rows the A buffer has no room for
    if constexpr (BRES) {
      // this tile's patch (and the weights) were waited for before the previous epilogue's stores
      // were issued, so those stores may still be in flight here: only the barrier is needed
      __builtin_amdgcn_s_barrier();
      const int next = tile + gridDim.x;
      if (next < a.ntiles) {
        int im2, hh2, ww2;
        decode(next, im2, hh2, ww2);
#pragma unroll
        for (int i = 0; i < APW; ++i) issue_a_piece(i, (it + 1) & 1, 0, im2, hh2, ww2);
      }
      compute_tile_pipelined(it & 1);
      wait_vmcnt<0>();  // next tile's patch has landed (it had the whole tap loop to arrive)
      cbuf = it & 1;
    } else {
      // ---- streaming path: the (slab, tap) units of ALL tiles of this workgroup form one stream ----------------------
      // Two units per barrier: weight tiles in two slot pairs, the pair for double-step d + 1 issued during d; halo
      // pieces of the next slab (<= 2 per double-step) issued AFTER the weight tiles so that the counted wait at the next
      // barrier may leave them in flight.  During a tile's LAST slab the pieces are those of the NEXT tile's first slab,
      // and its last double-step issues the next tile's first weight pair: the next tile starts without a DMA round
      // trip (2.5-3.4 us per tile).  `par` / `dpar`: which halo buffer holds slab 0 and which slot pair holds
      // double-step 0 of the current tile; the epilogue stages C in the halo buffer and the slot pair that the last slab
      // / double-step have just released (the other ones hold the prefetch).
      const int nsteps = ncb * 9;
      const int ndbl = (nsteps + 1) >> 1;
      constexpr int NSTORE = 256 * (BN * ES / 16) / 512;   // epilogue stores per wave (= NPASS there)
      auto unit = [&](int L, int& c, int& t) {
        c = L / 9;
        t = L - 9 * c;
      };
      const int next = tile + gridDim.x;
      const bool has_next = next < a.ntiles && !(UZ_CONV_SKEL & 32) && !(UZ_KFLAGS(a) & 0x4000);   // (0x4000: no cross-tile prefetch)
      int nim = 0, nh0 = 0, nw0 = 0;
      if (has_next) decode(next, nim, nh0, nw0);
      int kprev = 0;       // halo pieces this wave issued after the last weight tile
      if (!pre) {
        __builtin_amdgcn_s_barrier();  // previous tile (its C staging reads) is finished everywhere
#pragma unroll
        for (int i = 0; i < APW; ++i) issue_a_piece(i, par, 0, img, h0, w0);
        issue_b(dpar * 2, 0, 0);
        if (nsteps > 1) issue_b(dpar * 2 + 1, 0, 1);
      }
      int np = 0, npslab = 0;  // next halo piece of slab npslab + 1
#pragma unroll 1
      for (int d = 0; d < ndbl; ++d) {
        // (d == 0 of a prefetched tile: the previous epilogue's stores are younger than the prefetch -- wait for all)
        if (d == 0 && pre && a.ybytes != 0 && sizeof(T) == 2 && !(UZ_KFLAGS(a) & 1)) wait_vmcnt<NSTORE>();
        else if (kprev == 0 || d == 0) wait_vmcnt<0>();
        else if (kprev == 1) wait_vmcnt<1>();
        else wait_vmcnt<2>();
        __builtin_amdgcn_s_barrier();
        const int L0 = 2 * d, L1 = L0 + 1;
        int c0, t0, c1 = 0, t1 = 0;
        unit(L0, c0, t0);
        if (L1 < nsteps) unit(L1, c1, t1);
        auto issue_next = [&]() {
          if (UZ_CONV_SKEL & 4) return;   // measurement build: no DMA after the first double-step
          if (d + 1 < ndbl) {  // weight tiles of the next double-step into the other slot pair
            const int s2 = ((d + 1 + dpar) & 1) * 2;
            int c, t;
            unit(L0 + 2, c, t);
            issue_b(s2, c, t);
            if (L0 + 3 < nsteps) {
              unit(L0 + 3, c, t);
              issue_b(s2 + 1, c, t);
            }
          } else if (has_next) {   // the next tile's first double-step
            const int s2 = ((ndbl + dpar) & 1) * 2;
            issue_b(s2, 0, 0);
            if (nsteps > 1) issue_b(s2 + 1, 0, 1);
          }
          kprev = 0;
          if (c0 != npslab) {
            npslab = c0;
            np = 0;
          }
          // both units of this double-step lie in slabs >= c0, so the halo buffer of slab c0 + 1 (last read by slab
          // c0 - 1, or by the previous tile's epilogue) is free unless the second unit already belongs to slab c0 + 1
          // (t0 == 8: nothing left to issue)
          if (t0 < 8 && (c0 + 1 < ncb || has_next)) {
            const bool nxt = c0 + 1 == ncb;
#pragma unroll
            for (int k = 0; k < 2; ++k)
              if (np < APW) {
                const bool did = nxt ? issue_a_piece(np, (ncb + par) & 1, 0, nim, nh0, nw0)
                                     : issue_a_piece(np, (c0 + 1 + par) & 1, c0 + 1, img, h0, w0);
                if (did) ++kprev;
                ++np;
              }
          }
        };
        const int s0 = ((d + dpar) & 1) * 2;
        // the two waves of a SIMD (w, w + 4) run the same program in lockstep: with `late` the second one issues
        // its LDS-DMA pieces AFTER its first unit, so one wave's issue runs beside the other's MFMAs
        // (+2 ... 4.5 % on every non-resident layer; flag 0x10 of the ablation build switches it off)
        const bool late = !(UZ_KFLAGS(a) & 0x10) && wave >= 4;
        const bool two = L1 < nsteps;
        const UnitAddr u0 = unit_addr(t0, (c0 + par) & 1, s0), u1 = unit_addr(t1, (c1 + par) & 1, s0 + 1);
        if (!(UZ_CONV_SKEL & 16) || d == 0) read_unit(u0);   // (16: real fragments once per tile, then MFMAs only)
        if (!late) issue_next();
        mma_unit();
        __builtin_amdgcn_sched_barrier(0);
        if (two && !(UZ_CONV_SKEL & 16)) read_unit(u1);
        if (late) issue_next();
        if (two) mma_unit();
      }
      cbuf = (ncb - 1 + par) & 1;            // halo buffer of the last slab: free for the C staging
      cpair = (ndbl - 1 + dpar) & 1;         // slot pair of the last double-step: the staging's second segment
      par = (ncb + par) & 1;
      dpar = (ndbl + dpar) & 1;
      pre = has_next;
    }

    // ---- epilogue: bias + statistics from registers ------------------------------------------
    if (((UZ_KFLAGS(a) & 1) | (UZ_CONV_SKEL & 8)) != 0) {
      asm volatile("" ::"v"(acc[0][0][0]), "v"(acc[1][0][3]));
    } else if constexpr (sizeof(T) == 2) {
      constexpr int RSC = BN * ES + 16;  // C staging row stride (bytes)
      constexpr int CPR = BN * ES / 16;  // 16-byte chunks per pixel
      static_assert(512 % CPR == 0, "a thread keeps one channel chunk");
      constexpr int NPASS = 256 * CPR / 512, RPP = 512 / CPR;
      const int cc = tid % CPR;
      const int n = n0 + cc * VEC;
      // fused BatchNorm-backward reduction: the pre-activation values of this thread's read-back pixels are requested
      // now, a staging round trip ahead of their use
      Vec16<T> yb[NPASS];
      if constexpr (BNRED) {
        const T* __restrict__ by = static_cast<const T*>(a.bn_y);
#pragma unroll
        for (int k = 0; k < NPASS; ++k) {
          const int m = tid / CPR + k * RPP;
          const int mt = m >> 5, ml = m & 31;
          const int pi = (TW == 32) ? mt : 2 * mt + (ml >> 4);
          const int pj = (TW == 32) ? ml : (ml & 15);
          const int hh = min(h0 + pi, a.H - 1), ww = min(w0 + pj, a.W - 1);   // clamped: unused outside the image
          yb[k] = ld16(by + ((size_t)(img * a.H + hh) * a.W + ww) * a.ld_bny + (n < a.Nout ? n : 0));
        }
      }
      __builtin_amdgcn_s_barrier();  // every wave has finished reading A/B of this tile
      // staging rows 0 .. R0 - 1 in halo buffer `cbuf`, the rest (BN = 128: 95 rows) in slot pair `cpair`
      constexpr int R0 = (A_BYTES / RSC < 256) ? A_BYTES / RSC : 256;
      static_assert(BRES ? R0 == 256 : (256 - R0) * RSC <= 2 * B_STAGE, "C staging must fit a halo buffer + a slot pair");
      char* const sC0 = smem + cbuf * A_BYTES;
      char* const sC1 = sB + cpair * 2 * B_STAGE - R0 * RSC;   // row r >= R0 at sC1 + r * RSC
      auto stage_row = [&](int r) -> char* { return (R0 == 256 || r < R0 ? sC0 : sC1) + r * RSC; };
      // stage: lane = pixel (l31) of M tile mt, register quad q = 4 consecutive channels -> one
      // 8-byte LDS write (was: 64 two-byte writes per lane with sub-dword bank conflicts)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int mt = 2 * wm + i;
        char* rowp = stage_row(mt * 32 + l31);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            bf16x4 pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = (bf16_t)(BNRED ? acc[i][j][4 * q + e] : acc[i][j][4 * q + e] + bq[j][4 * q + e]);
            *reinterpret_cast<bf16x4*>(rowp + (wn * WTN + j * 32 + 8 * q + 4 * lh) * ES) = pk;
          }
        }
      }
      // the staging writes must have EXECUTED, not just issued, before another wave reads them: s_barrier alone
      // does not wait for the LDS queue.  (Found when a cross-tile prefetch of the next activation patch -- LDS-DMA
      // traffic on the LDS write path during the epilogue -- made 8 pixels x 16 channels of one tile in a million read
      // stale weight-slot bytes; the prefetch itself measured no gain and was dropped, the wait stays.)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const bool dostats = !(UZ_KFLAGS(a) & 2);
      float bsc[VEC], bsh[VEC], bmu[VEC], bis[VEC];   // (loaded here: the accumulators are dead by now)
      if constexpr (BNRED) {
        const int ch0 = (n < a.Nout) ? n : 0;
#pragma unroll
        for (int e = 0; e < VEC; e += 4) {   // channel chunks start at multiples of 8: 16-byte aligned
          *reinterpret_cast<f32x4*>(&bsc[e]) = *reinterpret_cast<const f32x4*>(a.bn_scale + ch0 + e);
          *reinterpret_cast<f32x4*>(&bsh[e]) = *reinterpret_cast<const f32x4*>(a.bn_shift + ch0 + e);
          *reinterpret_cast<f32x4*>(&bmu[e]) = *reinterpret_cast<const f32x4*>(a.bn_mean + ch0 + e);
          *reinterpret_cast<f32x4*>(&bis[e]) = *reinterpret_cast<const f32x4*>(a.bn_invstd + ch0 + e);
        }
      }
      // every thread reads back 256 * CPR / 512 chunks: all of them are requested before the first store (as a
      // rolled loop each pass waited for its own LDS round trip: ~250 cycles x 8 per tile with every wave idle)
      Vec16<T> vb[NPASS];
#pragma unroll
      for (int k = 0; k < NPASS; ++k)
        vb[k] = *reinterpret_cast<const Vec16<T>*>(stage_row(tid / CPR + k * RPP) + cc * 16);
#pragma unroll
      for (int k = 0; k < NPASS; ++k) {
        const int m = tid / CPR + k * RPP;
        const int mt = m >> 5, ml = m & 31;
        const int pi = (TW == 32) ? mt : 2 * mt + (ml >> 4);
        const int pj = (TW == 32) ? ml : (ml & 15);
        const int hh = h0 + pi, ww = w0 + pj;
        const bool inside = hh < a.H && ww < a.W && n < a.Nout;
        if (!BRES && a.ybytes != 0) {
          // a buffer store, executed by every wave with out-of-image lanes sent out of range: exactly NPASS
          // vector-memory operations per wave follow the prefetched LDS-DMA pieces, so the next tile's first barrier
          // can wait with vmcnt(NPASS) instead of draining these stores
          const unsigned off = inside ? (unsigned)((((img * a.H + hh) * a.W + ww) * a.ldy + n) * ES) : OOB;
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(&vb[k]), yr, off, 0, 0);
        } else if (inside) {
          st16(yg + ((size_t)(img * a.H + hh) * a.W + ww) * a.ldy + n, vb[k]);
        }
        if (inside) {
          if constexpr (BNRED) {   // sums of the BatchNorm backward, from the gradient values as stored
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
              const float yv = (float)yb[k].v[e];
              const float dz = fmaf(yv, bsc[e], bsh[e]) > 0.f ? (float)vb[k].v[e] : 0.f;
              sq1[e] += dz;
              sq2[e] += dz * ((yv - bmu[e]) * bis[e]);
            }
          } else if (dostats) {  // statistics of the stored values, pixels inside the image only
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
        const int mt = 2 * wm + i;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n0 + wn * WTN + j * 32 + l31;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int ml = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int pi = (TW == 32) ? mt : 2 * mt + (ml >> 4);
            const int pj = (TW == 32) ? ml : (ml & 15);
            const int hh = h0 + pi, ww = w0 + pj;
            if (n < a.Nout && hh < a.H && ww < a.W) {
              const T tv = (T)(acc[i][j][r] + bv[j]);
              yg[((size_t)(img * a.H + hh) * a.W + ww) * a.ldy + n] = tv;
              const float fv = (float)tv;
              s1[j] += fv;
              s2[j] += fv * fv;
            }
          }
        }
      }
    }
  }

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
    float* red = reinterpret_cast<float*>(smem);  // [4][BN][2]
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
      for (int k = 0; k < 4; ++k) {
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


// ------------------------------------------------------------------------------------------------
// Resident-weight kernel, second generation (bf16, Cin <= 64 = one K slab, 64 output channels per
// workgroup): the memory-bound DoubleConv layers (64 -> 64 at full resolution, common_layers.py:31)
// and every other layer whose input has at most 64 channels.
//
//  * The workgroup's [64][9 * 64] weight matrix lives in REGISTERS, not LDS: wave (wm, wn) keeps the
//    fragments of its 32 output channels for all nine taps (36 x 16 bytes per lane = 144 VGPRs, loaded
//    once).  LDS then holds only activation patches, and a (tap, K-chunk) group costs the LDS one read
//    per 32 MFMA cycles instead of 1.5.
//  * The epilogue is WAVE-LOCAL: a wave transposes its own 64 pixel x 32 channel tile through a
//    private 4 KB staging area (8-byte writes, 16-byte reads, no workgroup barrier) and stores 64
//    contiguous bytes per pixel.
//  * The two waves that share a SIMD (w and w + 4) run half a tile apart: waves 4-7 ("late") issue ALL
//    LDS-DMA pieces of the next patch and then run the epilogue of the PREVIOUS tile while waves 0-3
//    are in their MFMA stream; when waves 0-3 reach their epilogue, waves 4-7 are in theirs.  One
//    s_barrier per tile, the matrix pipe of every SIMD has work the whole interval.
//  * MFMA shape: S = 32 (v_mfma_f32_32x32x16_bf16) or S = 16 (v_mfma_f32_16x16x32_bf16), same LDS and
//    register traffic; the chip holds a higher clock on the 16x16x32 stream.
template <int TW, int S, int DEPTH>
__global__ __launch_bounds__(512, 2) void conv3x3_res64_kernel(const DirectArgs a) {
  typedef bf16_t T;
  constexpr int VEC = 8, ES = 2, BK = 64;
  // The patch is stored COLUMN-major with an odd column height: LDS row of patch pixel (pi, pj) = pj * PHP + pi,
  // 16-byte chunks XOR-swizzled with key(pj) = (pj >> 1) & 7.  A tap (ty, tx) then moves a lane's read by
  // ty * 128 bytes (an immediate offset of the ds_read) and selects one of three per-lane base registers (tx),
  // the K chunk is one XOR with a constant: ~1 VALU per read instead of 5 -- with the weights in registers
  // nothing else can stay hoisted, and two waves share a SIMD's issue port.  Odd PHP makes the row parity
  // alternate along pj, which together with the key keeps every ds_read_b128 lane group conflict-free.
  constexpr int TH = 256 / TW, PH = TH + 2, PW = TW + 2, PHP = PH | 1, PROWS = PW * PHP;
  constexpr int APIECES = (PROWS + 7) / 8;
  constexpr int A_BYTES = APIECES * 1024;
  constexpr int MT = 64 / S, NT = 32 / S;        // MFMA tiles of a wave: 64 pixels x 32 channels
  constexpr int KC = (S == 32) ? 16 : 32;        // K of one MFMA
  constexpr int NKC = BK / KC;                   // K chunks per tap
  constexpr int CPK = KC / 8;                    // 16-byte chunks per K chunk (lane / S selects one)
  constexpr int NACC = (S == 32) ? 16 : 4;
  typedef float accv_t __attribute__((ext_vector_type(NACC)));
  // LDS: two activation patches | 8 wave-private staging areas | bias table [64] | per-thread statistics [512][16]
  constexpr int OFF_BIAS = 2 * A_BYTES + 8 * 4096, OFF_STAT = OFF_BIAS + 256;
  __shared__ __attribute__((aligned(16))) char smem[OFF_STAT + 512 * 64];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const bool late = wave >= 4;
  const int ls = lane % S, lq = lane / S;
  const int n0 = blockIdx.y * 64;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.wbytes, 0x00020000);
  T* __restrict__ yg = static_cast<T*>(a.y);
  char* const sCw = smem + 2 * A_BYTES + wave * 4096;

  // ---- weights into registers: breg[tap][kc][u] = 8 K-elements of channel n0 + wn*32 + u*S + ls ----
  bf16x8 breg[9][NKC][NT];
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    const int n = n0 + wn * 32 + u * S + ls;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int kc = 0; kc < NKC; ++kc) {
        const int kk = kc * KC + lq * 8;
        const unsigned off = (n < a.Nout && kk < a.Cin) ? ((unsigned)n * (unsigned)a.K + tap * a.Cin + kk) * ES : OOB;
        auto v = __builtin_amdgcn_raw_buffer_load_b128(wr, off, 0, 0);
        breg[tap][kc][u] = *reinterpret_cast<bf16x8*>(&v);
      }
  }

  // ---- per-lane constants ----------------------------------------------------------------------
  int abase[MT][3];   // LDS byte offset of this lane's K-chunk-0 read of M tile t at tap (0, tx)
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int m = 64 * wm + t * S + ls;
    const int pi = (TW == 32) ? (m >> 5) : (m >> 4), pj = (TW == 32) ? (m & 31) : (m & 15);
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) abase[t][tx] = ((pj + tx) * PHP + pi) * 128 + ((lq ^ (((pj + tx) >> 1) & 7)) << 4);
  }
  // registers are the scarce resource (144 hold weights): the bias sits in an LDS table and is the accumulators'
  // initial value, the running BatchNorm sums of a thread live in LDS between epilogues
  float* const sBias = reinterpret_cast<float*>(smem + OFF_BIAS);
  f32x4* const sStat = reinterpret_cast<f32x4*>(smem + OFF_STAT + tid * 64);
  if (tid < 64) sBias[tid] = (a.bias != nullptr && n0 + tid < a.Nout) ? a.bias[n0 + tid] : 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) sStat[e] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto decode = [&](int tile, int& im, int& hh0, int& ww0) {
    const int per = a.th_n * a.tw_n;
    im = tile / per;
    const int rem = tile - im * per;
    const int ti = rem / a.tw_n;
    hh0 = ti * TH;
    ww0 = (rem - ti * a.tw_n) * TW;
  };
  // the four late waves bring the whole patch: piece = (wave - 4) + 4 i
  auto issue_patch = [&](int buf, int im, int hh0, int ww0) {
#pragma unroll 1
    for (int piece = wave - 4; piece < APIECES; piece += 4) {
      const int r = piece * 8 + (lane >> 3);
      const int pj = r / PHP, pi = r - pj * PHP;
      const int hh = hh0 - 1 + pi, ww = ww0 - 1 + pj;
      const int ch = ((lane & 7) ^ ((pj >> 1) & 7)) * VEC;
      const bool ok = r < PROWS && pi < PH && (unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)a.W && ch < a.Cin;
      const unsigned pix = a.ups ? (unsigned)((im * (a.H >> 1) + (hh >> 1)) * (a.W >> 1) + (ww >> 1))
                                 : (unsigned)((im * a.H + hh) * a.W + ww);
      const unsigned off = ok ? (pix * (unsigned)a.ldx + ch) * ES : OOB;
      dma16(xr, smem + buf * A_BYTES + piece * 1024, off);
    }
  };

  accv_t acc[MT][NT];
  auto mma = [&](const bf16x8& w8, const bf16x8& x8, accv_t& c) {
    if constexpr (S == 32) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w8, x8, c, 0, 0, 0);
    else c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w8, x8, c, 0, 0, 0);
  };
  // 9 * NKC groups as one software pipeline: group g + 1's MT reads are issued ahead of group g's MFMAs
  auto compute_tile = [&](int abuf) {
    const char* sA = smem + abuf * A_BYTES;
    // a pipeline step = (tap, K chunk, pair of M tiles): 2 reads, 2 * NT MFMAs
    // DEPTH register sets: a step's reads are issued DEPTH - 1 steps (64 MFMA cycles each) ahead of its MFMAs.
    // The two waves of a SIMD compute at different times here, so a wave must cover its LDS latency alone.
    constexpr int MP = MT / 2, NSTEP = 9 * NKC * MP;
    bf16x8 af[DEPTH][2];
    // keep the 3 * MT bases in registers and derive every other address next to its read (XOR + immediate
    // offset): hoisted out of the tile loop the 9 * NKC * MT addresses would take as many registers as the weights
    int ab[MT][3];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        ab[t][tx] = abase[t][tx];
        asm volatile("" : "+v"(ab[t][tx]));
      }
    auto load_step = [&](int sidx, int set) {
      const int g = sidx / MP, mp = sidx - g * MP;
      const int tap = g / NKC, kc = g - tap * NKC;
      const int ty = tap / 3, tx = tap - 3 * ty;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[set][i] = *reinterpret_cast<const bf16x8*>(sA + (ab[2 * mp + i][tx] ^ (kc * CPK * 16)) + ty * 128);
      }
    };
    // step s + DEPTH - 1's reads are issued before step s's MFMAs; full scheduling barriers pin that
    // order -- left alone, or with sched_group_barrier hints in a body this long, the scheduler folds the sets into
    // one and every MFMA pair waits for an LDS round trip
#pragma unroll
    for (int p = 0; p < DEPTH - 1; ++p) load_step(p, p);
#pragma unroll
    for (int sidx = 0; sidx < NSTEP; ++sidx) {
      if (sidx + DEPTH - 1 < NSTEP) load_step(sidx + DEPTH - 1, (sidx + DEPTH - 1) % DEPTH);
      __builtin_amdgcn_sched_barrier(0);
      const int g = sidx / MP, mp = sidx - g * MP;
      const int tap = g / NKC, kc = g - tap * NKC;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int u = 0; u < NT; ++u) mma(breg[tap][kc][u], af[sidx % DEPTH][i], acc[2 * mp + i][u]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // wave-local epilogue of the tile whose accumulators the wave holds
  auto epilogue = [&](int im, int hh0, int ww0) {
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int px = t * S + ls;
      char* rowp = sCw + px * 64;
      const int psw = (px >> 1) & 3;
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int q = 0; q < NACC / 4; ++q) {
          bf16x4 pk;
#pragma unroll
          for (int e = 0; e < 4; ++e) pk[e] = (bf16_t)acc[t][u][4 * q + e];
          const int c0 = u * S + ((S == 32) ? (8 * q + 4 * lq) : (4 * lq));   // first of 4 consecutive channels
          *reinterpret_cast<bf16x4*>(rowp + ((((c0 >> 3) ^ psw)) << 4) + (c0 & 4) * 2) = pk;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's own writes have executed
    Vec16<T> vb[4];
    const int cc = lane & 3;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int px = (lane >> 2) + 16 * k;
      vb[k] = *reinterpret_cast<const Vec16<T>*>(sCw + px * 64 + ((cc ^ ((px >> 1) & 3)) << 4));
    }
    const int n = n0 + wn * 32 + cc * VEC;
    float sq1[VEC], sq2[VEC];
    {
      const f32x4 s0 = sStat[0], s1 = sStat[1], s2 = sStat[2], s3 = sStat[3];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sq1[e] = s0[e];
        sq1[4 + e] = s1[e];
        sq2[e] = s2[e];
        sq2[4 + e] = s3[e];
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int m = 64 * wm + (lane >> 2) + 16 * k;
      const int pi = (TW == 32) ? (m >> 5) : (m >> 4), pj = (TW == 32) ? (m & 31) : (m & 15);
      const int hh = hh0 + pi, ww = ww0 + pj;
      if (hh < a.H && ww < a.W && n < a.Nout) {
        if (!(UZ_KFLAGS(a) & 0x1)) st16(yg + ((size_t)(im * a.H + hh) * a.W + ww) * a.ldy + n, vb[k]);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float fv = (float)vb[k].v[e];
          sq1[e] += fv;
          sq2[e] += fv * fv;
        }
      }
    }
    sStat[0] = f32x4{sq1[0], sq1[1], sq1[2], sq1[3]};
    sStat[1] = f32x4{sq1[4], sq1[5], sq1[6], sq1[7]};
    sStat[2] = f32x4{sq2[0], sq2[1], sq2[2], sq2[3]};
    sStat[3] = f32x4{sq2[4], sq2[5], sq2[6], sq2[7]};
  };

  int img = 0, h0 = 0, w0 = 0, pimg = 0, ph0 = 0, pw0 = 0;
  if (late) {
    decode(blockIdx.x, img, h0, w0);
    issue_patch(0, img, h0, w0);
  }
  wait_vmcnt<0>();   // everyone's weight registers
  int it = 0;
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x, ++it) {
    decode(tile, img, h0, w0);
    // late waves: this tile's patch pieces (issued a whole interval ago) have landed.  The early waves issued no DMA
    // and must NOT wait here: their youngest memory operations are the previous tile's output stores
    if (late) wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (late) {
      const int next = tile + gridDim.x;
      if (next < a.ntiles && !(UZ_KFLAGS(a) & 0x2)) {
        int im2, hh2, ww2;
        decode(next, im2, hh2, ww2);
        issue_patch((it + 1) & 1, im2, hh2, ww2);
      }
      if (it > 0) epilogue(pimg, ph0, pw0);
    }
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
      for (int q = 0; q < NACC / 4; ++q) {
        // accumulator registers 4q .. 4q+3 of N tile u are 4 consecutive channels
        const int c0 = wn * 32 + u * S + ((S == 32) ? (8 * q + 4 * lq) : (4 * lq));
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(sBias + c0);
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[t][u][4 * q + e] = b4[e];
      }
    compute_tile(it & 1);
    if (!late) epilogue(img, h0, w0);
    pimg = img;
    ph0 = h0;
    pw0 = w0;
  }
  if (late && it > 0) epilogue(pimg, ph0, pw0);

  if (a.stats != nullptr) {
    __syncthreads();
    const float* red = reinterpret_cast<const float*>(smem + OFF_STAT);  // [512][2 * VEC]
    if (tid < 128) {  // (which, channel): the 64 threads (4 wm x 16 lanes) that own this channel's chunk, fixed order
      const int which = tid >> 6, ch = tid & 63;
      const int cwn = ch >> 5, cc = (ch & 31) >> 3, e = ch & 7;
      float t = 0.f;
      for (int k = 0; k < 4; ++k)
        for (int l = 0; l < 16; ++l) {
          const int th = ((k * 2 + cwn) << 6) + l * 4 + cc;
          t += red[th * 2 * VEC + which * VEC + e];
        }
      if (n0 + ch < a.Nout) a.stats[((size_t)blockIdx.x * 2 + which) * a.Nout + n0 + ch] = t;
    }
  }
}

}  // namespace

// ---- host side ------------------------------------------------------------------------------
// returns 1 and fills the plan when the direct kernel applies to this descriptor, 0 otherwise
int uz_direct_plan(const uz_conv_desc* d, UzDirectPlan* p) {
  const bool up = d->taps_mode == UZ_TAPS_CONV_UP2;
  if (!(d->taps_mode == UZ_TAPS_CONV || up) || d->ntaps != 9 || d->dil != 1 || d->store_mode != UZ_STORE_PLAIN) return 0;
  if (up && ((d->H & 1) || (d->W & 1) || d->Hin * 2 != d->H || d->Win * 2 != d->W)) return 0;
  const int vec = d->dtype == UZ_BF16 ? 8 : 4, es = d->dtype == UZ_BF16 ? 2 : 4, bk = 8 * vec;
  if (d->Cin % vec != 0 || d->Nout % vec != 0 || d->ldy % vec != 0) return 0;
  const long long xbytes = ((long long)d->N * d->Hin * d->Win - 1) * d->ldx * es + (long long)d->Cin * es;
  const long long wbytes = (long long)d->Nout * 9 * d->Cin * es;
  if (xbytes >= (1LL << 31) || wbytes >= (1LL << 31)) return 0;
  {
    UzPpPlan pp;   // third generation (uz_conv3x3_pp.hip): bf16, input channels in multiples of 32
    if (uz_pp_plan(d, &pp)) {
      p->tw = (pp.cfg == UZ_PP_256W16 || pp.cfg == UZ_PP_128W16) ? 16 : 32;
      p->bn = pp.bn;
      p->bres = 3;
      p->ppcfg = pp.cfg;
      p->th_n = pp.th_n;
      p->tw_n = pp.tw_n;
      p->ntiles = pp.ntiles;
      p->tiles_n = pp.tiles_n;
      p->grid_m = pp.grid_m;
      p->ksplit = pp.ksplit;
      p->cps = pp.cps;
      return 1;
    }
  }
  p->tw = d->W >= 32 ? 32 : 16;
  const int th = 256 / p->tw;
  p->th_n = (d->H + th - 1) / th;
  p->tw_n = (d->W + p->tw - 1) / p->tw;
  p->ntiles = d->N * p->th_n * p->tw_n;
  p->bn = d->Nout <= 64 ? 64 : 128;
  if (p->bn == 128 && (long long)p->ntiles * ((d->Nout + 127) / 128) <= UZ_NUM_CU / 2) p->bn = 64;
  p->bres = (p->bn == 64 && d->Cin == bk) ? 1 : 0;
  // bf16 with at most one 64-channel slab of input: weights in registers, 64 output channels per workgroup
  // (conv3x3_res64_kernel); flag 0x40000000 of the ablation build keeps the first-generation kernels
  // (64 -> 64 exactly stays on the LDS-resident kernel above: 88-91 us against 90-94 us at 256^2; the register
  // kernel wins where that one does not apply: 64 -> 128 at 128^2 62 -> 49 us, Cin < 64, Nout > 64)
  if (d->dtype == UZ_BF16 && d->Cin <= 64 && !(uz_tune_flags() & 0x40000000) &&
      (!(d->Cin == 64 && d->Nout <= 64) || (uz_tune_flags() & 0x20000000))) {
    p->bn = 64;
    p->bres = 2;
  }
  p->tiles_n = (d->Nout + p->bn - 1) / p->bn;
  int cap = UZ_NUM_CU / p->tiles_n;
  if (cap < 1) cap = 1;
  p->grid_m = p->ntiles < cap ? p->ntiles : cap;
  return 1;
}

template <typename T>
static int direct_launch_t(const UzDirectPlan& p, const DirectArgs& a, hipStream_t s) {
  dim3 grid(p.grid_m, p.tiles_n), block(512);
  if constexpr (sizeof(T) == 2) {
    if (p.bres == 2) {
      // measured (tools/kbench.py, 64 -> 64 at 256^2 and 64 -> 128 at 128^2): the 16x16x32 stream is no faster than
      // 32x32x16 here (the kernel is bound by LDS-DMA issue and LDS latency, not by the matrix clock), and deeper
      // read pipelines (3, 4 register sets) are slower (registers); the ablation build keeps the variants
      const bool s16 = (a.flags & 0x10000000) != 0;
      const int depth = (a.flags & 0x8000000) ? 3 : 2;
#define UZ_RES64(TWv, Sv, Dv) hipLaunchKernelGGL((conv3x3_res64_kernel<TWv, Sv, Dv>), grid, block, 0, s, a)
      if (p.tw == 32) {
#ifdef UZ_ABLATE
        if (s16) { if (depth == 3) UZ_RES64(32, 16, 3); else UZ_RES64(32, 16, 2); }
        else if (depth == 3) UZ_RES64(32, 32, 3);
        else
#endif
          UZ_RES64(32, 32, 2);
      } else {
        UZ_RES64(16, 32, 2);
      }
#undef UZ_RES64
      UZ_LAUNCH_CHECK("uz_conv_igemm(direct3x3 res64)");
      return UZ_OK;
    }
  }
  if constexpr (sizeof(T) == 2) {
    if (a.bn_y != nullptr) {
#define UZ_BNRED(TWv, BNv, RESv) hipLaunchKernelGGL((conv3x3_direct_kernel<T, TWv, BNv, RESv, true>), grid, block, 0, s, a)
      if (p.bn == 64 && p.bres) { if (p.tw == 32) UZ_BNRED(32, 64, true); else UZ_BNRED(16, 64, true); }
      else if (p.bn == 64) { if (p.tw == 32) UZ_BNRED(32, 64, false); else UZ_BNRED(16, 64, false); }
      else { if (p.tw == 32) UZ_BNRED(32, 128, false); else UZ_BNRED(16, 128, false); }
#undef UZ_BNRED
      UZ_LAUNCH_CHECK("uz_conv_igemm_bnred(direct3x3)");
      return UZ_OK;
    }
  }
  if (p.bn == 64) {
    if (p.bres) {
      if (p.tw == 32) hipLaunchKernelGGL((conv3x3_direct_kernel<T, 32, 64, true>), grid, block, 0, s, a);
      else hipLaunchKernelGGL((conv3x3_direct_kernel<T, 16, 64, true>), grid, block, 0, s, a);
    } else {
      if (p.tw == 32) hipLaunchKernelGGL((conv3x3_direct_kernel<T, 32, 64, false>), grid, block, 0, s, a);
      else hipLaunchKernelGGL((conv3x3_direct_kernel<T, 16, 64, false>), grid, block, 0, s, a);
    }
  } else {
    if (p.tw == 32) hipLaunchKernelGGL((conv3x3_direct_kernel<T, 32, 128, false>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((conv3x3_direct_kernel<T, 16, 128, false>), grid, block, 0, s, a);
  }
  UZ_LAUNCH_CHECK("uz_conv_igemm(direct3x3)");
  return UZ_OK;
}

int uz_direct_launch(const uz_conv_desc* d, const UzDirectPlan& p, const void* x, const void* w,
                     const float* bias, void* y, float* stats, hipStream_t s, const UzBnRed* br, float* part,
                     const UzXf* xf) {
  const int es = d->dtype == UZ_BF16 ? 2 : 4;
  if (p.bres == 3) {
    UzPpPlan pp = {p.ppcfg, p.bn, p.th_n, p.tw_n, p.ntiles, p.tiles_n, p.grid_m, p.ksplit, p.cps};
    return uz_pp_launch(d, pp, x, w, bias, y, stats, s, br, part, xf);
  }
  UZ_REQUIRE(xf == nullptr, "uz_conv_igemm_xf: the input transform is the ping-pong kernel's");
  UZ_REQUIRE(part == nullptr, "uz_conv_igemm(direct3x3): split-K is the ping-pong kernel's");
  DirectArgs a;
  a.bn_y = br ? br->y : nullptr;
  a.bn_scale = br ? br->scale : nullptr;
  a.bn_shift = br ? br->shift : nullptr;
  a.bn_mean = br ? br->mean : nullptr;
  a.bn_invstd = br ? br->invstd : nullptr;
  a.ld_bny = br ? br->ldy : 0;
  if (br) UZ_REQUIRE(d->dtype == UZ_BF16 && p.bres != 2 && stats != nullptr,
                     "uz_conv_igemm_bnred: bf16 direct 3x3 kernels with LDS-staged epilogue only");
  a.x = x;
  a.w = w;
  a.y = y;
  a.bias = bias;
  a.stats = stats;
  a.xbytes = (unsigned)(((long long)d->N * d->Hin * d->Win - 1) * d->ldx * es + (long long)d->Cin * es);
  {
    const long long yb = ((long long)d->N * d->H * d->W - 1) * d->ldy * es + (long long)d->Nout * es;
    a.ybytes = yb < (1LL << 31) ? (unsigned)yb : 0u;
  }
  a.ups = d->taps_mode == UZ_TAPS_CONV_UP2 ? 1 : 0;
  a.wbytes = (unsigned)((long long)d->Nout * 9 * d->Cin * es);
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
  a.flags = uz_tune_flags();
  return d->dtype == UZ_BF16 ? direct_launch_t<bf16_t>(p, a, s) : direct_launch_t<float>(p, a, s);
}
