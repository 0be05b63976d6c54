// Weight gradient of ConvTranspose2d(k = 2, s = 2) -- and of every "2 x 2 gather" product (PatchExpand) -- bf16, gfx950:
//     dW[ci][co][dy][dx] = sum_p x[p][ci] * g[(2h + dy, 2w + dx)][co]          (p = (n, h, w) on the coarse grid)
// (autograd weight gradient of nn.ConvTranspose2d, reference unet_zoo/models/common_layers.py:104, entered from
// loss.backward(), unet_zoo/utils/training_loop.py:119).
//
// The round-3 route (uz_wgrad3x3.hip, gather mode) is four one-tap problems on blockIdx.y: the coarse operand x is
// fetched once per tap (216 MB read per launch against 94 MB algorithmic, profiles/r03_pmc_traffic.json: 191 us where
// ~90 would do).  Here a workgroup owns all FOUR taps of a (BI x-channels) x (64 g-channels) tile: a step = 64 coarse
// pixels; its x tile and the four "planes" of g (plane (dy, dx) = the fine pixels (2h + dy, 2w + dx), de-interleaved by
// the DMA's per-lane source address so that a plane is an ordinary pixel-major tile) enter LDS once, every byte of x
// and of g is read from memory exactly once.  The kernel is bound by streaming g (a step brings 48 KB for 4.2 MFLOP), so
// the schedule is the simple one: three stages, one barrier per step at the step's start, the request for step s + 2
// leaves right behind it.  Fragments through ds_read_b64_tr_b16 as in uz_wgrad9.hip; one fp32 slab [split][tap][Ci][Cj]
// per pixel split, summed in fixed order by uz_wgrad's reduce kernel (bitwise reproducible).
#include "uz_common.h"

namespace {

struct WgG4Args {
  const void* L;
  const void* R;
  float* slab;
  unsigned lbytes, rbytes;
  int H, W, Ci, ldl, Cj, ldr;
  int Hr, Wr;            // the fine grid (2H or 2H + 1, 2W or 2W + 1)
  int nstrips, hsteps;   // column strips per coarse row (W / KW), steps per (image, strip) (H / G)
  int units, upb;
  int tiles_j, ntiles, split;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(3))) char* lds_char_ptr;
template <int V> struct IntC { static constexpr int value = V; };

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt range");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
template <int OFF, int OFF0>
__device__ __forceinline__ void tr_pair(bf16x4& lo, bf16x4& hi, unsigned lds_addr) {
  static_assert(OFF0 >= 0 && OFF0 + OFF < 65536, "ds offset");
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
               : "=&v"(lo), "=&v"(hi)
               : "v"(lds_addr), "i"(OFF0), "i"(OFF0 + OFF));
}
__device__ __forceinline__ void pin(bf16x4& v) { asm volatile("" : "+v"(v)); }

// BI x-channels x 64 g-channels x four taps per workgroup, NW = BI / 16 waves of 32 x 32 channel tiles; KW-wide strips
template <int BI, int KW>
__global__ __launch_bounds__(BI * 4, 1) void wgrad_g4_kernel(const WgG4Args a) {
  constexpr int NW = BI / 16;
  constexpr int G = 64 / KW;                     // coarse rows per step
  constexpr int RBL = BI * 2, CPRL = RBL / 16, RPPL = 1024 / RBL, NLP = 64 / RPPL;
  constexpr int KL = NLP / NW;                   // x pieces per wave and step
  constexpr int KR = 8 / NW > 0 ? 8 / NW : 1;    // g pieces per wave and plane (a plane = 64 pixels x 128 B = 8 pieces)
  constexpr int LSTAGE = 64 * RBL, PLANE = 64 * 128, STAGE = LSTAGE + 4 * PLANE, NSL = 3;
  constexpr int PPS = KL + 4 * KR;               // pieces per wave and step
  constexpr unsigned ROW_OOB = 0x80000000u;
  static_assert(NLP % NW == 0 && 8 % NW == 0 && NSL * STAGE <= 160 * 1024 && PPS <= 63, "configuration");
  __shared__ __attribute__((aligned(1024))) char smem[NSL * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave >> 1, wj = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  int tile, z;
  {
    const int id = blockIdx.x;
    if ((a.split & 7) == 0) {   // the tiles of one pixel split share an XCD (speed only)
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
  const unsigned ldl2 = (unsigned)(a.ldl * 2), ldr2 = (unsigned)(a.ldr * 2);

  // DMA tables (per lane): byte offset of its 16 bytes relative to the step's first coarse pixel (x) / to the fine pixel
  // (2h, 2 w0) (g; the plane adds (dy Wr + dx) pixels).  64-byte granules XOR-swizzled with the pixel index as in
  // uz_wgrad9.hip.  Channel tails: out of range.
  unsigned ltab[KL], gtab[KR];
#pragma unroll
  for (int k = 0; k < KL; ++k) {
    const int kpx = (wave + NW * k) * RPPL + lane / CPRL, pc = lane % CPRL;
    const int r = kpx / KW, c = kpx % KW;
    const int sw = CPRL == 16 ? (kpx & 3) : ((kpx >> 1) & 1);
    const int lchunk = (((pc >> 2) ^ sw) << 2) + (pc & 3);
    ltab[k] = (ti0 + lchunk * 8 < a.Ci) ? (unsigned)(r * a.W + c) * ldl2 + (unsigned)(ti0 * 2 + lchunk * 16) : ROW_OOB;
  }
#pragma unroll
  for (int k = 0; k < KR; ++k) {
    const int i = (wave + NW * k) * 8 + (lane >> 3), pc = lane & 7;   // pixel i of the plane
    const int r = i / KW, c = i % KW;
    const int sw = (i >> 1) & 1;
    const int lchunk = (((pc >> 2) ^ sw) << 2) + (pc & 3);
    gtab[k] = (tj0 + lchunk * 8 < a.Cj) ? (unsigned)(2 * r * a.Wr + 2 * c) * ldr2 + (unsigned)(tj0 * 2 + lchunk * 16) : ROW_OOB;
  }

  int c_img, c_strip, c_hb;
  {
    const int per_img = a.nstrips * a.hsteps;
    c_img = u_beg / per_img;
    const int rem = u_beg - c_img * per_img;
    c_strip = rem / a.hsteps;
    c_hb = rem - c_strip * a.hsteps;
  }
  int i_t = 0, i_ls = 0;
  auto issue_batch = [&]() __attribute__((always_inline)) {
    const bool valid = i_t < nu;
    const unsigned h = (unsigned)(c_hb * G), w0 = (unsigned)(c_strip * KW);
    const unsigned lbase = valid ? (((unsigned)c_img * a.H + h) * a.W + w0) * ldl2 : ROW_OOB;
    const unsigned gpix = ((unsigned)c_img * a.Hr + 2 * h) * a.Wr + 2 * w0;
    char* dst = smem + i_ls * STAGE + wave * 1024;
#pragma unroll
    for (int k = 0; k < KL; ++k) {
      const unsigned voff = ltab[k] + lbase;   // (a variable of its own: see uz_wgrad9.hip)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(lr, (lds_ptr_t)(dst + k * (NW * 1024)), 16, voff, 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const unsigned gbase = valid ? (gpix + (unsigned)((t >> 1) * a.Wr + (t & 1))) * ldr2 : ROW_OOB;
#pragma unroll
      for (int k = 0; k < KR; ++k) {
        const unsigned voff = gtab[k] + gbase;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rr, (lds_ptr_t)(dst + LSTAGE + t * PLANE + k * (NW * 1024)), 16, voff, 0, 0, 0);
      }
    }
    if (++c_hb == a.hsteps) {
      c_hb = 0;
      if (++c_strip == a.nstrips) {
        c_strip = 0;
        ++c_img;
      }
    }
    ++i_t;
    i_ls = (i_ls + 1 == NSL) ? 0 : i_ls + 1;
  };

  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int lk = 8 * (g >> 1) + q4, lcol = 16 * (g & 1) + 4 * p4;
  const unsigned aoff = (unsigned)(lk * RBL + ((wi ^ (CPRL == 16 ? q4 : (q4 >> 1))) << 6) + lcol * 2);
  const unsigned boff = (unsigned)(LSTAGE + lk * 128 + ((wj ^ ((q4 >> 1) & 1)) << 6) + lcol * 2);

  // unit U = (16-pixel sub-step ks = U / 4, tap t = U % 4): one MFMA per wave; fragments requested two units ahead
  bf16x4 alo[2], ahi[2], blo[3], bhi[3];
  unsigned va = 0, vb = 0;
  auto fetch = [&](auto UC) __attribute__((always_inline)) {
    constexpr int U = decltype(UC)::value, ks = U / 4, t = U % 4;
    if constexpr (t == 0) tr_pair<4 * RBL, ks * 16 * RBL>(alo[ks & 1], ahi[ks & 1], va);
    tr_pair<4 * 128, t * PLANE + ks * 16 * 128>(blo[U % 3], bhi[U % 3], vb);
  };
  auto compute = [&](auto UC) __attribute__((always_inline)) {
    constexpr int U = decltype(UC)::value, ks = U / 4, t = U % 4;
    if constexpr (t == 0) {
      pin(alo[ks & 1]);
      pin(ahi[ks & 1]);
    }
    pin(blo[U % 3]);
    pin(bhi[U % 3]);
    const bf16x8 afr = __builtin_shufflevector(alo[ks & 1], ahi[ks & 1], 0, 1, 2, 3, 4, 5, 6, 7);
    const bf16x8 bfr = __builtin_shufflevector(blo[U % 3], bhi[U % 3], 0, 1, 2, 3, 4, 5, 6, 7);
    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr, acc[t], 0, 0, 0);
  };
#define UZ_G4_NREADS(U) (((U) % 4 == 0) ? 4 : 2)
#define UZ_G4_UNIT(U)                                                  \
  do {                                                                 \
    fetch(IntC<(U) + 2>{});                                            \
    wait_lgkm<UZ_G4_NREADS((U) + 1) + UZ_G4_NREADS((U) + 2)>();        \
    compute(IntC<(U)>{});                                              \
    __builtin_amdgcn_sched_barrier(0);                                 \
  } while (0)

  issue_batch();
  issue_batch();
  int m_ls = 0;
#pragma unroll 1
  for (int s = 0; s < nu; ++s) {
    wait_vmcnt<PPS>();   // step s has landed (step s + 1 may still be in flight)
    __builtin_amdgcn_s_barrier();
    issue_batch();       // step s + 2 into the stage step s - 1 has left
    va = smem_u + m_ls * STAGE + aoff;
    vb = smem_u + m_ls * STAGE + boff;
    m_ls = (m_ls + 1 == NSL) ? 0 : m_ls + 1;
    fetch(IntC<0>{});
    fetch(IntC<1>{});
    UZ_G4_UNIT(0);
    UZ_G4_UNIT(1);
    UZ_G4_UNIT(2);
    UZ_G4_UNIT(3);
    UZ_G4_UNIT(4);
    UZ_G4_UNIT(5);
    UZ_G4_UNIT(6);
    UZ_G4_UNIT(7);
    UZ_G4_UNIT(8);
    UZ_G4_UNIT(9);
    UZ_G4_UNIT(10);
    UZ_G4_UNIT(11);
    UZ_G4_UNIT(12);
    UZ_G4_UNIT(13);
    wait_lgkm<2>();
    compute(IntC<14>{});
    wait_lgkm<0>();
    compute(IntC<15>{});
    __builtin_amdgcn_sched_barrier(0);
  }
#undef UZ_G4_UNIT
#undef UZ_G4_NREADS
  wait_vmcnt<0>();

  const int cj = tj0 + wj * 32 + l31;
  const int ci0 = ti0 + wi * 32 + 4 * lh;
  float* slab0 = a.slab + (size_t)z * 4 * (size_t)a.Ci * a.Cj + cj;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = ci0 + (r & 3) + 8 * (r >> 2);
      if (ci < a.Ci && cj < a.Cj) slab0[((size_t)t * a.Ci + ci) * a.Cj] = acc[t][r];
    }
}

}  // namespace

int uz_wgrad_g4_plan(const uz_wgrad_desc* d, UzWgrad2Plan* p) {
  if (uz_tune_flags() & 0x8000000) return 0;   // ablation build: the round-3 kernels
  if (d->dtype != UZ_BF16 || d->taps_mode != UZ_TAPS_GATHER2X2 || d->ntaps != 4) return 0;
  if (d->Ci % 8 != 0 || d->Cj % 8 != 0) return 0;
  const int W = d->W, H = d->H;
  if (!(W == 16 || W == 32 || (W >= 64 && W % 64 == 0))) return 0;
  const int kw = W < 64 ? W : 64, g = 64 / kw;
  if (H % g != 0) return 0;
  const long long lbytes = ((long long)d->N * d->H * d->W - 1) * d->ldl * 2 + (long long)d->Ci * 2;
  const long long rbytes = ((long long)d->N * d->Hr * d->Wr - 1) * d->ldr * 2 + (long long)d->Cj * 2;
  if (lbytes >= (1LL << 31) || rbytes >= (1LL << 31)) return 0;
  const long long units = (long long)d->N * (W / kw) * (H / g);
  if (units >= (1LL << 30)) return 0;
  const int bi = d->Ci > 64 ? 128 : 64;
  p->v9 = 2;
  p->bi = bi;
  p->kw = kw;
  p->kr = g;
  p->H = H;
  p->W = W;
  p->big = 0;
  p->one_tap = 1;
  p->gather = 1;
  p->wide9 = 0;
  p->kg = 1;
  p->tiles_i = (d->Ci + bi - 1) / bi;
  p->tiles_j = (d->Cj + 63) / 64;
  p->units = (int)units;
  const long long ntiles = (long long)p->tiles_i * p->tiles_j;
  long long split = ntiles >= UZ_NUM_CU ? 1 : UZ_NUM_CU / ntiles;
  const long long max_split = units / 4 > 0 ? units / 4 : 1;
  if (split > max_split) split = max_split;
  if (split > 8) split -= split % 8;
  if (split < 1) split = 1;
  p->upb = (int)((units + split - 1) / split);
  p->split = (int)((units + p->upb - 1) / p->upb);
  p->nslabs = p->split;
  return 1;
}

int uz_wgrad_g4_launch(const uz_wgrad_desc* d, const UzWgrad2Plan& p, const void* L, const void* R, float* slab, hipStream_t s) {
  WgG4Args a;
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
  const dim3 grid((unsigned)(a.ntiles * p.split));
  if (p.bi == 128) {
    if (p.kw == 64) hipLaunchKernelGGL((wgrad_g4_kernel<128, 64>), grid, dim3(512), 0, s, a);
    else if (p.kw == 32) hipLaunchKernelGGL((wgrad_g4_kernel<128, 32>), grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((wgrad_g4_kernel<128, 16>), grid, dim3(512), 0, s, a);
  } else {
    if (p.kw == 64) hipLaunchKernelGGL((wgrad_g4_kernel<64, 64>), grid, dim3(256), 0, s, a);
    else if (p.kw == 32) hipLaunchKernelGGL((wgrad_g4_kernel<64, 32>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((wgrad_g4_kernel<64, 16>), grid, dim3(256), 0, s, a);
  }
  UZ_LAUNCH_CHECK("uz_wgrad(gather 2x2, four taps)");
  return UZ_OK;
}
