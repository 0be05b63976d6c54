// Bias gradients of a whole backward (phase) in two launches: out_i[c] = sum_p x_i[p][c] for up to 96 tensors per
// launch pair.  A token model pays one column sum per nn.Linear / strided convolution with a bias (missformer: 160
// per step, swin_unet_v2: 28); as separate two-stage reductions they are ~11 us of launch latency each.  The
// descriptors travel by value in the kernel arguments (no table upload, capturable in a hipGraph); every tensor
// keeps its own fixed partition into row blocks and a fixed summation order, so results are bitwise reproducible.
#include "uz_common.h"

namespace {

constexpr int CSB_MAX = 80;   // items per launch pair: 80 x 48 B + header < the 4 KB kernel-argument limit

struct CsItem {
  const void* x;
  float* out;
  int P, C, ld, gx;     // gx: row blocks (partial rows) of this tensor
  int ccb, gy;          // channel chunks per workgroup (power of two <= 64), chunk groups
  int blk0, fin0;       // first workgroup of this item in the reduce / finalize launch
  int ws0, pad_;        // first float of its partial rows
};
struct CsBatch {
  CsItem it[CSB_MAX];
  int n, pad_[3];
};

template <typename T> __device__ __forceinline__ void load_f(const T* p, float* f) {
  const Vec16<T> v = ld16(p);
#pragma unroll
  for (int i = 0; i < ElemTraits<T>::VEC; ++i) f[i] = (float)v.v[i];
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_batched_kernel(const CsBatch b, float* __restrict__ ws) {
  constexpr int VEC = ElemTraits<T>::VEC;
  __shared__ __attribute__((aligned(16))) float red[256 * VEC];
  int i = 0;
  {   // blk0 is increasing: binary search (a linear walk is one dependent scalar load from the argument segment per item)
    int lo = 0, hi = b.n;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if ((int)blockIdx.x >= b.it[mid].blk0) lo = mid;
      else hi = mid;
    }
    i = lo;
  }
  const CsItem& it = b.it[i];
  const int local = blockIdx.x - it.blk0;
  const int bx = local % it.gx, by = local / it.gx;
  const int ccb = it.ccb, pr = 256 / ccb;
  const int cx = threadIdx.x % ccb, ry = threadIdx.x / ccb;
  const int CC = it.C / VEC, cc = by * ccb + cx;
  const bool cok = cc < CC;
  const T* x = static_cast<const T*>(it.x) + (cok ? cc : 0) * VEC;
  float s[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) s[e] = 0.f;
  const long long stride = (long long)it.gx * pr;
  for (long long p = (long long)bx * pr + ry; p < it.P; p += 4 * stride) {
    // four rows in flight: clamped address + select, the loads issue together
    Vec16<T> raw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long long q = p + k * stride;
      raw[k] = ld16(x + (size_t)(q < it.P ? q : it.P - 1) * it.ld);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool ok = cok && p + k * stride < it.P;
#pragma unroll
      for (int e = 0; e < VEC; ++e) s[e] += ok ? (float)raw[k].v[e] : 0.f;
    }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) red[(ry * ccb + cx) * VEC + e] = s[e];
  __syncthreads();
  // thread (cx, e) adds the pr rows in order
  for (int e = ry; e < VEC; e += pr) {
    float t = 0.f;
    for (int r = 0; r < pr; ++r) t += red[(r * ccb + cx) * VEC + e];
    if (cok) ws[(size_t)it.ws0 + (size_t)bx * it.C + cc * VEC + e] = t;
  }
}

// one workgroup per (item, 32 columns): 32 row groups x 32 columns, then a fixed sum over the groups
__global__ __launch_bounds__(1024) void colsum_batched_finalize_kernel(const CsBatch b, const float* __restrict__ ws) {
  __shared__ double sh[32][33];
  int i = 0;
  {   // fin0 is increasing: binary search (a linear walk is one dependent scalar load from the argument segment per item)
    int lo = 0, hi = b.n;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if ((int)blockIdx.x >= b.it[mid].fin0) lo = mid;
      else hi = mid;
    }
    i = lo;
  }
  const CsItem& it = b.it[i];
  const int el = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int c = (blockIdx.x - it.fin0) * 32 + el;
  double s = 0.0;
  {   // eight rows per trip, unconditional loads (row 0 / column 0 out of range, dropped), same order of additions
    const bool in = c < it.C;
    const float* base = ws + (size_t)it.ws0 + (in ? c : 0);
    for (int r = g; r < it.gx; r += 8 * 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = base[(size_t)(r + u * 32 < it.gx ? r + u * 32 : 0) * it.C];
#pragma unroll
      for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(v[u]));
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (in && r + u * 32 < it.gx) s += (double)v[u];
    }
  }
  sh[g][el] = s;
  __syncthreads();
  if (g == 0 && c < it.C) {
    double t = 0.0;
    for (int r = 0; r < 32; ++r) t += sh[r][el];
    it.out[c] = (float)t;
  }
}

// fills the launch geometry of items[first, first + n); returns the number of partial floats
long long cs_plan(int dtype, const uz_colsum_item* items, int n, CsBatch* b, int* blocks, int* fblocks) {
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  long long ws = 0;
  int blk = 0, fin = 0;
  b->n = n;
  for (int i = 0; i < n; ++i) {
    CsItem& it = b->it[i];
    it.x = items[i].x;
    it.out = items[i].out;
    it.P = items[i].P;
    it.C = items[i].C;
    it.ld = items[i].ld;
    const int CC = it.C / vec;
    int ccb = 1;
    while (ccb < CC && ccb < 64) ccb <<= 1;
    it.ccb = ccb;
    it.gy = (CC + ccb - 1) / ccb;
    const int pr = 256 / ccb;
    long long gx = ((long long)it.P + (long long)pr * 16 - 1) / ((long long)pr * 16);   // >= 16 rows per thread
    const long long cap = 2LL * UZ_NUM_CU / it.gy > 0 ? 2LL * UZ_NUM_CU / it.gy : 1;
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    it.gx = (int)gx;
    it.blk0 = blk;
    it.fin0 = fin;
    it.ws0 = (int)ws;
    it.pad_ = 0;
    blk += it.gx * it.gy;
    fin += (it.C + 31) / 32;
    ws += gx * it.C;
  }
  *blocks = blk;
  *fblocks = fin;
  return ws;
}

int cs_check(const char* fn, int dtype, const uz_colsum_item* items, int n) {
  UZ_REQUIRE(dtype == UZ_F32 || dtype == UZ_BF16, "%s: bad dtype", fn);
  UZ_REQUIRE(items != nullptr && n > 0, "%s: no items", fn);
  const int vec = dtype == UZ_BF16 ? 8 : 4;
  for (int i = 0; i < n; ++i)
    UZ_REQUIRE(items[i].x && items[i].out && items[i].P > 0 && items[i].C > 0 && items[i].C % vec == 0 &&
                   items[i].ld % vec == 0 && items[i].ld >= items[i].C && ((uintptr_t)items[i].x & 15) == 0,
               "%s: bad item %d (P=%d C=%d ld=%d)", fn, i, items[i].P, items[i].C, items[i].ld);
  return UZ_OK;
}

}  // namespace

// ---- row sums of many partial buffers (LayerNorm dgamma | dbeta, depthwise weight gradients) in one launch -------
namespace {
constexpr int SRB_MAX = 80;
struct SrItem {
  const float* partial;
  float *out0, *out1;
  int rows, n, n0, blk0;
};
struct SrBatch {
  SrItem it[SRB_MAX];
  int n, pad_[3];
};

// the arithmetic of sum_rows_f32_kernel<32>: 32 row groups x 32 columns per workgroup, double accumulation
__global__ __launch_bounds__(1024) void sum_rows_batched_kernel(const SrBatch b) {
  __shared__ double sh[32][33];
  int i = 0;
  {   // blk0 is increasing: binary search (see colsum_batched_finalize_kernel)
    int lo = 0, hi = b.n;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if ((int)blockIdx.x >= b.it[mid].blk0) lo = mid;
      else hi = mid;
    }
    i = lo;
  }
  const SrItem& it = b.it[i];
  const int el = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int e = (blockIdx.x - it.blk0) * 32 + el;
  double s = 0.0;
  {   // eight rows per trip, unconditional loads, same order of additions
    const bool in = e < it.n;
    const float* base = it.partial + (in ? e : 0);
    for (int r = g; r < it.rows; r += 8 * 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = base[(size_t)(r + u * 32 < it.rows ? r + u * 32 : 0) * it.n];
#pragma unroll
      for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(v[u]));
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (in && r + u * 32 < it.rows) s += (double)v[u];
    }
  }
  sh[g][el] = s;
  __syncthreads();
  if (g == 0 && e < it.n) {
    double t = 0.0;
    for (int r = 0; r < 32; ++r) t += sh[r][el];
    if (e < it.n0) it.out0[e] = (float)t;
    else it.out1[e - it.n0] = (float)t;
  }
}
// the wide form of uz_sum_rows_f32 (few rows, many columns: window attention's d(bias) | d(tau) rows) for many buffers
template <int RG>
__global__ __launch_bounds__(256) void sum_rows_wide_batched_kernel(const SrBatch b) {
  int i = 0;
  {   // blk0 is increasing: binary search (a linear walk is one dependent scalar load from the argument segment per item)
    int lo = 0, hi = b.n;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if ((int)blockIdx.x >= b.it[mid].blk0) lo = mid;
      else hi = mid;
    }
    i = lo;
  }
  const SrItem& it = b.it[i];
  uz_sum_rows_wide_body<RG>(it.partial, it.n, it.rows, it.n, it.out0, it.n0, it.out1, (int)blockIdx.x - it.blk0);
}
}  // namespace

extern "C" int uz_sum_rows_f32_batched(const uz_sum_rows_item* items, int n, void* stream) {
  UZ_REQUIRE(items != nullptr && n > 0, "uz_sum_rows_f32_batched: no items");
  for (int i = 0; i < n; ++i)
    UZ_REQUIRE(items[i].partial && items[i].out0 && items[i].rows > 0 && items[i].n > 0 && items[i].n0 >= 0 &&
                   items[i].n0 <= items[i].n && (items[i].out1 || items[i].n0 == items[i].n),
               "uz_sum_rows_f32_batched: bad item %d", i);
  hipStream_t s = (hipStream_t)stream;
  // three classes, each the arithmetic uz_sum_rows_f32 would use for the buffer on its own: 0 = 32 row groups x 32 columns,
  // 16 / 4 = the wide form with that many row groups; one launch per class and SRB_MAX buffers
  for (int cls : {0, 16, 4}) {
    SrBatch b;
    b.n = 0;
    int blk = 0;
    auto flush = [&]() -> int {
      if (b.n == 0) return UZ_OK;
      if (cls == 0) hipLaunchKernelGGL(sum_rows_batched_kernel, dim3(blk), dim3(1024), 0, s, b);
      else if (cls == 16) hipLaunchKernelGGL(sum_rows_wide_batched_kernel<16>, dim3(blk), dim3(256), 0, s, b);
      else hipLaunchKernelGGL(sum_rows_wide_batched_kernel<4>, dim3(blk), dim3(256), 0, s, b);
      UZ_LAUNCH_CHECK("uz_sum_rows_f32_batched");
      b.n = 0;
      blk = 0;
      return UZ_OK;
    };
    for (int i = 0; i < n; ++i) {
      const uz_sum_rows_item& src = items[i];
      if (uz_sum_rows_wide_rg(src.partial, src.n, src.rows, src.n) != cls) continue;
      SrItem& d = b.it[b.n++];
      d.partial = src.partial;
      d.out0 = src.out0;
      d.out1 = src.out1;
      d.rows = src.rows;
      d.n = src.n;
      d.n0 = src.n0;
      d.blk0 = blk;
      blk += cls == 0 ? (src.n + 31) / 32 : uz_cdiv(src.n, cls == 16 ? 64 : 256);
      if (b.n == SRB_MAX) {
        const int rc = flush();
        if (rc != UZ_OK) return rc;
      }
    }
    const int rc = flush();
    if (rc != UZ_OK) return rc;
  }
  return UZ_OK;
}

extern "C" long long uz_colsum_batched_workspace_bytes(int dtype, const uz_colsum_item* items, int n) {
  if (cs_check("uz_colsum_batched_workspace_bytes", dtype, items, n) != UZ_OK) return -1;
  long long most = 0;
  for (int first = 0; first < n; first += CSB_MAX) {
    CsBatch b;
    int blocks, fblocks;
    const long long w = cs_plan(dtype, items + first, n - first < CSB_MAX ? n - first : CSB_MAX, &b, &blocks, &fblocks);
    if (w >= (1LL << 31)) {
      uz_set_error("uz_colsum_batched_workspace_bytes: partial rows exceed 2^31 floats");
      return -1;
    }
    most = w > most ? w : most;
  }
  return most * (long long)sizeof(float);
}

extern "C" int uz_colsum_batched(int dtype, const uz_colsum_item* items, int n, void* workspace, void* stream) {
  const int rc = cs_check("uz_colsum_batched", dtype, items, n);
  if (rc != UZ_OK) return rc;
  UZ_REQUIRE(workspace != nullptr, "uz_colsum_batched: null workspace");
  hipStream_t s = (hipStream_t)stream;
  for (int first = 0; first < n; first += CSB_MAX) {   // launch pairs run in order on the stream: the workspace is reused
    CsBatch b;
    int blocks, fblocks;
    cs_plan(dtype, items + first, n - first < CSB_MAX ? n - first : CSB_MAX, &b, &blocks, &fblocks);
    if (dtype == UZ_BF16) hipLaunchKernelGGL((colsum_batched_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, b, (float*)workspace);
    else hipLaunchKernelGGL((colsum_batched_kernel<float>), dim3(blocks), dim3(256), 0, s, b, (float*)workspace);
    UZ_LAUNCH_CHECK("uz_colsum_batched");
    hipLaunchKernelGGL(colsum_batched_finalize_kernel, dim3(fblocks), dim3(1024), 0, s, b, (const float*)workspace);
    UZ_LAUNCH_CHECK("uz_colsum_batched(finalize)");
  }
  return UZ_OK;
}
