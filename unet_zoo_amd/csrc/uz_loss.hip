// Step tail on the device: BCEWithLogitsLoss (mean; scripts/train.py:135, training_loop.py:113-119)
// and the Dice coefficient of the thresholded prediction (utils/metrics.py:7-24) in one pass over the logits,
// with the gradient of the loss written on the way: no host synchronisation, fixed summation order.
#include "uz_common.h"

namespace {

constexpr int BD_THREADS = 256;
constexpr int BD_MAX_ROWS = 1024;

// part[row][4] = sum bce, sum pred*t, sum pred, sum t  (pred = sigmoid(x) > 0.5  <=>  x > 0)
__global__ __launch_bounds__(BD_THREADS) void bce_dice_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                              long long n, float inv_n, float* __restrict__ dlogits,
                                                              double* __restrict__ part) {
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  for (long long i = (long long)blockIdx.x * BD_THREADS + threadIdx.x; i < n; i += (long long)gridDim.x * BD_THREADS) {
    const float xv = x[i], tv = t[i];
    // ATen's binary_cross_entropy_with_logits: (1 - t) x + max(-x, 0) + log(exp(-max(-x, 0)) + exp(-x - max(-x, 0)))
    const float m = fmaxf(-xv, 0.f);
    s[0] += (double)((1.f - tv) * xv + m + logf(expf(-m) + expf(-xv - m)));
    const float pred = xv > 0.f ? 1.f : 0.f;
    s[1] += (double)(pred * tv);
    s[2] += (double)pred;
    s[3] += (double)tv;
    if (dlogits != nullptr) dlogits[i] = (1.f / (1.f + expf(-xv)) - tv) * inv_n;
  }
  __shared__ double red[BD_THREADS / 64][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    double v = s[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    double v = 0.0;
    for (int w = 0; w < BD_THREADS / 64; ++w) v += red[w][threadIdx.x];
    part[(size_t)blockIdx.x * 4 + threadIdx.x] = v;
  }
}

// one workgroup: thread k adds rows k, k + 256, ... then a fixed tree over the threads
__global__ __launch_bounds__(BD_THREADS) void bce_dice_finalize_kernel(const double* __restrict__ part, int rows,
                                                                       long long n, float* __restrict__ out) {
  __shared__ double red[BD_THREADS][4];
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  for (int r = threadIdx.x; r < rows; r += BD_THREADS)
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] += part[(size_t)r * 4 + k];
#pragma unroll
  for (int k = 0; k < 4; ++k) red[threadIdx.x][k] = s[k];
  __syncthreads();
  for (int o = BD_THREADS / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
#pragma unroll
      for (int k = 0; k < 4; ++k) red[threadIdx.x][k] += red[threadIdx.x + o][k];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = (float)(red[0][0] / (double)n);
    const double uni = red[0][2] + red[0][3];
    out[1] = uni == 0.0 ? 1.f : (float)((2.0 * red[0][1] + 1e-7) / (uni + 1e-7));
  }
}

int bd_rows(long long n) {
  long long r = (n + (long long)BD_THREADS * 8 - 1) / ((long long)BD_THREADS * 8);
  if (r > BD_MAX_ROWS) r = BD_MAX_ROWS;
  if (r < 1) r = 1;
  return (int)r;
}

}  // namespace

extern "C" long long uz_bce_dice_workspace_bytes(long long n) { return n > 0 ? (long long)bd_rows(n) * 4 * sizeof(double) : -1; }

extern "C" int uz_bce_dice(const float* logits, const float* target, long long n, float* dlogits, float* out2,
                           void* workspace, void* stream) {
  UZ_REQUIRE(logits && target && out2 && workspace && n > 0, "uz_bce_dice: bad arguments");
  const int rows = bd_rows(n);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(bce_dice_kernel, dim3(rows), dim3(BD_THREADS), 0, s, logits, target, n, 1.f / (float)n, dlogits,
                     (double*)workspace);
  UZ_LAUNCH_CHECK("uz_bce_dice");
  hipLaunchKernelGGL(bce_dice_finalize_kernel, dim3(1), dim3(BD_THREADS), 0, s, (const double*)workspace, rows, n, out2);
  UZ_LAUNCH_CHECK("uz_bce_dice(finalize)");
  return UZ_OK;
}
