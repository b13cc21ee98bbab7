// Deterministic folds of the env-reconstruction partial sums (shared by sgr_recon.hip and the fused
// objective, sgr_fused_recon.hip): double accumulation in a fixed order, no atomics, no host sync.
#pragma once
#include "sgr_launch.h"

namespace sgr {

constexpr int kRThreads = 256;

// deterministic block sum in double: thread t adds elements t, t+256, ...; fixed LDS tree afterwards
template <int N>
__device__ __forceinline__ void block_sum_double(double (&v)[N], double* lds /* [256*N] */) {
#pragma unroll
  for (int i = 0; i < N; ++i) lds[threadIdx.x * N + i] = v[i];
  __syncthreads();
  for (int s = kRThreads / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
#pragma unroll
      for (int i = 0; i < N; ++i) lds[threadIdx.x * N + i] += lds[(threadIdx.x + s) * N + i];
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = lds[i];
}

// fold stage-0 partials: coef[b], den partial per image   (one block per image)
static __global__ __launch_bounds__(kRThreads) void recon_fold0(const float* __restrict__ ws, float* __restrict__ coef,
                                                          float* __restrict__ den_img, int nblk) {
  __shared__ double lds[kRThreads * 3];
  const int b = blockIdx.x;
  double v[3] = {0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < nblk; i += kRThreads) {
    v[0] += (double)ws[((size_t)b * nblk + i) * 3 + 0];
    v[1] += (double)ws[((size_t)b * nblk + i) * 3 + 1];
    v[2] += (double)ws[((size_t)b * nblk + i) * 3 + 2];
  }
  block_sum_double<3>(v, lds);
  if (threadIdx.x == 0) {
    coef[b] = fminf(fmaxf((float)v[0] / fmaxf((float)v[1], 1e-5f), 0.001f), 1000.0f);
    den_img[b] = (float)v[2];
  }
}

// The tail of the light objective when the batch is not sharded (otherwise sgr_objective_finalize after the all-reduce):
//   reconstErr = num / max(den, 1e-5) / divisor (wrapperBRDFLight.py:179-188), objective = ren_w renderErr + rec_w reconstErr (trainLight.py:237)
struct ObjectiveTail { const float* render_err; float ren_w, rec_w, divisor_e; float* objective; float* recon_err; float* one; };

static __global__ __launch_bounds__(kRThreads) void recon_fold1(const float* __restrict__ ws, const float* __restrict__ den_img,
                                                          float* __restrict__ parts, int bn, int nblk, ObjectiveTail tail) {
  __shared__ double lds[kRThreads * 2];
  // four independent partial sums per thread (loads of one round in flight together: a single dependent chain over 9 600 partials
  // was 13.8 us of pure latency in the training loop), combined in a fixed order
  double v[2] = {0.0, 0.0}, w1 = 0.0, w2 = 0.0, w3 = 0.0;
  const int n = bn * nblk;
  for (int i = threadIdx.x; i < n; i += 4 * kRThreads) {
    const float a0 = ws[i];
    const float a1 = i + kRThreads < n ? ws[i + kRThreads] : 0.0f;
    const float a2 = i + 2 * kRThreads < n ? ws[i + 2 * kRThreads] : 0.0f;
    const float a3 = i + 3 * kRThreads < n ? ws[i + 3 * kRThreads] : 0.0f;
    v[0] += (double)a0; w1 += (double)a1; w2 += (double)a2; w3 += (double)a3;
  }
  v[0] = (v[0] + w1) + (w2 + w3);
  for (int i = threadIdx.x; i < bn; i += kRThreads) v[1] += (double)den_img[i];
  block_sum_double<2>(v, lds);
  if (threadIdx.x == 0) {
    const float num = (float)v[0], den = (float)v[1];
    parts[0] = num;
    parts[1] = den;
    if (tail.objective) {
      const float rec = num / fmaxf(den, 1e-5f) / tail.divisor_e;
      tail.recon_err[0] = rec;
      tail.objective[0] = tail.ren_w * tail.render_err[0] + tail.rec_w * rec;
      if (tail.one) tail.one[0] = 1.0f;      // the factor the gradients are scaled by so far (sgr_rescale_inplace_flip's slot 0)
    }
  }
}


}  // namespace sgr
