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

// fold stage-0 partials of image b: coef[b] (LSregress scale, models.py:7-21), den partial per image.  One workgroup of kRThreads.
__device__ __forceinline__ void recon_fold0_image(const float* __restrict__ ws, float* __restrict__ coef, float* __restrict__ den_img, int nblk, int b,
                                                  double* lds /* [kRThreads * 3] */) {
  double v[3] = {0.0, 0.0, 0.0};
  // four triples requested together per round (600 partials per image at config 2: one round), added in index order as before
  for (int i = threadIdx.x; i < nblk; i += 4 * kRThreads) {
    float t[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool in = i + j * kRThreads < nblk;
      const float* q = ws + ((size_t)b * nblk + (in ? i + j * kRThreads : i)) * 3;
      t[j][0] = in ? q[0] : 0.0f; t[j][1] = in ? q[1] : 0.0f; t[j][2] = in ? q[2] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[0] += (double)t[j][0]; v[1] += (double)t[j][1]; v[2] += (double)t[j][2]; }
  }
  block_sum_double<3>(v, lds);
  if (threadIdx.x == 0) {
    coef[b] = fminf(fmaxf((float)v[0] / fmaxf((float)v[1], 1e-5f), 0.001f), 1000.0f);
    den_img[b] = (float)v[2];
  }
}
// (one block per image; the fused light objective on one rank runs the same fold as an extra workgroup per image of the render loss's first
// pass instead -- sgr_light_objective_fwd, FoldJob in sgr_launch.h -- and saves this launch)
static __global__ __launch_bounds__(kRThreads) void recon_fold0(const float* __restrict__ ws, float* __restrict__ coef,
                                                          float* __restrict__ den_img, int nblk) {
  __shared__ double lds[kRThreads * 3];
  recon_fold0_image(ws, coef, den_img, nblk, (int)blockIdx.x, lds);
}

// The tail of the light objective when the batch is not sharded (otherwise sgr_objective_finalize after the all-reduce):
//   reconstErr = num / max(den, 1e-5) / divisor (wrapperBRDFLight.py:179-188), objective = ren_w renderErr + rec_w reconstErr (trainLight.py:237)
struct ObjectiveTail { const float* render_err; float ren_w, rec_w, divisor_e; float* objective; float* recon_err; float* one; };

// One workgroup of 1024 threads (round 4; 256 before).  Thread t owns partials t, t + 1024, ...: eight of them requested together, each
// added to its own double accumulator, the eight combined in a fixed order, then the fixed LDS tree -- the result depends on n alone.
// The wave trace showed this kernel as 13 us of pure memory latency behind the objective's 320 us backward (ten dependent rounds of four
// loads on four waves); 9 600 partials are now two rounds on sixteen waves.
constexpr int kFold1Threads = 1024;
static __global__ __launch_bounds__(kFold1Threads) void recon_fold1(const float* __restrict__ ws, const float* __restrict__ den_img,
                                                              float* __restrict__ parts, int bn, int nblk, ObjectiveTail tail) {
  __shared__ double lds[kFold1Threads * 2];
  constexpr int kInFlight = 8;
  double w[kInFlight];
#pragma unroll
  for (int j = 0; j < kInFlight; ++j) w[j] = 0.0;
  const int n = bn * nblk;
  for (int i = threadIdx.x; i < n; i += kInFlight * kFold1Threads) {
    float a[kInFlight];
#pragma unroll
    for (int j = 0; j < kInFlight; ++j) a[j] = i + j * kFold1Threads < n ? ws[i + j * kFold1Threads] : 0.0f;
#pragma unroll
    for (int j = 0; j < kInFlight; ++j) w[j] += (double)a[j];
  }
  double v[2] = {((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7])), 0.0};
  for (int i = threadIdx.x; i < bn; i += kFold1Threads) v[1] += (double)den_img[i];
#pragma unroll
  for (int i = 0; i < 2; ++i) lds[threadIdx.x * 2 + i] = v[i];
  __syncthreads();
  for (int s = kFold1Threads / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      lds[threadIdx.x * 2] += lds[(threadIdx.x + s) * 2];
      lds[threadIdx.x * 2 + 1] += lds[(threadIdx.x + s) * 2 + 1];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float num = (float)lds[0], den = (float)lds[1];
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
