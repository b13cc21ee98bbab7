// Device-side glue either side of the render path, gfx950.
//
//   sgr_light_albedo_scale ..... the global light / albedo scale of testReal.py:421-432 (two ratios of sums + a clip),
//                                 which the reference evaluates on the host through four `.item()` synchronisations
//   sgr_light_input_fwd ........ the light encoder's input of wrapperBRDFLight.py:138-156: per-image mean-normalisation
//                                 of albedo and depth, bilinear resize (F.interpolate, align_corners=False) of the five
//                                 maps to 480x640 and their concatenation [im, albedo, (normal+1)/2, (rough+1)/2, depth]
//
// Streaming, HBM-bound, launch-latency-sized work: block-partial reductions folded in a fixed order (no atomics,
// bit-reproducible), everything stays on the caller's stream.
#include "sgr_launch.h"

namespace sgr {

constexpr int kGlueThreads = 256;
constexpr int kGlueSplit = 64;

template <int N>
__device__ __forceinline__ void glue_block_sum(float (&v)[N], float* lds /* [4*N] */) {
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[i] += __shfl_down(v[i], off, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0)
#pragma unroll
    for (int i = 0; i < N; ++i) lds[wave * N + i] = v[i];
  __syncthreads();
  if (threadIdx.x == 0)
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = (lds[i] + lds[N + i]) + (lds[2 * N + i] + lds[3 * N + i]);
  __syncthreads();
}

// ---- testReal.py:421-432 --------------------------------------------------------------------------------------------
// stage 1: partial sums of (diffuseScaled, diffuse, specScaled, spec) and the partial max of albedo
__global__ __launch_bounds__(kGlueThreads) void scale_stage1(const float* __restrict__ dn, const float* __restrict__ d,
                                                              const float* __restrict__ sn, const float* __restrict__ s,
                                                              const float* __restrict__ albedo, float* __restrict__ ws /* [kGlueSplit,5] */,
                                                              long long n, long long n_alb) {
  __shared__ float lds[4 * 4];
  __shared__ float mx[4];
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (long long i = (long long)blockIdx.x * kGlueThreads + threadIdx.x; i < n; i += (long long)kGlueSplit * kGlueThreads) {
    acc[0] += dn[i]; acc[1] += d[i]; acc[2] += sn[i]; acc[3] += s[i];
  }
  float m = -INFINITY;
  for (long long i = (long long)blockIdx.x * kGlueThreads + threadIdx.x; i < n_alb; i += (long long)kGlueSplit * kGlueThreads) m = fmaxf(m, albedo[i]);
  glue_block_sum<4>(acc, lds);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
  if ((threadIdx.x & 63) == 0) mx[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float* w = ws + (size_t)blockIdx.x * 5;
    w[0] = acc[0]; w[1] = acc[1]; w[2] = acc[2]; w[3] = acc[3];
    w[4] = fmaxf(fmaxf(mx[0], mx[1]), fmaxf(mx[2], mx[3]));
  }
}
// stage 2: cDiff = sum(dNew)/sum(d), cSpec = sum(sNew)/sum(s); the branch and the clip of testReal.py:422-430
__global__ void scale_stage2(const float* __restrict__ ws, float* __restrict__ out /* (cLight, cAlbedo, cDiff, cSpec) */) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double t[4] = {0, 0, 0, 0};
  float amax = -INFINITY;
  for (int b = 0; b < kGlueSplit; ++b) {
    for (int i = 0; i < 4; ++i) t[i] += (double)ws[b * 5 + i];
    amax = fmaxf(amax, ws[b * 5 + 4]);
  }
  const float cDiff = (float)t[0] / (float)t[1], cSpec = (float)t[2] / (float)t[3];
  const double inv_amax = 1.0 / (double)amax;                  // the reference does this part in Python floats (double)
  double cAlbedo, cLight;
  if ((double)cSpec < 1e-3) {
    cAlbedo = inv_amax;
    cLight = (double)cDiff / cAlbedo;
  } else {
    cLight = (double)cSpec;
    cAlbedo = (double)cDiff / cLight;
    cAlbedo = fmin(fmax(cAlbedo, 1e-3), inv_amax);             // np.clip(a, lo, hi) == minimum(maximum(a, lo), hi)
    cLight = (double)cDiff / cAlbedo;
  }
  out[0] = (float)cLight; out[1] = (float)cAlbedo; out[2] = cDiff; out[3] = cSpec;
}

// ---- wrapperBRDFLight.py:138-156 --------------------------------------------------------------------------------------
// per-image sums of albedo [3*h*w] and depth [h*w]: grid (kGlueSplit, bn)
__global__ __launch_bounds__(kGlueThreads) void mean_stage(const float* __restrict__ albedo, const float* __restrict__ depth,
                                                            float* __restrict__ ws /* [bn,kGlueSplit,2] */, int hw) {
  __shared__ float lds[4 * 2];
  const int b = blockIdx.y;
  float acc[2] = {0.f, 0.f};
  const float* a = albedo + (size_t)b * 3 * hw;
  const float* dp = depth + (size_t)b * hw;
  for (int i = blockIdx.x * kGlueThreads + threadIdx.x; i < 3 * hw; i += kGlueSplit * kGlueThreads) acc[0] += a[i];
  for (int i = blockIdx.x * kGlueThreads + threadIdx.x; i < hw; i += kGlueSplit * kGlueThreads) acc[1] += dp[i];
  glue_block_sum<2>(acc, lds);
  if (threadIdx.x == 0) {
    ws[((size_t)b * kGlueSplit + blockIdx.x) * 2 + 0] = acc[0];
    ws[((size_t)b * kGlueSplit + blockIdx.x) * 2 + 1] = acc[1];
  }
}
// torch's upsample_bilinear2d source index (align_corners = False): max(scale * (dst + 0.5) - 0.5, 0)
__device__ __forceinline__ void src_index(int dst, float scale, int in_size, int& i0, int& i1, float& l0, float& l1) {
  const float r = fmaxf(scale * ((float)dst + 0.5f) - 0.5f, 0.0f);
  i0 = min((int)r, in_size - 1);
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = r - (float)i0;
  l0 = 1.0f - l1;
}
// one thread per output pixel: 11 channels, the per-image albedo / depth scales applied per tap (the reference normalises
// first, then interpolates); also writes the normalised albedo / depth maps the wrapper returns (threads with oy<h, ox<w)
__global__ __launch_bounds__(kGlueThreads) void light_input_kernel(const float* __restrict__ im, const float* __restrict__ albedo,
                                                                    const float* __restrict__ normal, const float* __restrict__ rough,
                                                                    const float* __restrict__ depth, const float* __restrict__ ws,
                                                                    float* __restrict__ out, float* __restrict__ albedo_n,
                                                                    float* __restrict__ depth_n, int h, int w, int H, int W) {
  const int b = blockIdx.y;
  const int o = blockIdx.x * kGlueThreads + threadIdx.x;
  const int hw = h * w;
  // per-image means, folded in a fixed order (every thread does the same 2 x kGlueSplit adds: cheaper than a third launch)
  double sa = 0.0, sd = 0.0;
  for (int s = 0; s < kGlueSplit; ++s) { sa += (double)ws[((size_t)b * kGlueSplit + s) * 2]; sd += (double)ws[((size_t)b * kGlueSplit + s) * 2 + 1]; }
  const float ma = fmaxf((float)(sa / (double)(3 * hw)), 1e-10f), md = fmaxf((float)(sd / (double)hw), 1e-10f);
  if (o < hw) {
#pragma unroll
    for (int c = 0; c < 3; ++c) albedo_n[((size_t)b * 3 + c) * hw + o] = albedo[((size_t)b * 3 + c) * hw + o] / ma / 3.0f;
    depth_n[(size_t)b * hw + o] = depth[(size_t)b * hw + o] / md / 3.0f;
  }
  if (o >= H * W) return;
  const int oy = o / W, ox = o - oy * W;
  int y0, y1, x0, x1;
  float ly0, ly1, lx0, lx1;
  src_index(oy, (float)h / (float)H, h, y0, y1, ly0, ly1);
  src_index(ox, (float)w / (float)W, w, x0, x1, lx0, lx1);
  const int i00 = y0 * w + x0, i01 = y0 * w + x1, i10 = y1 * w + x0, i11 = y1 * w + x1;
  auto tap = [&](const float* p) { return ly0 * (lx0 * p[i00] + lx1 * p[i01]) + ly1 * (lx0 * p[i10] + lx1 * p[i11]); };
  auto tapn = [&](const float* p, float m) {
    return ly0 * (lx0 * (p[i00] / m / 3.0f) + lx1 * (p[i01] / m / 3.0f)) + ly1 * (lx0 * (p[i10] / m / 3.0f) + lx1 * (p[i11] / m / 3.0f));
  };
  float* ob = out + (size_t)b * 11 * H * W + o;
  const size_t HW = (size_t)H * W;
#pragma unroll
  for (int c = 0; c < 3; ++c) ob[(size_t)c * HW] = tap(im + ((size_t)b * 3 + c) * hw);
#pragma unroll
  for (int c = 0; c < 3; ++c) ob[(size_t)(3 + c) * HW] = tapn(albedo + ((size_t)b * 3 + c) * hw, ma);
#pragma unroll
  for (int c = 0; c < 3; ++c) ob[(size_t)(6 + c) * HW] = 0.5f * (tap(normal + ((size_t)b * 3 + c) * hw) + 1.0f);
  ob[(size_t)9 * HW] = 0.5f * (tap(rough + (size_t)b * hw) + 1.0f);
  ob[(size_t)10 * HW] = tapn(depth + (size_t)b * hw, md);
}

}  // namespace sgr

using namespace sgr;

extern "C" int sgr_glue_workspace_floats(int bn) { return kGlueSplit * 5 + bn * kGlueSplit * 2; }

extern "C" int sgr_light_albedo_scale(const float* diffuse_scaled, const float* diffuse, const float* spec_scaled, const float* spec,
                                      const float* albedo, float* out4, float* workspace, long long n, long long n_albedo, void* stream) {
  SGR_REQUIRE(diffuse_scaled && diffuse && spec_scaled && spec && albedo && out4 && workspace, "sgr_light_albedo_scale: NULL tensor");
  SGR_REQUIRE(n > 0 && n_albedo > 0, "sgr_light_albedo_scale: non-positive size");
  const hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(scale_stage1, dim3(kGlueSplit), dim3(kGlueThreads), 0, st, diffuse_scaled, diffuse, spec_scaled, spec, albedo,
                     workspace, n, n_albedo);
  hipLaunchKernelGGL(scale_stage2, dim3(1), dim3(64), 0, st, workspace, out4);
  return sgr_check((int)hipGetLastError(), "sgr_light_albedo_scale");
}

extern "C" int sgr_light_input_fwd(const float* im, const float* albedo, const float* normal, const float* rough, const float* depth,
                                   float* out, float* albedo_norm, float* depth_norm, float* workspace, int bn, int h, int w, int H,
                                   int W, void* stream) {
  SGR_REQUIRE(im && albedo && normal && rough && depth && out && albedo_norm && depth_norm && workspace, "sgr_light_input_fwd: NULL tensor");
  SGR_REQUIRE(bn > 0 && h > 0 && w > 0 && H > 0 && W > 0, "sgr_light_input_fwd: non-positive size");
  const hipStream_t st = (hipStream_t)stream;
  float* ws = workspace + kGlueSplit * 5;
  hipLaunchKernelGGL(mean_stage, dim3(kGlueSplit, bn), dim3(kGlueThreads), 0, st, albedo, depth, ws, h * w);
  const int n = (H * W > h * w ? H * W : h * w);
  hipLaunchKernelGGL(light_input_kernel, dim3((n + kGlueThreads - 1) / kGlueThreads, bn), dim3(kGlueThreads), 0, st, im, albedo, normal,
                     rough, depth, ws, out, albedo_norm, depth_norm, h, w, H, W);
  return sgr_check((int)hipGetLastError(), "sgr_light_input_fwd");
}
