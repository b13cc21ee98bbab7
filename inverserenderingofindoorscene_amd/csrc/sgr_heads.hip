// Output activations of the three light decoders (models.py:336-346) on gfx950:
//   axis   = normalize3(1.01 tanh(x))            (mode 0: per lobe, / clamp(|.|, min 1e-6))
//   lamb   = clamp(0.5 (1.01 tanh(x) + 1), 0, 1) (mode 1)
//   weight = clamp(0.5 (1.01 tanh(x) + 1), 0, 1) (mode 2)
// and, on request, the packed [bn,7K,R,C] cascade hand-off tensor of wrapperBRDFLight.py:167-168
// (axis 3K | lamb K | weight 3K channels).  One pass each way: a thread owns one lobe of one pixel (7 values),
// lanes run along the pixels; ~14 separate elementwise kernels and their autograd graph in the reference.
#include "sgr_launch.h"
#include "sgr_math.h"

namespace sgr {

// tanh to ~1 ulp: odd polynomial below 0.625 (no cancellation), 1 - 2/(e^{2|x|} + 1) above
__device__ __forceinline__ float tanh_f(float x) {
  const float ax = fabsf(x);
  const float z = x * x;
  float p = -5.70498872745e-3f;
  p = fmaf(p, z, 2.06390887954e-2f);
  p = fmaf(p, z, -5.37397155531e-2f);
  p = fmaf(p, z, 1.33314422036e-1f);
  p = fmaf(p, z, -3.33332819422e-1f);
  const float small = fmaf(p * z, x, x);
  const float e = fexp2(ax * 2.8853900817779268f);        // e^{2|x|}
  const float big = copysignf(1.0f - 2.0f / (e + 1.0f), x);
  return ax < 0.625f ? small : big;
}

// 0.5 * (1.01 t + 1) with torch's op-by-op rounding (the clamp kinks sit on these bits)
__device__ __forceinline__ float unit_pre(float t) { return fmul_rn(0.5f, fadd_rn(fmul_rn(1.01f, t), 1.0f)); }

__global__ __launch_bounds__(256) void heads_fwd_kernel(const float* __restrict__ xa, const float* __restrict__ xl,
                                                         const float* __restrict__ xw, float* __restrict__ axis,
                                                         float* __restrict__ lamb, float* __restrict__ weight,
                                                         float* __restrict__ packed, int K, int RC) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int bk = blockIdx.y;                      // b*K + k
  if (p >= RC) return;
  const int b = bk / K, k = bk - b * K;
  const size_t o3 = (size_t)bk * 3 * RC + p, o1 = (size_t)bk * RC + p;
  const float a0 = fmul_rn(1.01f, tanh_f(xa[o3])), a1 = fmul_rn(1.01f, tanh_f(xa[o3 + RC])), a2 = fmul_rn(1.01f, tanh_f(xa[o3 + 2 * (size_t)RC]));
  const float n = fmaxf(sqrtf(fadd_rn(fadd_rn(fmul_rn(a0, a0), fmul_rn(a1, a1)), fmul_rn(a2, a2))), 1e-6f);
  const float y0 = a0 / n, y1 = a1 / n, y2 = a2 / n;
  const float l = fminf(fmaxf(unit_pre(tanh_f(xl[o1])), 0.0f), 1.0f);
  const float w0 = fminf(fmaxf(unit_pre(tanh_f(xw[o3])), 0.0f), 1.0f);
  const float w1 = fminf(fmaxf(unit_pre(tanh_f(xw[o3 + RC])), 0.0f), 1.0f);
  const float w2 = fminf(fmaxf(unit_pre(tanh_f(xw[o3 + 2 * (size_t)RC])), 0.0f), 1.0f);
  axis[o3] = y0; axis[o3 + RC] = y1; axis[o3 + 2 * (size_t)RC] = y2;
  lamb[o1] = l;
  weight[o3] = w0; weight[o3 + RC] = w1; weight[o3 + 2 * (size_t)RC] = w2;
  if (packed) {
    float* pb = packed + (size_t)b * 7 * K * RC + p;
    pb[(size_t)(3 * k) * RC] = y0; pb[(size_t)(3 * k + 1) * RC] = y1; pb[(size_t)(3 * k + 2) * RC] = y2;
    pb[(size_t)(3 * K + k) * RC] = l;
    pb[(size_t)(4 * K + 3 * k) * RC] = w0; pb[(size_t)(4 * K + 3 * k + 1) * RC] = w1; pb[(size_t)(4 * K + 3 * k + 2) * RC] = w2;
  }
}

// cotangents may arrive through the separate outputs, the packed tensor, or both (any may be NULL)
__global__ __launch_bounds__(256) void heads_bwd_kernel(const float* __restrict__ xa, const float* __restrict__ xl,
                                                         const float* __restrict__ xw, const float* __restrict__ g_axis,
                                                         const float* __restrict__ g_lamb, const float* __restrict__ g_weight,
                                                         const float* __restrict__ g_packed, float* __restrict__ gxa,
                                                         float* __restrict__ gxl, float* __restrict__ gxw, int K, int RC) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int bk = blockIdx.y;
  if (p >= RC) return;
  const int b = bk / K, k = bk - b * K;
  const size_t o3 = (size_t)bk * 3 * RC + p, o1 = (size_t)bk * RC + p;
  float ga[3] = {0.f, 0.f, 0.f}, gl = 0.f, gw[3] = {0.f, 0.f, 0.f};
  if (g_axis) { ga[0] = g_axis[o3]; ga[1] = g_axis[o3 + RC]; ga[2] = g_axis[o3 + 2 * (size_t)RC]; }
  if (g_lamb) gl = g_lamb[o1];
  if (g_weight) { gw[0] = g_weight[o3]; gw[1] = g_weight[o3 + RC]; gw[2] = g_weight[o3 + 2 * (size_t)RC]; }
  if (g_packed) {
    const float* pb = g_packed + (size_t)b * 7 * K * RC + p;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      ga[c] += pb[(size_t)(3 * k + c) * RC];
      gw[c] += pb[(size_t)(4 * K + 3 * k + c) * RC];
    }
    gl += pb[(size_t)(3 * K + k) * RC];
  }
  // axis: y = a / max(|a|, 1e-6), a = 1.01 tanh(x)
  float t[3], a[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { t[c] = tanh_f(xa[o3 + (size_t)c * RC]); a[c] = fmul_rn(1.01f, t[c]); }
  const float nr = sqrtf(fadd_rn(fadd_rn(fmul_rn(a[0], a[0]), fmul_rn(a[1], a[1])), fmul_rn(a[2], a[2])));
  const float n = fmaxf(nr, 1e-6f);
  const float inv = 1.0f / n;
  float dot = 0.f;
  if (nr >= 1e-6f) dot = (a[0] * ga[0] + a[1] * ga[1] + a[2] * ga[2]) * inv * inv;   // (y . g) / n ; the min-clamp blocks it below 1e-6
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float da = (ga[c] - a[c] * dot) * inv;
    gxa[o3 + (size_t)c * RC] = da * 1.01f * (1.0f - t[c] * t[c]);
  }
  // lamb / weight: clamp passes the cotangent on 0 <= pre <= 1 (inclusive, like torch)
  {
    const float tt = tanh_f(xl[o1]), pre = unit_pre(tt);
    gxl[o1] = (pre >= 0.0f && pre <= 1.0f) ? gl * 0.505f * (1.0f - tt * tt) : 0.0f;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float tt = tanh_f(xw[o3 + (size_t)c * RC]), pre = unit_pre(tt);
    gxw[o3 + (size_t)c * RC] = (pre >= 0.0f && pre <= 1.0f) ? gw[c] * 0.505f * (1.0f - tt * tt) : 0.0f;
  }
}

}  // namespace sgr

using namespace sgr;

extern "C" int sgr_light_heads_fwd(const float* x_axis, const float* x_lamb, const float* x_weight, float* axis, float* lamb,
                                   float* weight, float* packed, int bn, int K, int R, int C, void* stream) {
  SGR_REQUIRE(x_axis && x_lamb && x_weight && axis && lamb && weight, "sgr_light_heads_fwd: NULL tensor");
  SGR_REQUIRE(bn > 0 && K > 0 && R > 0 && C > 0, "sgr_light_heads_fwd: non-positive size");
  SGR_SUPPORTED((long long)bn * K <= 65535, "sgr_light_heads_fwd: bn * SGNum > 65535");
  const int RC = R * C;
  hipLaunchKernelGGL(heads_fwd_kernel, dim3((RC + 255) / 256, bn * K), dim3(256), 0, (hipStream_t)stream, x_axis, x_lamb, x_weight, axis,
                     lamb, weight, packed, K, RC);
  return sgr_check((int)hipGetLastError(), "sgr_light_heads_fwd");
}

extern "C" int sgr_light_heads_bwd(const float* x_axis, const float* x_lamb, const float* x_weight, const float* g_axis,
                                   const float* g_lamb, const float* g_weight, const float* g_packed, float* gx_axis, float* gx_lamb,
                                   float* gx_weight, int bn, int K, int R, int C, void* stream) {
  SGR_REQUIRE(x_axis && x_lamb && x_weight && gx_axis && gx_lamb && gx_weight, "sgr_light_heads_bwd: NULL tensor");
  SGR_REQUIRE(bn > 0 && K > 0 && R > 0 && C > 0, "sgr_light_heads_bwd: non-positive size");
  SGR_SUPPORTED((long long)bn * K <= 65535, "sgr_light_heads_bwd: bn * SGNum > 65535");
  const int RC = R * C;
  hipLaunchKernelGGL(heads_bwd_kernel, dim3((RC + 255) / 256, bn * K), dim3(256), 0, (hipStream_t)stream, x_axis, x_lamb, x_weight,
                     g_axis, g_lamb, g_weight, g_packed, gx_axis, gx_lamb, gx_weight, K, RC);
  return sgr_check((int)hipGetLastError(), "sgr_light_heads_bwd");
}
