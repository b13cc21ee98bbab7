// Raw throughput of the env-image write pattern (development tool, round 2).
//   hipcc -O2 --offload-arch=gfx950 tools/storebench.hip -o tools/storebench
// The forward kernels write env[b, c, pixel, 128 directions] (512 B per pixel and colour) one table row (64 B) at a time:
// a wave owns 64 consecutive pixels, a store instruction covers 16 pixels x 64 B.  How fast can the chip absorb 472 MB
// written like that, against 128 / 256 / 512-byte segments, with and without the non-temporal hint, with 2 waves per SIMD?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// SEG = bytes written per (pixel, colour) per step (64 .. 512); NT = non-temporal
template <int SEG, bool NT>
__global__ __launch_bounds__(64, 2) void store_kernel(float* __restrict__ env, int RC, int J, int spin) {
  const int tiles = (RC + 63) / 64;
  const int b = blockIdx.x / tiles, p0 = (blockIdx.x - b * tiles) * 64;
  const int lane = threadIdx.x;
  constexpr int LPR = SEG / 16;            // lanes per pixel row
  constexpr int RPI = 64 / LPR;            // pixel rows per instruction
  const int lrow = lane / LPR, col = (lane % LPR) * 4;
  float* img = env + (size_t)b * 3 * RC * J;
  f32x4 v = {1.0f * lane, 2.0f, 3.0f, 4.0f};
  for (int j0 = 0; j0 < J; j0 += SEG / 4) {
    // stand-in for the row's arithmetic
    for (int s = 0; s < spin; ++s) { asm volatile("v_fmac_f32 %0, %0, %0" : "+v"(v.x)); }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float* cbase = img + ((size_t)c * RC + p0) * J + j0;
#pragma unroll
      for (int it = 0; it < 64 / RPI; ++it) {
        f32x4* dst = reinterpret_cast<f32x4*>(cbase + (size_t)(it * RPI + lrow) * J + col);
        if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
      }
    }
  }
}
// one pixel per lane, 16-byte pieces straight from "registers" (no transpose)
template <bool NT>
__global__ __launch_bounds__(64, 2) void store_direct(float* __restrict__ env, int RC, int J, int spin) {
  const int tiles = (RC + 63) / 64;
  const int b = blockIdx.x / tiles, p = (blockIdx.x - b * tiles) * 64 + threadIdx.x;
  float* img = env + (size_t)b * 3 * RC * J;
  f32x4 v = {1.0f * threadIdx.x, 2.0f, 3.0f, 4.0f};
  for (int j0 = 0; j0 < J; j0 += 16) {
    for (int s = 0; s < spin; ++s) { asm volatile("v_fmac_f32 %0, %0, %0" : "+v"(v.x)); }
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4* dst = reinterpret_cast<f32x4*>(img + ((size_t)c * RC + p) * J + j0 + q * 4);
        if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
      }
  }
}

// read side: the backward streams the env cotangent the same way (one 64-byte row segment per pixel and colour at a time)
template <int SEG, bool NT>
__global__ __launch_bounds__(64, 2) void load_kernel(const float* __restrict__ env, float* __restrict__ out, int RC, int J, int spin) {
  const int tiles = (RC + 63) / 64;
  const int b = blockIdx.x / tiles, p0 = (blockIdx.x - b * tiles) * 64;
  const int lane = threadIdx.x;
  constexpr int LPR = SEG / 16, RPI = 64 / LPR;
  const int lrow = lane / LPR, col = (lane % LPR) * 4;
  const float* img = env + (size_t)b * 3 * RC * J;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int j0 = 0; j0 < J; j0 += SEG / 4) {
    for (int s = 0; s < spin; ++s) { asm volatile("v_fmac_f32 %0, %0, %0" : "+v"(acc.x)); }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* cbase = img + ((size_t)c * RC + p0) * J + j0;
#pragma unroll
      for (int it = 0; it < 64 / RPI; ++it) {
        const f32x4* src = reinterpret_cast<const f32x4*>(cbase + (size_t)(it * RPI + lrow) * J + col);
        acc += NT ? __builtin_nontemporal_load(src) : *src;
      }
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

int main(int argc, char** argv) {
  const int bn = 16, R = 120, C = 160, RC = R * C, J = 128;
  const size_t n = (size_t)bn * 3 * RC * J;
  float *env, *other; CHECK(hipMalloc(&env, n * 4)); CHECK(hipMalloc(&other, n * 4));
  const dim3 grid(bn * ((RC + 63) / 64)), block(64);
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto bench = [&](const char* name, auto launch) {
    for (int spin : {0, 600, 1200}) {
      launch(spin); CHECK(hipDeviceSynchronize());
      float tot = 0;
      const int reps = 10;
      for (int i = 0; i < reps; ++i) {
        CHECK(hipMemsetAsync(other, i, n * 4, 0));      // cold caches, like the bench loop's 1.3 GB working set
        CHECK(hipEventRecord(e0, 0)); launch(spin); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); tot += ms;
      }
      printf("%-34s spin %5d  %7.1f us  %7.1f GB/s\n", name, spin, tot / reps * 1e3, n * 4 / (tot / reps * 1e-3) * 1e-9);
    }
  };
  bench("64-B segments, nt", [&](int s) { hipLaunchKernelGGL((store_kernel<64, true>), grid, block, 0, 0, env, RC, J, s); });
  bench("64-B segments, plain", [&](int s) { hipLaunchKernelGGL((store_kernel<64, false>), grid, block, 0, 0, env, RC, J, s); });
  bench("128-B segments, nt", [&](int s) { hipLaunchKernelGGL((store_kernel<128, true>), grid, block, 0, 0, env, RC, J, s); });
  bench("128-B segments, plain", [&](int s) { hipLaunchKernelGGL((store_kernel<128, false>), grid, block, 0, 0, env, RC, J, s); });
  bench("256-B segments, nt", [&](int s) { hipLaunchKernelGGL((store_kernel<256, true>), grid, block, 0, 0, env, RC, J, s); });
  bench("512-B segments, nt", [&](int s) { hipLaunchKernelGGL((store_kernel<512, true>), grid, block, 0, 0, env, RC, J, s); });
  bench("512-B segments, plain", [&](int s) { hipLaunchKernelGGL((store_kernel<512, false>), grid, block, 0, 0, env, RC, J, s); });
  bench("16-B pieces per lane, nt", [&](int s) { hipLaunchKernelGGL((store_direct<true>), grid, block, 0, 0, env, RC, J, s); });
  bench("16-B pieces per lane, plain", [&](int s) { hipLaunchKernelGGL((store_direct<false>), grid, block, 0, 0, env, RC, J, s); });
  float* outp; CHECK(hipMalloc(&outp, 4));
  bench("LOAD 64-B segments, plain", [&](int s) { hipLaunchKernelGGL((load_kernel<64, false>), grid, block, 0, 0, env, outp, RC, J, s); });
  bench("LOAD 64-B segments, nt", [&](int s) { hipLaunchKernelGGL((load_kernel<64, true>), grid, block, 0, 0, env, outp, RC, J, s); });
  bench("LOAD 128-B segments, plain", [&](int s) { hipLaunchKernelGGL((load_kernel<128, false>), grid, block, 0, 0, env, outp, RC, J, s); });
  bench("LOAD 128-B segments, nt", [&](int s) { hipLaunchKernelGGL((load_kernel<128, true>), grid, block, 0, 0, env, outp, RC, J, s); });
  bench("LOAD 512-B segments, plain", [&](int s) { hipLaunchKernelGGL((load_kernel<512, false>), grid, block, 0, 0, env, outp, RC, J, s); });
  {
    float tot = 0; const int reps = 10;
    for (int i = 0; i < reps; ++i) { CHECK(hipEventRecord(e0, 0)); CHECK(hipMemsetAsync(env, i, n * 4, 0)); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); tot += ms; }
    printf("%-34s             %7.1f us  %7.1f GB/s\n", "hipMemsetAsync of the same bytes", tot / reps * 1e3, n * 4 / (tot / reps * 1e-3) * 1e-9);
  }
  return 0;
}
