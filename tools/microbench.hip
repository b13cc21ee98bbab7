// VALU / transcendental issue-rate microbenchmark for gfx950 (development tool, not product).
// Answers the questions the kernel design hinges on: how many lanes/clk/CU do
// v_fma_f32, v_pk_fma_f32, v_exp_f32, v_rcp_f32, v_rsq_f32 sustain, alone and mixed 6:1 like
// the SG inner loop?   Build: hipcc --offload-arch=gfx950 -O3 microbench.hip -o microbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITERS = 2048;
constexpr int NACC = 16;   // independent chains per lane

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
  float a[NACC];
  f32x2 p[NACC / 2];
#pragma unroll
  for (int i = 0; i < NACC; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
#pragma unroll
  for (int i = 0; i < NACC / 2; ++i) p[i] = f32x2{a[2 * i], a[2 * i + 1]};
  const float m = 0.999f, c = 1e-3f;
  const f32x2 m2 = {m, m}, c2 = {c, c};
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      if (MODE == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      if (MODE == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      if (MODE == 3) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
      if (MODE == 5) {   // SG inner-loop mix: 6 fma : 1 exp
        asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_exp_f32 %0, %0\n"
                     "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2"
                     : "+v"(a[i]) : "v"(m), "v"(c));
      }
      if (MODE == 6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (MODE == 7) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
    }
    if (MODE == 4) {
#pragma unroll
      for (int i = 0; i < NACC / 2; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
    }
    if (MODE == 8) {   // 3 pk_fma : 1 exp (packed variant of the SG loop)
#pragma unroll
      for (int i = 0; i < NACC / 2; ++i)
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
#pragma unroll
      for (int i = 0; i < NACC / 2; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += a[i];
#pragma unroll
  for (int i = 0; i < NACC / 2; ++i) s += p[i].x + p[i].y;
  if (s == 12345.678f) out[0] = s;
}

template <int MODE>
static void run(const char* name, double lane_ops_per_iter_per_thread, int waves_per_simd) {
  float* out; CHECK(hipMalloc(&out, 4));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int blocks = cus * waves_per_simd;   // 256 threads = 4 waves = 1 per SIMD
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  const int reps = 5;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  const double ops = (double)blocks * 256 * ITERS * lane_ops_per_iter_per_thread;
  const double per_s = ops / (ms * 1e-3);
  printf("%-28s waves/SIMD=%d  %8.3f ms  %8.2f T lane-instr/s  = %6.1f lanes/clk/CU @2.4GHz (%d CUs)\n", name,
         waves_per_simd, ms, per_s / 1e12, per_s / cus / 2.4e9, cus);
  CHECK(hipFree(out));
}

int main() {
  for (int w : {1, 2, 4}) {
    run<0>("v_fma_f32", NACC, w);
    run<7>("v_fmac_f32", NACC, w);
    run<6>("v_mul_f32", NACC, w);
    run<4>("v_pk_fma_f32 (instr)", NACC / 2, w);
    run<1>("v_exp_f32", NACC, w);
    run<2>("v_rcp_f32", NACC, w);
    run<3>("v_rsq_f32", NACC, w);
    run<5>("6 fma + 1 exp (instr)", NACC * 7, w);
    run<8>("3 pk_fma + 1 exp (instr)", NACC / 2 * 4, w);
  }
  return 0;
}
