// VGPR-bank / encoding microbenchmark (development tool): does v_fma_f32 slow down when its three
// source VGPRs share a bank (index mod 4), or when a source is an SGPR / literal?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITERS = 4096;

#define REP8(X) X X X X X X X X
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
  float r = seed + threadIdx.x;
  asm volatile("v_mov_b32 v20, %0\n v_mov_b32 v21, %0\n v_mov_b32 v22, %0\n v_mov_b32 v23, %0\n"
               "v_mov_b32 v24, %0\n v_mov_b32 v25, %0\n v_mov_b32 v26, %0\n v_mov_b32 v27, %0\n"
               "v_mov_b32 v28, %0\n v_mov_b32 v29, %0\n v_mov_b32 v30, %0\n v_mov_b32 v31, %0\n"
               "v_mov_b32 v32, 0x3f7fbe77\n v_mov_b32 v33, 0x3a83126f\n v_mov_b32 v36, 0x3f7fbe77\n v_mov_b32 v40, 0x3a83126f\n s_mov_b32 s20, 0x3f7fbe77\n"
               :: "v"(r) : "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v36","v40","s20");
  for (int it = 0; it < ITERS; ++it) {
    if (MODE == 0)   // 8 independent chains, sources in different banks: dst/src0 v20+i, src1 v33 (bank1), src2... 
      asm volatile(REP8("v_fma_f32 v20, v20, v33, v32\n v_fma_f32 v21, v21, v32, v33\n v_fma_f32 v22, v22, v33, v32\n v_fma_f32 v23, v23, v32, v33\n") ::: "v20","v21","v22","v23");
    if (MODE == 1)   // all three sources in bank 0: v20/v24/v28 with v32, v36, v40
      asm volatile(REP8("v_fma_f32 v20, v20, v32, v36\n v_fma_f32 v24, v24, v36, v40\n v_fma_f32 v28, v28, v40, v32\n v_fma_f32 v20, v20, v36, v40\n") ::: "v20","v24","v28");
    if (MODE == 2)   // SGPR as src1
      asm volatile(REP8("v_fma_f32 v20, v20, s20, v33\n v_fma_f32 v21, v21, s20, v32\n v_fma_f32 v22, v22, s20, v33\n v_fma_f32 v23, v23, s20, v32\n") ::: "v20","v21","v22","v23");
    if (MODE == 3)   // VOP2 fmac, different banks
      asm volatile(REP8("v_fmac_f32 v20, v33, v21\n v_fmac_f32 v21, v32, v22\n v_fmac_f32 v22, v33, v23\n v_fmac_f32 v23, v32, v20\n") ::: "v20","v21","v22","v23");
    if (MODE == 4)   // VOP2 fmac, same bank (v20,v24,v28 / v32,v36)
      asm volatile(REP8("v_fmac_f32 v20, v32, v24\n v_fmac_f32 v24, v36, v28\n v_fmac_f32 v28, v32, v20\n v_fmac_f32 v20, v36, v24\n") ::: "v20","v24","v28");
    if (MODE == 5)   // fma with neg modifier + sgpr (like fma(-s, U, C))
      asm volatile(REP8("v_fma_f32 v20, -s20, v21, v33\n v_fma_f32 v21, -s20, v22, v32\n v_fma_f32 v22, -s20, v23, v33\n v_fma_f32 v23, -s20, v20, v32\n") ::: "v20","v21","v22","v23");
  }
  float s;
  asm volatile("v_add_f32 %0, v20, v21\n v_add_f32 %0, %0, v24\n v_add_f32 %0, %0, v28" : "=v"(s) :: "v20","v21","v24","v28");
  if (s == 12345.678f) out[0] = s;
}
template <int MODE> static void run(const char* name) {
  float* out; CHECK(hipMalloc(&out, 4));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  for (int w : {1, 2, 4}) {
    const int blocks = prop.multiProcessorCount * w;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0f); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double instr = (double)blocks * 4 /*waves*/ * ITERS * 32;
    const double per_simd_per_s = instr / (ms * 1e-3) / (prop.multiProcessorCount * 4);
    printf("%-36s waves/SIMD=%d  %7.3f ms  %6.2f cycles/wave-instr/SIMD @2.3GHz\n", name, w, ms, 2.3e9 / per_simd_per_s);
  }
  CHECK(hipFree(out));
}
int main() {
  run<0>("v_fma_f32 srcs in distinct banks");
  run<1>("v_fma_f32 3 srcs in ONE bank");
  run<2>("v_fma_f32 with SGPR src");
  run<5>("v_fma_f32 -SGPR src (neg mod)");
  run<3>("v_fmac_f32 distinct banks");
  run<4>("v_fmac_f32 one bank");
  return 0;
}
