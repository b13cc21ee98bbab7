// Go / no-go microbenchmark: can v_mfma_f32_4x4x1_16B_f32 take the 3 K J colour accumulation
// off the VALU port next to the v_exp_f32-bound stream of the SG inner loop?
//
//   hipcc -O2 --offload-arch=gfx950 tools/ubench_mfma.hip -o tools/ubench_mfma && tools/ubench_mfma
//
// The forward's inner loop per lobe and azimuth quad is  8 packed (exponents) + 8 v_exp_f32 + 12 packed (3 colours x 4 pairs)
// = 512 (pixel, direction, lobe) items per wave.  In a lanes <-> (16 pixels x 4 directions) layout one
// v_mfma_f32_4x4x1_16B_f32 (A = w[k][colour] in lane 4 p + i, B = E[k][direction] in lane 4 p + j, D += A x B: 16 blocks of
// 4 x 4) does the colour accumulation of one v_exp_f32's worth of items, i.e. the same 512 items cost
// 6 packed (exponents) + 8 v_exp_f32 + 8 MFMA.  Whether that wins depends on what an MFMA costs the SIMD's issue port
// while transcendentals and packed FMAs are in flight -- measured here, per wave-instruction, at 1 / 2 / 3 waves per SIMD:
//
//   exp8            8 v_exp_f32
//   pk20            20 v_pk_fma_f32
//   cur             8 v_pk_fma + 8 v_exp + 12 v_pk_fma                      (the shipped inner loop)
//   mfma8           8 v_mfma_f32_4x4x1
//   exp8_mfma8      8 v_exp + 8 v_mfma (interleaved)
//   new             6 v_pk_fma + 8 v_exp + 8 v_mfma                         (the candidate inner loop)
//   new_valu        6 v_pk_fma + 8 v_exp + 8 v_pk_fma                       (the candidate if an MFMA cost exactly one packed FMA)
//
// Reported: SIMD cycles per loop iteration = kernel cycles (s_memtime at both ends of one wave, max over waves) / iterations,
// also divided by the number of waves resident on the SIMD (cycles of SIMD time per iteration of work).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PK(acc, a, b) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define EXP(d, s) asm volatile("v_exp_f32 %0, %1" : "=v"(d) : "v"(s))
#define MFMA(acc, a, b) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

template <int MODE>
__global__ __launch_bounds__(64) void kern(float* out, unsigned long long* cyc, int iters, float seed) {
  const int lane = threadIdx.x;
  f32x2 p[12];
  f32x4 m[8];
  float e[8], t[8];
  const f32x2 ca = {seed + lane * 1e-3f, seed * 0.5f}, cb = {0.999f, 0.998f};
#pragma unroll
  for (int i = 0; i < 12; ++i) p[i] = f32x2{seed * i, seed + i};
#pragma unroll
  for (int i = 0; i < 8; ++i) { m[i] = f32x4{seed, seed * i, 1.f, 2.f}; t[i] = -seed * (i + 1) - lane * 1e-4f; e[i] = 0.f; }
  f32x2 q[4] = {f32x2{seed, seed}, f32x2{seed * 2, seed}, f32x2{seed * 3, seed}, f32x2{seed * 4, seed}};
  const float a1 = seed * 0.25f, b1 = seed * 0.125f + lane * 1e-5f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {            // exp8
#pragma unroll
      for (int i = 0; i < 8; ++i) EXP(e[i], t[i]);
    } else if (MODE == 1) {     // pk20
#pragma unroll
      for (int i = 0; i < 12; ++i) PK(p[i], ca, cb);
#pragma unroll
      for (int i = 0; i < 8; ++i) PK(p[i], cb, ca);
    } else if (MODE == 2) {     // cur: 8 pk + 8 exp + 12 pk, interleaved like the shipped loop (2 pk, 2 exp, 3 pk per pair)
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        PK(q[h], ca, cb); PK(q[(h + 1) & 3], cb, ca);
        EXP(e[2 * h], t[2 * h]); EXP(e[2 * h + 1], t[2 * h + 1]);
        PK(p[3 * h], ca, cb); PK(p[3 * h + 1], ca, cb); PK(p[3 * h + 2], ca, cb);
      }
    } else if (MODE == 3) {     // mfma8
#pragma unroll
      for (int i = 0; i < 8; ++i) MFMA(m[i], a1, b1);
    } else if (MODE == 4) {     // exp8_mfma8
#pragma unroll
      for (int i = 0; i < 8; ++i) { EXP(e[i], t[i]); MFMA(m[i], a1, e[(i + 4) & 7]); }
    } else if (MODE == 5) {     // new: 6 pk + 8 exp + 8 mfma  (per 4 exps: 3 pk, 4 exp, 4 mfma)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        PK(q[0], ca, cb); PK(q[1], cb, ca); PK(q[2], ca, cb);
#pragma unroll
        for (int i = 0; i < 4; ++i) { EXP(e[4 * h + i], t[4 * h + i]); MFMA(m[4 * h + i], a1, e[(4 * h + i + 4) & 7]); }
      }
    } else if (MODE == 6) {     // new_valu: 6 pk + 8 exp + 8 pk
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        PK(q[0], ca, cb); PK(q[1], cb, ca); PK(q[2], ca, cb);
#pragma unroll
        for (int i = 0; i < 4; ++i) { EXP(e[4 * h + i], t[4 * h + i]); PK(p[4 * h + i], ca, cb); }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 12; ++i) s += p[i].x + p[i].y;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += m[i].x + m[i].y + m[i].z + m[i].w + e[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += q[i].x + q[i].y;
  out[blockIdx.x * 64 + lane] = s;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  int cus = 0, dev = 0;
  CHECK(hipGetDevice(&dev));
  CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int simds = cus * 4, iters = 20000;
  float* out; unsigned long long* cyc;
  CHECK(hipMalloc(&out, (size_t)simds * 4 * 64 * 4)); CHECK(hipMalloc(&cyc, (size_t)simds * 4 * 8));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const char* names[7] = {"exp8", "pk20", "cur  (8 pk + 8 exp + 12 pk)", "mfma8", "exp8_mfma8", "new  (6 pk + 8 exp + 8 mfma)", "new_valu (6 pk + 8 exp + 8 pk)"};
  printf("# %d CUs, %d SIMDs, %d iterations per wave; s_memtime ticks at 100 MHz are converted with the event time\n", cus, simds, iters);
  printf("%-34s %6s %12s %16s %18s\n", "loop body", "waves", "kernel us", "ns / iteration", "ns / iter / wave");
  for (int mode = 0; mode < 7; ++mode)
    for (int w = 1; w <= 3; ++w) {
      const int blocks = simds * w;
      auto launch = [&]() {
        switch (mode) {
          case 0: hipLaunchKernelGGL(kern<0>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 0.001f); break;
          case 1: hipLaunchKernelGGL(kern<1>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 0.001f); break;
          case 2: hipLaunchKernelGGL(kern<2>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 0.001f); break;
          case 3: hipLaunchKernelGGL(kern<3>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 0.001f); break;
          case 4: hipLaunchKernelGGL(kern<4>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 0.001f); break;
          case 5: hipLaunchKernelGGL(kern<5>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 0.001f); break;
          default: hipLaunchKernelGGL(kern<6>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 0.001f); break;
        }
      };
      launch(); CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0, 0));
      for (int r = 0; r < 3; ++r) launch();
      CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / 3.0;
      printf("%-34s %6d %12.1f %16.2f %18.2f\n", names[mode], w, us, us * 1e3 / iters, us * 1e3 / iters / w);
    }
  return 0;
}
