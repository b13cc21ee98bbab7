// What does it cost a VALU-bound wave to stream the ground-truth env rows (development tool, round 5)?
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_gtstream.hip -o tools/ubench_gtstream && tools/ubench_gtstream
// The fused objective's kernels (csrc/sgr_fused_recon.hip, sgr_pk.inl: fwd_pk_half_gt_kernel) read one table row of the ground-truth env
// per 32-pixel wave and row -- 3 colours x 32 pixels x 64 bytes, the pixels 512 bytes apart -- through six LDS-DMA requests
// (buffer_load_dwordx4 ... lds) next to ~1 200 packed VALU instructions.  tools/ablate.sh showed that the stream costs the objective backward
// 15 % although nothing waits for it.  This benchmark isolates the question: one wave = 32 pixels x 2 halves, `rows` rows per wave, per row
// `VALU_PER_ROW` dependent-chain-free v_pk_fma_f32 and
//   mode 0: no memory traffic at all,
//   mode 1: six LDS-DMA requests for the NEXT row (double-buffered 6 KB tiles) + the twelve ds_read_b64 a lane needs from the current one,
//   mode 2: the same bytes as six global_load_dwordx4 per lane straight into registers (lane = (pixel, half row): two 16-byte loads per colour),
//           consumed one row later,
// at two resident waves per SIMD (launch bounds + 20 KB of LDS per one-wave workgroup, in every mode).  Prints us per launch for each mode.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* LdsPtr;
constexpr int kPx = 32, J = 128, EH = 8, VALU_PER_ROW = 1200;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const float* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)bytes, 0x00020000);
}

template <int MODE>
__global__ __launch_bounds__(64, 2) void k(const float* __restrict__ gt, float* __restrict__ out, int RC, int tiles) {
  __shared__ __attribute__((aligned(16))) float tile[2 * 3 * kPx * 16 + 2048];      // 12 KB of tiles + padding = 20 KB per one-wave workgroup: eight per CU = two per SIMD, in EVERY mode
  const int lane = threadIdx.x, half = lane >> 5, pl = lane & 31;
  const int b = blockIdx.x / tiles, p0 = (blockIdx.x - b * tiles) * kPx;
  const float* img = gt + (size_t)b * 3 * RC * J;
  const __amdgpu_buffer_rsrc_t r = rsrc_of(img, (size_t)3 * RC * J * 4);
  f32x2 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x2{(float)lane, 1.0f + i};
  f32x2 sum = {0.f, 0.f};
  auto issue_dma = [&](float* dst, int e) {
    const int lrow = lane >> 2, slot = lane & 3;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int row = it * 16 + lrow;
      const int voff = (row * J + slot * 4) * 4;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int soff = (int)((((size_t)c * RC + p0) * J + e * 16) * 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (LdsPtr)(dst + (c * kPx + it * 16) * 16), 16, voff, soff, 0, 0);
      }
    }
  };
  f32x4 cur[6], nxt[6];
  auto issue_direct = [&](f32x4 (&dst)[6], int e) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* base = img + ((size_t)c * RC + p0 + pl) * J + e * 16 + half * 8;
      dst[2 * c] = *reinterpret_cast<const f32x4*>(base);
      dst[2 * c + 1] = *reinterpret_cast<const f32x4*>(base + 4);
    }
  };
  if (MODE == 1) issue_dma(tile, 0);
  if (MODE == 2) issue_direct(cur, 0);
  for (int e = 0; e < EH; ++e) {
    if (MODE == 1) {
      if (e + 1 < EH) { issue_dma(tile + ((e + 1) & 1) * 3 * kPx * 16, e + 1); asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (MODE == 2 && e + 1 < EH) issue_direct(nxt, e + 1);
#pragma unroll 1
    for (int blk = 0; blk < 4; ++blk) {
#pragma unroll
      for (int i = 0; i < VALU_PER_ROW / 4 / 8; ++i)
#pragma unroll
        for (int a = 0; a < 8; ++a) acc[a] = __builtin_elementwise_fma(acc[a], f32x2{1.0001f, 0.9999f}, f32x2{1e-3f, 1e-3f});
      if (MODE == 1) {      // a quarter of the row's reads per block: 3 x ds_read_b64, as tile32_read_pair does
        const float* t = tile + (e & 1) * 3 * kPx * 16;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const f32x2 v = *reinterpret_cast<const f32x2*>(t + (c * kPx + pl) * 16 + half * 8 + blk * 2);
          sum += v;
        }
      }
      if (MODE == 2) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const f32x4 v = cur[2 * c + (blk >> 1)];
          sum += (blk & 1) ? f32x2{v.z, v.w} : f32x2{v.x, v.y};
        }
      }
    }
    if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 6; ++i) cur[i] = nxt[i];
    }
  }
  f32x2 tot = sum;
#pragma unroll
  for (int i = 0; i < 8; ++i) tot += acc[i];
  if (MODE != 1) { tile[lane] = tot.x; __syncthreads(); tot.y += tile[63 - lane]; }      // keeps the allocation alive in the modes that do not stream through it
  out[(size_t)blockIdx.x * 64 + lane] = tot.x + tot.y;
}

int main() {
  const int bn = 16, R = 120, C = 160, RC = R * C, tiles = RC / kPx;
  const size_t n = (size_t)bn * 3 * RC * J;
  float *gt, *out;
  CHECK(hipMalloc(&gt, n * 4));
  CHECK(hipMemset(gt, 0, n * 4));
  CHECK(hipMalloc(&out, (size_t)bn * tiles * 64 * 4));
  hipStream_t st; CHECK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto run = [&](int mode) {
    auto launch = [&] {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(bn * tiles), dim3(64), 0, st, gt, out, RC, tiles);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(bn * tiles), dim3(64), 0, st, gt, out, RC, tiles);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(bn * tiles), dim3(64), 0, st, gt, out, RC, tiles);
    };
    for (int i = 0; i < 3; ++i) launch();
    CHECK(hipEventRecord(e0, st));
    for (int i = 0; i < 20; ++i) launch();
    CHECK(hipEventRecord(e1, st));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 20 * 1e3;
  };
  printf("# %d waves of 32 pixels x 2 halves, %d rows, %d v_pk_fma_f32 per row and lane; ground truth %.0f MB per launch\n", bn * tiles, EH, VALU_PER_ROW, n * 4 / 1e6);
  for (int rep = 0; rep < 3; ++rep)
    printf("no stream %.1f us | six LDS-DMA requests per row + ds_read_b64 %.1f us | six global_load_dwordx4 per lane and row %.1f us\n", run(0), run(1), run(2));
  return 0;
}
