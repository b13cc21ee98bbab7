// What does FETCH_SIZE (rocprofv3 --pmc) count for the access patterns of this repository?  (development tool, round 6)
//   hipcc -O3 --offload-arch=gfx950 tools/fetch_calib.hip -o tools/fetch_calib
//   tools/fetch_calib                                          # times every pattern: useful GB/s
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/fetch_calib 1      # one launch per pattern, counters per kernel name
// MI355X_MICROARCH.md calibrates the counter for ONE pattern -- a wide coalesced stream, 16 B per lane: FETCH_SIZE reports half the bytes
// (128-byte requests tallied at 64) -- and calls every other width uncalibrated.  tools/parse_pmc.py doubles FETCH_SIZE for every kernel;
// the objective's backward (csrc/sgr_fused_recon.hip) reads its ground truth as 64-byte HALF lines through LDS-DMA, a row of arithmetic
// apart, and showed 1.33-1.50 x its algorithmic bytes (Weak 4: counter artefact or real re-fetch?).
//
// Every kernel below touches a KNOWN number of bytes of a 1 GiB buffer (four times L2 + Infinity Cache), each byte at most once per launch:
//   wide16        every byte, 16 B per lane, consecutive lanes consecutive                          -> N useful bytes      (the guide's case)
//   dword4        every byte, 4 B per lane                                                          -> N
//   half_lines    ONLY the first 64 bytes of every 128-byte line (16 B per lane, 4 lanes per line)  -> N/2 useful; N/2 from memory if the L2
//                 fills 64-byte sectors, N if it fills whole lines -- the TIME against wide16 says which, independently of the counter
//   half_far      first halves of all lines, then (1 GiB later, nothing can still be cached) the second halves -> N useful; N from memory if
//                 sectored, 2 N if whole lines
//   dma_rows<V>   the objective's own pattern (tools/ubench_gtstream mode 1): per wave and row 3 colours x 32 pixels x 64 bytes by six
//                 buffer_load_dwordx4 ... lds, rows e and e+1 of a pixel = the two halves of one line, V packed FMAs per lane between them
//                 -> N useful (every byte of a [bn,3,RC,128] image once)
// Output per pattern: useful bytes, us, useful GB/s.  With the counter pass: raw FETCH_SIZE x 1024 / useful bytes = the factor to apply.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* LdsPtr;

constexpr size_t kBytes = (size_t)1 << 30;
constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void wide16(const f32x4* __restrict__ p, float* __restrict__ out, size_t n16) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n16; i += stride) acc += __builtin_nontemporal_load(p + i);
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.0f;
}
__global__ __launch_bounds__(kBlock) void dword4(const float* __restrict__ p, float* __restrict__ out, size_t n4) {
  float acc = 0.f;
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) acc += __builtin_nontemporal_load(p + i);
  if (acc == 123.456f) out[0] = 1.0f;
}
// lane l of a group of four reads 16 B at byte (line * 128 + half * 64 + l * 16)
template <int HALF>
__device__ __forceinline__ f32x4 half_line_sum(const f32x4* __restrict__ p, size_t lines) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < lines * 4; t += stride) {
    const size_t line = t >> 2, l = t & 3;
    acc += p[line * 8 + HALF * 4 + l];
  }
  return acc;
}
__global__ __launch_bounds__(kBlock) void half_lines(const f32x4* __restrict__ p, float* __restrict__ out, size_t lines) {
  const f32x4 a = half_line_sum<0>(p, lines);
  if (a.x + a.y + a.z + a.w == 123.456f) out[0] = 1.0f;
}
__global__ __launch_bounds__(kBlock) void half_far_first(const f32x4* __restrict__ p, float* __restrict__ out, size_t lines) {
  const f32x4 a = half_line_sum<0>(p, lines);
  if (a.x + a.y + a.z + a.w == 123.456f) out[0] = 1.0f;
}
__global__ __launch_bounds__(kBlock) void half_far_second(const f32x4* __restrict__ p, float* __restrict__ out, size_t lines) {
  const f32x4 a = half_line_sum<1>(p, lines);
  if (a.x + a.y + a.z + a.w == 123.456f) out[0] = 1.0f;
}

constexpr int kPx = 32, J = 128, EH = 8;
template <int VALU_PER_ROW>
__global__ __launch_bounds__(64, 2) void dma_rows(const float* __restrict__ gt, float* __restrict__ out, int RC, int tiles) {
  __shared__ __attribute__((aligned(16))) float tile[2 * 3 * kPx * 16 + 2048];      // 20 KB per one-wave workgroup: two waves per SIMD
  const int lane = threadIdx.x, half = lane >> 5, pl = lane & 31;
  const int b = blockIdx.x / tiles, p0 = (blockIdx.x - b * tiles) * kPx;
  const float* img = gt + (size_t)b * 3 * RC * J;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(img), 0, (int)((size_t)3 * RC * J * 4), 0x00020000);
  f32x2 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x2{(float)lane, 1.0f + i};
  f32x2 sum = {0.f, 0.f};
  auto issue_dma = [&](float* dst, int e) {
    const int lrow = lane >> 2, slot = lane & 3;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int voff = ((it * 16 + lrow) * J + slot * 4) * 4;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int soff = (int)((((size_t)c * RC + p0) * J + e * 16) * 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (LdsPtr)(dst + (c * kPx + it * 16) * 16), 16, voff, soff, 0, 0);
      }
    }
  };
  issue_dma(tile, 0);
  for (int e = 0; e < EH; ++e) {
    if (e + 1 < EH) { issue_dma(tile + ((e + 1) & 1) * 3 * kPx * 16, e + 1); asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll 1
    for (int blk = 0; blk < 4; ++blk) {
#pragma unroll
      for (int i = 0; i < VALU_PER_ROW / 4 / 8; ++i)
#pragma unroll
        for (int a = 0; a < 8; ++a) acc[a] = __builtin_elementwise_fma(acc[a], f32x2{1.0001f, 0.9999f}, f32x2{1e-3f, 1e-3f});
      const float* t = tile + (e & 1) * 3 * kPx * 16;
#pragma unroll
      for (int c = 0; c < 3; ++c) sum += *reinterpret_cast<const f32x2*>(t + (c * kPx + pl) * 16 + half * 8 + blk * 2);
    }
  }
  f32x2 tot = sum;
#pragma unroll
  for (int i = 0; i < 8; ++i) tot += acc[i];
  out[(size_t)blockIdx.x * 64 + lane] = tot.x + tot.y;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 10;
  float *buf, *out;
  CHECK(hipMalloc(&buf, kBytes));
  CHECK(hipMemset(buf, 0, kBytes));
  // the objective's image: [bn, 3, RC, 128] floats with RC = 19200 (120 x 160): 29.5 MB per image, 32 images = 944 MB <= 1 GiB
  const int bn = 32, RC = 120 * 160, tiles = RC / kPx;
  const size_t img_bytes = (size_t)bn * 3 * RC * J * 4;
  CHECK(hipMalloc(&out, (size_t)bn * tiles * 64 * 4));
  hipStream_t st; CHECK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const dim3 grid(256 * 16), block(kBlock);
  const size_t lines = kBytes / 128;
  auto time_us = [&](auto launch) {
    launch();      // warm the code, not the data: every launch streams 1 GiB, four times what the caches hold
    CHECK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(e1, st));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps * 1e3;
  };
  auto report = [&](const char* name, double useful, double us) { printf("%-28s useful %8.1f MB  %8.1f us  %7.1f GB/s useful\n", name, useful / 1e6, us, useful / us / 1e3); };
  report("wide16", (double)kBytes, time_us([&] { hipLaunchKernelGGL(wide16, grid, block, 0, st, (const f32x4*)buf, out, kBytes / 16); }));
  report("dword4", (double)kBytes, time_us([&] { hipLaunchKernelGGL(dword4, grid, block, 0, st, buf, out, kBytes / 4); }));
  report("half_lines", (double)kBytes / 2, time_us([&] { hipLaunchKernelGGL(half_lines, grid, block, 0, st, (const f32x4*)buf, out, lines); }));
  report("half_far (first + second)", (double)kBytes, time_us([&] {
    hipLaunchKernelGGL(half_far_first, grid, block, 0, st, (const f32x4*)buf, out, lines);
    hipLaunchKernelGGL(half_far_second, grid, block, 0, st, (const f32x4*)buf, out, lines);
  }));
  report("dma_rows<0>", (double)img_bytes, time_us([&] { hipLaunchKernelGGL(dma_rows<0>, dim3(bn * tiles), dim3(64), 0, st, buf, out, RC, tiles); }));
  report("dma_rows<320>", (double)img_bytes, time_us([&] { hipLaunchKernelGGL(dma_rows<320>, dim3(bn * tiles), dim3(64), 0, st, buf, out, RC, tiles); }));
  report("dma_rows<1280>", (double)img_bytes, time_us([&] { hipLaunchKernelGGL(dma_rows<1280>, dim3(bn * tiles), dim3(64), 0, st, buf, out, RC, tiles); }));
  CHECK(hipDeviceSynchronize());
  return 0;
}
