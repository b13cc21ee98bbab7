// Shared device-side plumbing for the sgrender kernels (gfx950 only).
//
// Work decomposition used by every hot kernel ("lanes <-> pixels"):
//   * one 64-lane wavefront owns 64 consecutive env-grid cells of one image (flat index
//     p = r*C + c), so every [.., R, C] plane is read as one fully coalesced 256-byte row;
//   * the K lobes' parameters and the pixel's shading frame live in VGPRs for the whole
//     kernel, the quadrature directions are wave-uniform and arrive through scalar loads;
//   * the env image is [.., p, j]-major (512 B per pixel for J=128), i.e. lane-strided, so
//     env tiles of 64 pixels x 32 directions are transposed through LDS and move to/from HBM
//     as full 128-byte lines.
#pragma once

#include <hip/hip_runtime.h>

#include "sgr_math.h"

namespace sgr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// The direction table is read-only for the lifetime of every kernel and indexed wave-uniformly.
// Reading it through the constant address space makes the compiler emit scalar loads
// (s_load_dwordx4 -> SGPR operands) instead of per-lane vector loads, even though the kernel
// also stores to global memory through other pointers.
typedef const f32x4 __attribute__((address_space(4))) * DirTable;
__device__ __forceinline__ DirTable as_dir_table(const float4* p) { return (DirTable)(p); }

constexpr int kWave = 64;        // pixels per workgroup (one wavefront)
// LDS env tile: 64 pixels x TJ directions x RGB, rows padded by 4 dwords (16-B aligned rows,
// conflict-free ds_write_b128 / ds_read_b128 by the owning lane).  TJ = 32 moves full 128-byte
// lines (27.6 KB per wave, 5 waves/CU); TJ = 16 moves 64-byte segments (15.4 KB, 10 waves/CU).
template <int TJ> struct Tile {
  static constexpr int kStride = TJ + 4;
  static constexpr int kFloats = 3 * kWave * kStride;
  static constexpr int kLanesPerRow = TJ / 4;           // lanes covering one pixel's segment
  static constexpr int kRowsPerIt = kWave / kLanesPerRow;
  static constexpr int kIts = kWave / kRowsPerIt;
};

// ---- env tile <-> global (coalesced: 8 lanes cover one pixel's 128-byte segment) ----------
// tile[c][row][col], row = pixel within the wave's run, col = direction within the chunk.
template <int TJ, bool VEC>
__device__ __forceinline__ void tile_store_global(const float* tile, float* __restrict__ env_img /* [3,RC,J] of image b */,
                                                  int p0, int RC, int J, int j0, int lane) {
#pragma unroll 1   // one colour (<= 32 VGPRs of payload) in flight at a time
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int it = 0; it < Tile<TJ>::kIts; ++it) {
      const int row = it * Tile<TJ>::kRowsPerIt + lane / Tile<TJ>::kLanesPerRow;
      const int col = (lane % Tile<TJ>::kLanesPerRow) * 4;
      const int px = p0 + row;
      const int j = j0 + col;
      const float4 v = *reinterpret_cast<const float4*>(tile + (c * kWave + row) * Tile<TJ>::kStride + col);
      if (px < RC) {
        float* dst = env_img + ((size_t)c * RC + px) * J + j;
        if (VEC) {
          if (j < J) {
            f32x4 nv = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(nv, reinterpret_cast<f32x4*>(dst));
          }
        } else {
          if (j + 0 < J) dst[0] = v.x;
          if (j + 1 < J) dst[1] = v.y;
          if (j + 2 < J) dst[2] = v.z;
          if (j + 3 < J) dst[3] = v.w;
        }
      }
    }
  }
}

template <int TJ, bool VEC>
__device__ __forceinline__ void tile_load_global(float* tile, const float* __restrict__ env_img, int p0, int RC, int J,
                                                 int j0, int lane) {
#pragma unroll 1   // one colour (<= 32 VGPRs of payload) in flight at a time
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int it = 0; it < Tile<TJ>::kIts; ++it) {
      const int row = it * Tile<TJ>::kRowsPerIt + lane / Tile<TJ>::kLanesPerRow;
      const int col = (lane % Tile<TJ>::kLanesPerRow) * 4;
      const int px = p0 + row;
      const int j = j0 + col;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (px < RC) {
        const float* src = env_img + ((size_t)c * RC + px) * J + j;
        if (VEC) {
          if (j < J) {
            const f32x4 nv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src));
            v = make_float4(nv.x, nv.y, nv.z, nv.w);
          }
        } else {
          if (j + 0 < J) v.x = src[0];
          if (j + 1 < J) v.y = src[1];
          if (j + 2 < J) v.z = src[2];
          if (j + 3 < J) v.w = src[3];
        }
      }
      *reinterpret_cast<float4*>(tile + (c * kWave + row) * Tile<TJ>::kStride + col) = v;
    }
  }
}

// ---- pooled BRDF-map fetch ---------------------------------------------------------------
// POOL == 1: maps are already on the env grid.  POOL == 2: 2x2 average (the integer-ratio
// case of F.adaptive_avg_pool2d, models.py:465-469), read as two 8-byte loads per plane.
template <int POOL>
__device__ __forceinline__ float fetch_pooled(const float* __restrict__ plane, int r, int c, int imW) {
  if (POOL == 1) {
    return plane[(size_t)r * imW + c];
  } else {
    const float2 t = *reinterpret_cast<const float2*>(plane + (size_t)(2 * r) * imW + 2 * c);
    const float2 u = *reinterpret_cast<const float2*>(plane + (size_t)(2 * r + 1) * imW + 2 * c);
    return (((t.x + t.y) + u.x) + u.y) * 0.25f;
  }
}

// ---- kernel argument block (superset used by every hot kernel) ------------------------------
struct Args {
  // BRDF maps [bn,{3,3,1},imH,imW]
  const float* albedo;
  const float* normal;
  const float* rough;
  // SG parameters (raw decoder outputs when premap != 0, post-tan otherwise)
  const float* axis;      // [bn,K,3,R,C]
  const float* lamb;      // [bn,K,R,C]
  const float* weight;    // [bn,3K,R,C]   channel k*3 + rgb
  const float* env_in;    // [bn,3,R,C,J]
  const float4* dirs;     // [Jpad] (lx, ly, lz, omega)
  const float* view;      // [3,R,C]
  // cotangents
  const float* g_env;     // [bn,3,R,C,J]
  const float* g_diffuse; // [bn,3,R,C]
  const float* g_spec;    // [bn,3,R,C]
  // outputs
  float* env_out;         // [bn,3,R,C,J]
  float* lamb_tan;        // [bn,K,R,C]     nullable
  float* weight_tan;      // [bn,3K,R,C]    nullable
  float* diffuse;         // [bn,3,R,C]
  float* spec;            // [bn,3,R,C]
  float* g_axis;          // [bn,K,3,R,C]
  float* g_lamb;          // [bn,K,R,C]
  float* g_weight;        // [bn,3K,R,C]
  float* g_env_out;       // [bn,3,R,C,J]
  float* g_albedo;        // [bn,3,imH,imW]
  float* g_normal;        // [bn,3,imH,imW]
  float* g_rough;         // [bn,1,imH,imW]
  int bn, K, R, C, J, Jpad, imH, imW;
  float F0;
  int premap;
};

// Which pixel does this lane own?  One wave = 64 consecutive cells of one image.
struct Pix {
  int b, p0, p, lane;
  bool active;
};
__device__ __forceinline__ Pix locate(const Args& a) {
  Pix x;
  x.lane = threadIdx.x;
  const int RC = a.R * a.C;
  const int tiles = (RC + kWave - 1) / kWave;
  x.b = blockIdx.x / tiles;
  x.p0 = (blockIdx.x - x.b * tiles) * kWave;
  x.active = (x.p0 + x.lane) < RC;
  x.p = x.active ? (x.p0 + x.lane) : (RC - 1);
  return x;
}
static inline dim3 wave_grid(int bn, int R, int C) {
  return dim3((unsigned)(bn * ((R * C + kWave - 1) / kWave)));
}

// Pooled BRDF maps + shading frame of the lane's pixel.  pooled[7] = albedo rgb, normal xyz, rough.
template <int POOL>
__device__ __forceinline__ Frame load_frame_pooled(const Args& a, const Pix& x, float pooled[7]) {
  const int RC = a.R * a.C;
  const int r = x.p / a.C, c = x.p - r * a.C;
  const size_t plane = (size_t)a.imH * a.imW;
  const float* al = a.albedo + (size_t)x.b * 3 * plane;
  const float* no = a.normal + (size_t)x.b * 3 * plane;
  const float* ro = a.rough + (size_t)x.b * plane;
  pooled[0] = fetch_pooled<POOL>(al, r, c, a.imW);
  pooled[1] = fetch_pooled<POOL>(al + plane, r, c, a.imW);
  pooled[2] = fetch_pooled<POOL>(al + 2 * plane, r, c, a.imW);
  pooled[3] = fetch_pooled<POOL>(no, r, c, a.imW);
  pooled[4] = fetch_pooled<POOL>(no + plane, r, c, a.imW);
  pooled[5] = fetch_pooled<POOL>(no + 2 * plane, r, c, a.imW);
  pooled[6] = fetch_pooled<POOL>(ro, r, c, a.imW);
  return make_frame(pooled[3], pooled[4], pooled[5], pooled[6], a.view[x.p], a.view[RC + x.p], a.view[2 * RC + x.p]);
}
template <int POOL>
__device__ __forceinline__ Frame load_frame(const Args& a, const Pix& x, float alb[3]) {
  float pooled[7];
  const Frame f = load_frame_pooled<POOL>(a, x, pooled);
  alb[0] = pooled[0]; alb[1] = pooled[1]; alb[2] = pooled[2];
  return f;
}

// Adjoint of fetch_pooled: the env cell's gradient goes to its POOL x POOL image pixels / POOL^2.
template <int POOL>
__device__ __forceinline__ void scatter_pooled(float* __restrict__ plane, int r, int c, int imW, float g) {
  if (POOL == 1) {
    plane[(size_t)r * imW + c] = g;
  } else {
    const float q = 0.25f * g;
    *reinterpret_cast<float2*>(plane + (size_t)(2 * r) * imW + 2 * c) = make_float2(q, q);
    *reinterpret_cast<float2*>(plane + (size_t)(2 * r + 1) * imW + 2 * c) = make_float2(q, q);
  }
}

// Per-lane access to the lane's own row of the RGB tile (4 consecutive directions).
template <int TJ>
__device__ __forceinline__ void tile_row_read(const float* tile, int lane, int jj, float e0[4], float e1[4], float e2[4]) {
  const float4 t0 = *reinterpret_cast<const float4*>(tile + (0 * kWave + lane) * Tile<TJ>::kStride + jj);
  const float4 t1 = *reinterpret_cast<const float4*>(tile + (1 * kWave + lane) * Tile<TJ>::kStride + jj);
  const float4 t2 = *reinterpret_cast<const float4*>(tile + (2 * kWave + lane) * Tile<TJ>::kStride + jj);
  e0[0] = t0.x; e0[1] = t0.y; e0[2] = t0.z; e0[3] = t0.w;
  e1[0] = t1.x; e1[1] = t1.y; e1[2] = t1.z; e1[3] = t1.w;
  e2[0] = t2.x; e2[1] = t2.y; e2[2] = t2.z; e2[3] = t2.w;
}
template <int TJ>
__device__ __forceinline__ void tile_row_write(float* tile, int lane, int jj, const float e0[4], const float e1[4], const float e2[4]) {
  *reinterpret_cast<float4*>(tile + (0 * kWave + lane) * Tile<TJ>::kStride + jj) = make_float4(e0[0], e0[1], e0[2], e0[3]);
  *reinterpret_cast<float4*>(tile + (1 * kWave + lane) * Tile<TJ>::kStride + jj) = make_float4(e1[0], e1[1], e1[2], e1[3]);
  *reinterpret_cast<float4*>(tile + (2 * kWave + lane) * Tile<TJ>::kStride + jj) = make_float4(e2[0], e2[1], e2[2], e2[3]);
}

}  // namespace sgr
