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

#ifndef SGR_DMA_AUX
#define SGR_DMA_AUX 0   // cache policy of the LDS-DMA row loads: 0 default, 2 non-temporal (A/B switch)
#endif
#ifndef SGR_NT
#define SGR_NT 1   // non-temporal hint on the streamed env tiles (A/B switch)
#endif
__device__ __forceinline__ void stream_store(f32x4 v, f32x4* p) {
#if SGR_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
__device__ __forceinline__ f32x4 stream_load(const f32x4* p) {
#if SGR_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}

// The direction table is read-only for the lifetime of every kernel and indexed wave-uniformly.
// Reading it through the constant address space makes the compiler emit scalar loads
// (s_load_dwordx4 -> SGPR operands) instead of per-lane vector loads, even though the kernel
// also stores to global memory through other pointers.
typedef const f32x4 __attribute__((address_space(4))) * DirTable;
__device__ __forceinline__ DirTable as_dir_table(const float4* p) { return (DirTable)(p); }
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef const f32x8 __attribute__((address_space(4))) * SepTable;
__device__ __forceinline__ SepTable as_sep_table(const float* p) { return (SepTable)(p); }

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
// Addressing discipline (all hot kernels): a wave-uniform 64-bit base (SGPR pair) plus ONE 32-bit
// per-lane offset, so global accesses select the `saddr + voffset` form and no 64-bit per-lane
// addresses are kept live in VGPRs (hoisted per-lane addresses were the dominant register hog).
template <int TJ, bool VEC>
__device__ __forceinline__ void tile_store_global(const float* tile, float* __restrict__ env_img /* [3,RC,J] of image b */,
                                                  int p0, int RC, int J, int j0, int lane) {
  const int lrow = lane / Tile<TJ>::kLanesPerRow;
  const int col = (lane % Tile<TJ>::kLanesPerRow) * 4;
  const unsigned lane_off = (unsigned)(lrow * J + col);
  const int rows_valid = RC - p0;
  const bool col_ok = (j0 + col) < J;
  if (TJ == 16 && VEC) {
    // 16-wide tiles: all 12 LDS reads in flight at once (48 VGPRs), one wait, then 12 stores
    float4 v[3][Tile<TJ>::kIts];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int it = 0; it < Tile<TJ>::kIts; ++it)
        v[c][it] = *reinterpret_cast<const float4*>(tile + (c * kWave + it * Tile<TJ>::kRowsPerIt + lrow) * Tile<TJ>::kStride + col);
    const bool full = rows_valid >= kWave && (j0 + TJ) <= J;   // wave-uniform: every lane stores (all but a ragged last tile)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float* cbase = env_img + ((size_t)c * RC + p0) * J + j0;   // wave-uniform
#pragma unroll
      for (int it = 0; it < Tile<TJ>::kIts; ++it) {
        const int row = it * Tile<TJ>::kRowsPerIt + lrow;
        float* dst = cbase + (size_t)(it * Tile<TJ>::kRowsPerIt) * J;   // uniform
        f32x4 nv = {v[c][it].x, v[c][it].y, v[c][it].z, v[c][it].w};
        if (full) {
          stream_store(nv, reinterpret_cast<f32x4*>(dst + lane_off));
        } else if (row < rows_valid && col_ok) {
          stream_store(nv, reinterpret_cast<f32x4*>(dst + lane_off));
        }
      }
    }
    return;
  }
#pragma unroll 1   // one colour (<= 32 VGPRs of payload) in flight at a time
  for (int c = 0; c < 3; ++c) {
    float* cbase = env_img + ((size_t)c * RC + p0) * J + j0;   // wave-uniform
#pragma unroll
    for (int it = 0; it < Tile<TJ>::kIts; ++it) {
      const int row = it * Tile<TJ>::kRowsPerIt + lrow;
      const float4 v = *reinterpret_cast<const float4*>(tile + (c * kWave + row) * Tile<TJ>::kStride + col);
      float* dst = cbase + (size_t)(it * Tile<TJ>::kRowsPerIt) * J;   // uniform
      if (row < rows_valid) {
        if (VEC) {
          if (col_ok) {
            f32x4 nv = {v.x, v.y, v.z, v.w};
            stream_store(nv, reinterpret_cast<f32x4*>(dst + lane_off));
          }
        } else {
          const int j = j0 + col;
          if (j + 0 < J) dst[lane_off + 0] = v.x;
          if (j + 1 < J) dst[lane_off + 1] = v.y;
          if (j + 2 < J) dst[lane_off + 2] = v.z;
          if (j + 3 < J) dst[lane_off + 3] = v.w;
        }
      }
    }
  }
}

template <int TJ, bool VEC>
__device__ __forceinline__ void tile_load_global(float* tile, const float* __restrict__ env_img, int p0, int RC, int J,
                                                 int j0, int lane) {
  const int lrow = lane / Tile<TJ>::kLanesPerRow;
  const int col = (lane % Tile<TJ>::kLanesPerRow) * 4;
  const unsigned lane_off = (unsigned)(lrow * J + col);
  const int rows_valid = RC - p0;
  const bool col_ok = (j0 + col) < J;
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {
    const float* cbase = env_img + ((size_t)c * RC + p0) * J + j0;   // wave-uniform
#pragma unroll
    for (int it = 0; it < Tile<TJ>::kIts; ++it) {
      const int row = it * Tile<TJ>::kRowsPerIt + lrow;
      const float* src = cbase + (size_t)(it * Tile<TJ>::kRowsPerIt) * J;   // uniform
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < rows_valid) {
        if (VEC) {
          if (col_ok) {
            const f32x4 nv = stream_load(reinterpret_cast<const f32x4*>(src + lane_off));
            v = make_float4(nv.x, nv.y, nv.z, nv.w);
          }
        } else {
          const int j = j0 + col;
          if (j + 0 < J) v.x = src[lane_off + 0];
          if (j + 1 < J) v.y = src[lane_off + 1];
          if (j + 2 < J) v.z = src[lane_off + 2];
          if (j + 3 < J) v.w = src[lane_off + 3];
        }
      }
      *reinterpret_cast<float4*>(tile + (c * kWave + row) * Tile<TJ>::kStride + col) = v;
    }
  }
}

// ---- env tile via LDS-DMA (global_load_lds_dwordx4): no VGPR staging, loads stay in flight -------
// One wave-instruction moves 64 x 16 B = 1 KB from per-lane global addresses to a LINEAR 1 KB of LDS,
// so the tile is unpadded [3][64][TJ] and bank conflicts are avoided by an XOR swizzle of the 16-byte
// slots within a row, applied on the SOURCE column (the lane that fills physical slot s of row r
// fetches logical column s ^ swz(r)) and again by the reader.  swz(r) = (r / rows_per_256B) & (slots-1)
// makes the 16 lanes of a ds_read_b128 group hit 16 different slots (conflict-free) and a 32-lane
// ds_read_b64 group 2-way.
template <int TJ> struct DmaTile {
  static constexpr int kSlots = TJ / 4;                 // 16-byte slots per row
  static constexpr int kRowsPerInstr = kWave / kSlots;
  static constexpr int kInstrPerColour = kWave / kRowsPerInstr;
  static constexpr int kInstr = 3 * kInstrPerColour;    // DMA instructions per tile
  static constexpr int kFloats = 3 * kWave * TJ;
  static constexpr int kRowsPerBankRow = 64 / TJ;
  __device__ static __forceinline__ int swz(int row) { return (row / kRowsPerBankRow) & (kSlots - 1); }
};

typedef void __attribute__((address_space(3))) * LdsPtr;

// Buffer resource over one image's env tensor ([3,RC,J] floats): DMA addresses are then
// {SGPR descriptor, SGPR byte offset, one 32-bit VGPR byte offset} -- no 64-bit per-lane pointers --
// and reads past the image return 0 (hardware bounds check).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t env_rsrc(const float* img, int RC, int J) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(img), 0, (int)((size_t)3 * RC * J * 4), 0x00020000);
}

// Issue the DMA for tile (p0.., j0..j0+TJ) of the image behind `rsrc`.  Requires j0 + TJ <= J.
template <int TJ>
__device__ __forceinline__ void tile_dma_issue(float* tile, __amdgpu_buffer_rsrc_t rsrc, int p0, int RC, int J, int j0, int lane) {
  using D = DmaTile<TJ>;
  const int lrow = lane / D::kSlots, slot = lane % D::kSlots;
#pragma unroll
  for (int it = 0; it < D::kInstrPerColour; ++it) {
    const int row = it * D::kRowsPerInstr + lrow;
    const int col4 = slot ^ D::swz(row);
    const int voff = (row * J + col4 * 4) * 4;                           // the lane's byte offset (32-bit)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int soff = (int)((((size_t)c * RC + p0) * J + j0) * 4);      // wave-uniform byte offset
      float* dst = tile + (c * kWave + it * D::kRowsPerInstr) * TJ;      // wave-uniform, + lane*16 B implicitly
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LdsPtr)dst, 16, voff, soff, 0, SGR_DMA_AUX);
    }
  }
}
// The lane's own row of the DMA tile: directions (jjA, jjA+1) and (jjB, jjB+1) of all three colours.
// The reads are inline asm on purpose: for a ds_read the compiler can see, it drains EVERY outstanding
// LDS-DMA first (s_waitcnt vmcnt(0)), which would serialise the prefetch of the next row with the
// consumption of this one.  Ordering is ours: the caller has waited (counted vmcnt) for this tile,
// and nothing uses the outputs before the lgkmcnt(0) below.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned lds_addr(const float* p) {
  return (unsigned)(unsigned long)((const float __attribute__((address_space(3)))*)p);
}
template <int TJ>
__device__ __forceinline__ void tile_dma_read_pairs(const float* tile, int lane, int jjA, int jjB, float (&g)[2][3][2]) {
  using D = DmaTile<TJ>;
  const unsigned rowb = lds_addr(tile) + (unsigned)(lane * TJ * 4);
  const unsigned aA = rowb + (unsigned)((((jjA >> 2) ^ D::swz(lane)) * 4 + (jjA & 3)) * 4);
  const unsigned aB = rowb + (unsigned)((((jjB >> 2) ^ D::swz(lane)) * 4 + (jjB & 3)) * 4);
  f32x2 r[2][3];
  asm volatile("ds_read_b64 %0, %1" : "=v"(r[0][0]) : "v"(aA) : "memory");
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[0][1]) : "v"(aA), "n"(1 * kWave * TJ * 4) : "memory");
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[0][2]) : "v"(aA), "n"(2 * kWave * TJ * 4) : "memory");
  asm volatile("ds_read_b64 %0, %1" : "=v"(r[1][0]) : "v"(aB) : "memory");
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[1][1]) : "v"(aB), "n"(1 * kWave * TJ * 4) : "memory");
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[1][2]) : "v"(aB), "n"(2 * kWave * TJ * 4) : "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int sg = 0; sg < 2; ++sg)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      g[sg][c][0] = r[sg][c].x;
      g[sg][c][1] = r[sg][c].y;
    }
}

// write-back companion of tile_dma_read_pairs (same swizzled addresses), also inline asm
template <int TJ>
__device__ __forceinline__ void tile_dma_write_pairs(float* tile, int lane, int jjA, int jjB, const float (&g)[2][3][2]) {
  using D = DmaTile<TJ>;
  const unsigned rowb = lds_addr(tile) + (unsigned)(lane * TJ * 4);
  const unsigned aA = rowb + (unsigned)((((jjA >> 2) ^ D::swz(lane)) * 4 + (jjA & 3)) * 4);
  const unsigned aB = rowb + (unsigned)((((jjB >> 2) ^ D::swz(lane)) * 4 + (jjB & 3)) * 4);
  f32x2 r[2][3];
#pragma unroll
  for (int sg = 0; sg < 2; ++sg)
#pragma unroll
    for (int c = 0; c < 3; ++c) r[sg][c] = f32x2{g[sg][c][0], g[sg][c][1]};
  asm volatile("ds_write_b64 %0, %1" :: "v"(aA), "v"(r[0][0]) : "memory");
  asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(aA), "v"(r[0][1]), "n"(1 * kWave * TJ * 4) : "memory");
  asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(aA), "v"(r[0][2]), "n"(2 * kWave * TJ * 4) : "memory");
  asm volatile("ds_write_b64 %0, %1" :: "v"(aB), "v"(r[1][0]) : "memory");
  asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(aB), "v"(r[1][1]), "n"(1 * kWave * TJ * 4) : "memory");
  asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(aB), "v"(r[1][2]), "n"(2 * kWave * TJ * 4) : "memory");
}
// DMA of one half of a tile's instructions (multi-wave workgroups split the issue): `part` of `nparts`
template <int TJ>
__device__ __forceinline__ void tile_dma_issue_part(float* tile, __amdgpu_buffer_rsrc_t rsrc, int p0, int RC, int J, int j0,
                                                    int lane, int part, int nparts) {
  using D = DmaTile<TJ>;
  const int lrow = lane / D::kSlots, slot = lane % D::kSlots;
  const int per = D::kInstrPerColour / nparts;
#pragma unroll
  for (int i = 0; i < D::kInstrPerColour / 2; ++i) {      // nparts == 2 in all callers
    const int it = part * per + i;
    const int row = it * D::kRowsPerInstr + lrow;
    const int col4 = slot ^ D::swz(row);
    const int voff = (row * J + col4 * 4) * 4;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int soff = (int)((((size_t)c * RC + p0) * J + j0) * 4);
      float* dst = tile + (c * kWave + it * D::kRowsPerInstr) * TJ;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LdsPtr)dst, 16, voff, soff, 0, SGR_DMA_AUX);
    }
  }
}
// workgroup barrier that does NOT drain outstanding LDS-DMA (unlike __syncthreads): LDS ops only
__device__ __forceinline__ void barrier_lds_only() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt();
template <> __device__ __forceinline__ void wait_vmcnt<0>() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<6>() { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<12>() { asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<24>() { asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); }

// ---- half-wave kernels: one wave = 32 pixels x 2 lobe groups (lanes l and l+32 own the same pixel) ----------
// env-sized rows of a 32-pixel tile by LDS-DMA: [3][32][16] floats (6 KB), 16-byte slots XOR-swizzled by (row >> 2) & 3
constexpr int kPx = 32;
constexpr int kT32Floats = 3 * kPx * 16;
template <int AUX = SGR_DMA_AUX>
__device__ __forceinline__ void tile32_dma_issue(float* tile, __amdgpu_buffer_rsrc_t rsrc, int p0, int RC, int J, int j0, int lane) {
  // Rows lrow and 16 + lrow carry the same swizzle ((row >> 2) & 3 is unchanged by + 16), so the second request's lane offset is the
  // first's + 16 rows: ONE lane-offset register, the 16 rows ride in the wave-uniform offset.  Round 6: as two registers the pair was what
  // the objective's backward kernel spilled -- and reloaded from scratch in front of every row's requests, where the reload's
  // s_waitcnt vmcnt(0) also waited for the three requests just issued (a full memory latency per row and wave: 4-6 % of that kernel).
  const int lrow = lane >> 2, slot = lane & 3;
  const int col4 = slot ^ ((lrow >> 2) & 3);
  const int voff = (lrow * J + col4 * 4) * 4;                            // the lane's byte offset (32-bit)
#pragma unroll
  for (int it = 0; it < 2; ++it) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int soff = (int)((((size_t)c * RC + p0 + it * 16) * J + j0) * 4);      // wave-uniform byte offset
      float* dst = tile + (c * kPx + it * 16) * 16;                      // wave-uniform, + lane*16 B implicitly
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LdsPtr)dst, 16, voff, soff, 0, AUX);
    }
  }
}
// directions (jj, jj+1) of the three colours of pixel row `pl`; jj may differ per lane (inline asm: see tile_dma_read_pairs)
__device__ __forceinline__ void tile32_read_pair(const float* tile, int pl, int jj, float (&g)[3][2]) {
  const unsigned addr = lds_addr(tile) + (unsigned)(pl * 64) + (unsigned)((((jj >> 2) ^ ((pl >> 2) & 3)) * 4 + (jj & 3)) * 4);
  f32x2 r[3];
  asm volatile("ds_read_b64 %0, %1" : "=v"(r[0]) : "v"(addr) : "memory");
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[1]) : "v"(addr), "n"(1 * kPx * 64) : "memory");
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[2]) : "v"(addr), "n"(2 * kPx * 64) : "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int c = 0; c < 3; ++c) { g[c][0] = r[c].x; g[c][1] = r[c].y; }
}
// both half rows' pairs with a single wait (six reads in flight)
__device__ __forceinline__ void tile32_read_two_pairs(const float* tile, int pl, int jjA, int jjB, float (&gA)[3][2], float (&gB)[3][2]) {
  const unsigned base = lds_addr(tile) + (unsigned)(pl * 64);
  const unsigned aA = base + (unsigned)((((jjA >> 2) ^ ((pl >> 2) & 3)) * 4 + (jjA & 3)) * 4);
  const unsigned aB = base + (unsigned)((((jjB >> 2) ^ ((pl >> 2) & 3)) * 4 + (jjB & 3)) * 4);
  f32x2 r[2][3];
  asm volatile("ds_read_b64 %0, %1" : "=v"(r[0][0]) : "v"(aA) : "memory");
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[0][1]) : "v"(aA), "n"(1 * kPx * 64) : "memory");
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[0][2]) : "v"(aA), "n"(2 * kPx * 64) : "memory");
  asm volatile("ds_read_b64 %0, %1" : "=v"(r[1][0]) : "v"(aB) : "memory");
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[1][1]) : "v"(aB), "n"(1 * kPx * 64) : "memory");
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[1][2]) : "v"(aB), "n"(2 * kPx * 64) : "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int c = 0; c < 3; ++c) { gA[c][0] = r[0][c].x; gA[c][1] = r[0][c].y; gB[c][0] = r[1][c].x; gB[c][1] = r[1][c].y; }
}
// D's lanes 32..63 <-> S's lanes 0..31.  Costs as much issue time as a transcendental (8.3-9.3 cycles, tools/ubench4).
__device__ __forceinline__ void swap32(float& d, float& s) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(d), __float_as_uint(s), false, false);
  d = __uint_as_float(r[0]);
  s = __uint_as_float(r[1]);
}

// env tile of a 32-pixel half-wave kernel: [3][32][TD (+4 pad)] floats; the owning lane writes 4 directions at a time,
// the flush reads it back with TD/4 lanes per pixel and streams it out
// TD = directions per pixel row of the tile: 16 (one table row, 64-byte segments) or 32 (two rows, 128-byte segments)
template <int TD> struct T32Out {
  // TD = 16: rows padded by 4 floats (conflict-free as they are).  TD >= 32: unpadded 128-byte rows with the 16-byte slots
  // XOR-swizzled by (row >> 1) & 7 -- a 16-lane group of the owner's ds_write_b128 (rows r .. r+15, one column) and of the
  // flush's ds_read_b128 (two rows x eight columns) both touch 16 different slots of the 64 banks -- so that the two-row tile
  // of the 8x16 grid is 12 KB and three waves per SIMD (twelve per CU) fit the 160 KB of LDS.
  static constexpr bool kSwz = TD >= 32;
  static constexpr int kStride = kSwz ? TD : TD + 4;
  static constexpr int kFloats = 3 * kPx * kStride;
  static constexpr int kLanesPerRow = TD / 4;
  static constexpr int kRowsPerIt = kWave / kLanesPerRow;
  static constexpr int kIts = kPx / kRowsPerIt;
  __device__ static __forceinline__ int col(int row, int c) {      // physical column of logical column c (a multiple of 4) in `row`
    return kSwz ? ((((c >> 2) ^ ((row >> 1) & 7)) << 2) | (c & ~31)) : c;
  }
};
template <int TD>
__device__ __forceinline__ void tile32_write4(float* tile, int pl, int col, const float (&e0)[4], const float (&e1)[4], const float (&e2)[4]) {
  using T = T32Out<TD>;
  const int pc = T::col(pl, col);
  *reinterpret_cast<float4*>(tile + (0 * kPx + pl) * T::kStride + pc) = make_float4(e0[0], e0[1], e0[2], e0[3]);
  *reinterpret_cast<float4*>(tile + (1 * kPx + pl) * T::kStride + pc) = make_float4(e1[0], e1[1], e1[2], e1[3]);
  *reinterpret_cast<float4*>(tile + (2 * kPx + pl) * T::kStride + pc) = make_float4(e2[0], e2[1], e2[2], e2[3]);
}
// flush the first `nd` (16 or TD) directions of every pixel row, starting at direction j0 of the image
template <int TD>
__device__ __forceinline__ void tile32_store_global(const float* tile, float* __restrict__ env_img /* [3,RC,J] of image b */, int p0,
                                                    int RC, int J, int j0, int nd, int lane) {
  using T = T32Out<TD>;
  const int lrow = lane / T::kLanesPerRow, col = (lane % T::kLanesPerRow) * 4;
  const unsigned lane_off = (unsigned)(lrow * J + col);
  const int rows_valid = RC - p0;
  const bool full = rows_valid >= kPx && nd == TD;       // wave-uniform
  // one colour at a time when the tile is wide (TD = 64: 8 float4 per colour), all three at once otherwise
#pragma unroll(TD >= 64 ? 1 : 3)
  for (int c = 0; c < 3; ++c) {
    float4 v[T::kIts];
#pragma unroll
    for (int it = 0; it < T::kIts; ++it) {
      const int row = it * T::kRowsPerIt + lrow;
      v[it] = *reinterpret_cast<const float4*>(tile + (c * kPx + row) * T::kStride + T::col(row, col));
    }
    float* cbase = env_img + ((size_t)c * RC + p0) * J + j0;   // wave-uniform
#pragma unroll
    for (int it = 0; it < T::kIts; ++it) {
      float* dst = cbase + (size_t)(it * T::kRowsPerIt) * J;    // uniform
      f32x4 nv = {v[it].x, v[it].y, v[it].z, v[it].w};
      if (full || (it * T::kRowsPerIt + lrow < rows_valid && col < nd)) stream_store(nv, reinterpret_cast<f32x4*>(dst + lane_off));
    }
  }
}

// ---- pooled BRDF-map fetch ---------------------------------------------------------------
// POOL == 1: maps are already on the env grid.  POOL == 2: 2x2 average (the integer-ratio
// case of F.adaptive_avg_pool2d, models.py:465-469), read as two 8-byte loads per plane.
// `plane` is wave-uniform; `off` is the lane's 32-bit offset of its cell's top-left image pixel
// (r*imW + c for POOL 1, 2r*imW + 2c for POOL 2).
template <int POOL>
__device__ __forceinline__ float fetch_pooled(const float* __restrict__ plane, unsigned off, int imW) {
  if (POOL == 1) {
    return plane[off];
  } else {
    const float2 t = *reinterpret_cast<const float2*>(plane + off);
    const float2 u = *reinterpret_cast<const float2*>(plane + imW + off);
    return (((t.x + t.y) + u.x) + u.y) * 0.25f;
  }
}
template <int POOL>
__device__ __forceinline__ unsigned pooled_offset(int p, int C, int imW) {
  const int r = p / C, c = p - r * C;
  return POOL == 1 ? (unsigned)(r * imW + c) : (unsigned)(2 * r * imW + 2 * c);
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
  const float* rows;      // [ehp,8] (s, c, omega, s^2, 2sc, c^2, 0, 0)   separable form of the table
  const float* cols;      // [ew,8]  (ca, sa, ca^2, 2 ca sa, sa^2, 0, 0, 0)
  const float* view;      // [3,R,C]
  // cotangents
  const float* g_env;     // [bn,3,R,C,J]
  const float* g_diffuse; // [bn,3,R,C]
  const float* g_spec;    // [bn,3,R,C]
  // env reconstruction (fused objective): ground-truth env, pooled env mask inputs, per-image scale
  const float* env_gt;    // [bn,3,R,C,J]
  const float* seg_small; // [bn,R,C], or [bn,2R,2C] when seg_pool2 (pooled 2x2 on the fly, wrapperBRDFLight.py:171)
  int seg_pool2;
  const float* env_ind;   // [bn]
  const float* coef;      // [bn]      LSregress scale (constant in backward)
  const float* mask_in;   // [bn,R,C]  env mask from the forward pass
  const float* den_img;   // [bn]      per-image sums of the env mask (forward pass)
  const float* den_global;// [1]       mask sum all-reduced over ranks, or NULL
  float rec_w3j;          //           reconstruction weight / (3 J)
  float* mask;            // [bn,R,C]
  float* ws;              // per-wave partial sums
  float offset;
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
  int bn, K, R, C, J, Jpad, imH, imW, eh, ew;
  float F0;
  int premap;
};

// pooled object mask of env cell p of image b (the seg_small factor of the env mask, wrapperBRDFLight.py:171-174): read, or the
// 2x2 average of the full-resolution mask in adaptive_avg_pool2d's summation order (bit-identical to pooling first)
__device__ __forceinline__ float seg_small_at(const Args& a, int b, int p) {
  const int RC = a.R * a.C;
  if (a.seg_pool2) {
    const int r = p / a.C, c = p - r * a.C, W = 2 * a.C;
    const float* pl = a.seg_small + (size_t)b * 4 * RC + (size_t)(2 * r) * W + 2 * c;
    const float2 t = *reinterpret_cast<const float2*>(pl), u = *reinterpret_cast<const float2*>(pl + W);
    return (((t.x + t.y) + u.x) + u.y) * 0.25f;
  }
  return (a.seg_small + (size_t)b * RC)[(unsigned)p];
}


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
  const unsigned off = pooled_offset<POOL>(x.p, a.C, a.imW);
  const unsigned up = (unsigned)x.p;
  const size_t plane = (size_t)a.imH * a.imW;
  const float* al = a.albedo + (size_t)x.b * 3 * plane;     // wave-uniform bases
  const float* no = a.normal + (size_t)x.b * 3 * plane;
  const float* ro = a.rough + (size_t)x.b * plane;
  pooled[0] = fetch_pooled<POOL>(al, off, a.imW);
  pooled[1] = fetch_pooled<POOL>(al + plane, off, a.imW);
  pooled[2] = fetch_pooled<POOL>(al + 2 * plane, off, a.imW);
  pooled[3] = fetch_pooled<POOL>(no, off, a.imW);
  pooled[4] = fetch_pooled<POOL>(no + plane, off, a.imW);
  pooled[5] = fetch_pooled<POOL>(no + 2 * plane, off, a.imW);
  pooled[6] = fetch_pooled<POOL>(ro, off, a.imW);
  return make_frame(pooled[3], pooled[4], pooled[5], pooled[6], a.view[up], (a.view + RC)[up], (a.view + 2 * RC)[up]);
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
__device__ __forceinline__ void scatter_pooled(float* __restrict__ plane, unsigned off, int imW, float g) {
  if (POOL == 1) {
    plane[off] = g;
  } else {
    const float q = 0.25f * g;
    *reinterpret_cast<float2*>(plane + off) = make_float2(q, q);
    *reinterpret_cast<float2*>(plane + imW + off) = make_float2(q, q);
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
