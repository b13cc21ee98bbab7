// Forward kernel template: SG -> env image, env -> (diffuse, specular), and the fused pass.
// gfx950 (MI355X) only.  See sgr_common.h for the work decomposition.
#pragma once
#include <stdlib.h>
#include "sgr_common.h"
#include "sgr_launch.h"
#include <string.h>
#include "sgr_pk.inl"

#ifndef SGR_TJ
#define SGR_TJ 32
#endif

namespace sgr {

// KP: lobes held in registers (K <= KP, the rest are zero lobes)
// POOL: BRDF-map pooling ratio (1 or 2)
// FROM_SG: evaluate the SG mixture (else read the env image)
// WRITE_ENV: materialise the env image            DO_RENDER: run the microfacet quadrature
// VEC: J % 4 == 0 -> 16-byte env accesses
template <int KP, int POOL, bool FROM_SG, bool WRITE_ENV, bool DO_RENDER, bool VEC>
__global__ __launch_bounds__(kWave, 2) void fwd_kernel(const Args a) {
  constexpr int TJ = SGR_TJ;
  constexpr bool USE_TILE = WRITE_ENV || !FROM_SG;
  __shared__ __attribute__((aligned(16))) float tile[USE_TILE ? Tile<TJ>::kFloats : 4];

  const Pix x = locate(a);
  const int lane = x.lane, b = x.b, p = x.p;
  const int RC = a.R * a.C;

  // ---- SG lobes -> registers --------------------------------------------------------------
  float ax[KP], ay[KP], az[KP], lam[KP], w0[KP], w1[KP], w2[KP];
  if (FROM_SG) {
    const int K = a.K;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      ax[k] = ay[k] = az[k] = lam[k] = w0[k] = w1[k] = w2[k] = 0.0f;
      if (k < K) {
        const size_t ab = ((size_t)(b * K + k) * 3) * RC;   // wave-uniform plane bases + 32-bit lane offset
        const size_t lb = (size_t)(b * K + k) * RC;
        const unsigned up = (unsigned)p;
        ax[k] = (a.axis + ab)[up];
        ay[k] = (a.axis + ab + RC)[up];
        az[k] = (a.axis + ab + 2 * (size_t)RC)[up];
        float l = (a.lamb + lb)[up];
        float t0 = (a.weight + ab)[up], t1 = (a.weight + ab + RC)[up], t2 = (a.weight + ab + 2 * (size_t)RC)[up];
        if (a.premap == 1) {
          l = premap(l);
          t0 = premap(t0); t1 = premap(t1); t2 = premap(t2);
          if (a.lamb_tan && x.active) (a.lamb_tan + lb)[up] = l;
          if (a.weight_tan && x.active) {
            (a.weight_tan + ab)[up] = t0; (a.weight_tan + ab + RC)[up] = t1; (a.weight_tan + ab + 2 * (size_t)RC)[up] = t2;
          }
        }
        lam[k] = l * kLog2e;   // exp(lam*t) == exp2(lam*log2e*t)
        w0[k] = t0; w1[k] = t1; w2[k] = t2;
      }
    }
  }

  // ---- shading frame ----------------------------------------------------------------------
  Frame f;
  float alb[3] = {0.f, 0.f, 0.f};
  if (DO_RENDER) f = load_frame<POOL>(a, x, alb);

  float d0 = 0.f, d1 = 0.f, d2 = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f;
  const DirTable dirs = as_dir_table(a.dirs);
  const size_t img = (size_t)b * 3 * RC * a.J;

  for (int j0 = 0; j0 < a.Jpad; j0 += TJ) {
    if (!FROM_SG) {
      tile_load_global<TJ, VEC>(tile, a.env_in + img, x.p0, RC, a.J, j0, lane);
      __syncthreads();
    }
#pragma unroll 1
    for (int jj = 0; jj < TJ; jj += 4) {
      float e0[4], e1[4], e2[4];
      if (!FROM_SG) tile_row_read<TJ>(tile, lane, jj, e0, e1, e2);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const f32x4 dir = dirs[j0 + jj + u];   // wave-uniform -> scalar load
        if (FROM_SG) {
          float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
          for (int k = 0; k < KP; ++k) {
            float t = fmaf(az[k], dir.z, -1.0f);
            t = fmaf(ay[k], dir.y, t);
            t = fmaf(ax[k], dir.x, t);
            const float ex = fexp2(lam[k] * t);
            c0 = fmaf(w0[k], ex, c0);
            c1 = fmaf(w1[k], ex, c1);
            c2 = fmaf(w2[k], ex, c2);
          }
          e0[u] = c0; e1[u] = c1; e2[u] = c2;
        }
        if (DO_RENDER) {
          float ndl, sp;
          brdf_dir(f, dir.x, dir.y, dir.z, a.F0, ndl, sp);
          const float wt = ndl * dir.w;
          const float q0 = wt * e0[u], q1 = wt * e1[u], q2 = wt * e2[u];
          d0 += q0; d1 += q1; d2 += q2;
          s0 = fmaf(sp, q0, s0); s1 = fmaf(sp, q1, s1); s2 = fmaf(sp, q2, s2);
        }
        // keep the scheduler from interleaving directions: 12 independent lobe chains per direction
        // are enough ILP, and interleaving 4-8 directions only inflates register pressure.
        __builtin_amdgcn_sched_barrier(0);
      }
      if (FROM_SG && WRITE_ENV) tile_row_write<TJ>(tile, lane, jj, e0, e1, e2);
    }
    if (FROM_SG && WRITE_ENV) {
      __syncthreads();
      tile_store_global<TJ, VEC>(tile, a.env_out + img, x.p0, RC, a.J, j0, lane);
    }
    if (USE_TILE) __syncthreads();
  }

  if (DO_RENDER && x.active) {
    const size_t o = (size_t)b * 3 * RC;
    const unsigned up = (unsigned)p;
    (a.diffuse + o)[up] = (alb[0] * kInvPi) * d0;
    (a.diffuse + o + RC)[up] = (alb[1] * kInvPi) * d1;
    (a.diffuse + o + 2 * (size_t)RC)[up] = (alb[2] * kInvPi) * d2;
    (a.spec + o)[up] = s0;
    (a.spec + o + RC)[up] = s1;
    (a.spec + o + 2 * (size_t)RC)[up] = s2;
  }
}

// ---- launch plumbing -------------------------------------------------------------------------
template <int KP, int POOL, bool FROM_SG, bool WRITE_ENV, bool DO_RENDER>
static int fwd_launch_vec(const Args& a, hipStream_t st) {
  const dim3 grid = wave_grid(a.bn, a.R, a.C), block(kWave);
  if (a.J % 4 == 0)
    hipLaunchKernelGGL((fwd_kernel<KP, POOL, FROM_SG, WRITE_ENV, DO_RENDER, true>), grid, block, 0, st, a);
  else
    hipLaunchKernelGGL((fwd_kernel<KP, POOL, FROM_SG, WRITE_ENV, DO_RENDER, false>), grid, block, 0, st, a);
  return (int)hipGetLastError();
}

template <int POOL, bool FROM_SG, bool WRITE_ENV, bool DO_RENDER>
static int fwd_launch_k(const Args& a, hipStream_t st) {
  if (!FROM_SG) return fwd_launch_vec<1, POOL, FROM_SG, WRITE_ENV, DO_RENDER>(a, st);
  if (a.K <= 4) return fwd_launch_vec<4, POOL, FROM_SG, WRITE_ENV, DO_RENDER>(a, st);
  if (a.K <= 12) return fwd_launch_vec<12, POOL, FROM_SG, WRITE_ENV, DO_RENDER>(a, st);
  if (a.K <= 24) return fwd_launch_vec<24, POOL, FROM_SG, WRITE_ENV, DO_RENDER>(a, st);
  return fwd_launch_vec<32, POOL, FROM_SG, WRITE_ENV, DO_RENDER>(a, st);
}

// ---- fast path (separable table, envWidth 16 or 32, SGNum <= 24): the packed-fp32 kernels of sgr_pk.inl ---------------
// packed forward, one pixel per lane (envWidth 16, SGNum <= KP)
template <int KP, bool WRITE_ENV, bool DO_RENDER, bool HEADS = false>
static int fwd_pk_launch(const Args& a, hipStream_t st) {
  const dim3 grid = wave_grid(a.bn, a.R, a.C), block(kWave);
  if (!DO_RENDER || (a.imH == a.R && a.imW == a.C))
    hipLaunchKernelGGL((fwd_pk_kernel<KP, 1, WRITE_ENV, DO_RENDER, false, HEADS>), grid, block, 0, st, a);
  else
    hipLaunchKernelGGL((fwd_pk_kernel<KP, 2, WRITE_ENV, DO_RENDER, false, HEADS>), grid, block, 0, st, a);
  return (int)hipGetLastError();
}
// packed half-wave forward: 32 pixels x 2 groups of KPW lobes per wave, OCC resident waves per SIMD, RPF table rows per env flush
template <bool WRITE_ENV, bool DO_RENDER, int OCC, int KPW, int EW, int RPF, bool HEADS>
static int fwd_pk_half_launch(const Args& a, hipStream_t st) {
  const dim3 grid((unsigned)(a.bn * ((a.R * a.C + kPx - 1) / kPx))), block(kWave);
  if (!DO_RENDER || (a.imH == a.R && a.imW == a.C))
    hipLaunchKernelGGL((fwd_pk_half_kernel<1, WRITE_ENV, DO_RENDER, OCC, KPW, EW, RPF, HEADS>), grid, block, 0, st, a);
  else
    hipLaunchKernelGGL((fwd_pk_half_kernel<2, WRITE_ENV, DO_RENDER, OCC, KPW, EW, RPF, HEADS>), grid, block, 0, st, a);
  return (int)hipGetLastError();
}
// Which packed kernel runs which shape.  Every choice is a measured one (the A/B records are
// profiles/r02*, r03c_fwd_mode_sweep.txt); the superseded scalar kernels and the knobs that selected them were retired in round 4.
//   SGNum 13..24          half-wave, 12 lobes per half (config 5: 16x32 grid, 24 lobes)
//   envWidth 32, <= 12    half-wave, 6 lobes per half, three waves per SIMD
//   envWidth 16, 7..12    env image written: half-wave with a two-row (128-byte line) env tile, three waves per SIMD --
//                         in the bench loop the forward tracks the WRITE path and whole lines write at ~5 TB/s where 64-byte
//                         segments reach ~3;  render only: one pixel per lane (no duplicated frame set-up)
//   envWidth 16, <= 6     one pixel per lane, six lobe slots
template <bool WRITE_ENV, bool DO_RENDER, bool HEADS>
static int fwd_fast_launch_h(const Args& a, hipStream_t st) {
  if (a.K > 12)
    return a.ew == 16 ? fwd_pk_half_launch<WRITE_ENV, DO_RENDER, 2, 12, 16, 1, HEADS>(a, st)
                      : fwd_pk_half_launch<WRITE_ENV, DO_RENDER, 2, 12, 32, 1, HEADS>(a, st);
  if (a.ew == 32) return fwd_pk_half_launch<WRITE_ENV, DO_RENDER, 3, 6, 32, 1, HEADS>(a, st);
  if constexpr (WRITE_ENV) {
    if (a.K > 6) return fwd_pk_half_launch<WRITE_ENV, DO_RENDER, 3, 6, 16, 2, HEADS>(a, st);
  }
  if constexpr (!HEADS) {
    if (a.K <= 6) return fwd_pk_launch<6, WRITE_ENV, DO_RENDER>(a, st);
  }
  return fwd_pk_launch<12, WRITE_ENV, DO_RENDER, HEADS>(a, st);
}
template <bool WRITE_ENV, bool DO_RENDER>
static int fwd_fast_launch(const Args& a, hipStream_t st) {
  if constexpr (DO_RENDER) {      // premap == 3 (decoder heads as the prologue; fwd_heads_ok holds): the same kernels built with HEADS
    if (a.premap == 3) return fwd_fast_launch_h<WRITE_ENV, DO_RENDER, true>(a, st);
  }
  return fwd_fast_launch_h<WRITE_ENV, DO_RENDER, false>(a, st);
}

// forwardEnv alone (env image given): the packed half-wave kernel of sgr_pk.inl (round 4; round 1's scalar one-pixel-per-lane kernel
// with its 24 KB of LDS per wave measured 110 us against 105 warm, 160 against 158 cold -- profiles/r04c_kbench.txt -- and is gone)
static int render_fast_launch(const Args& a, hipStream_t st) {
  const bool p1 = (a.imH == a.R && a.imW == a.C);
  const dim3 grid((unsigned)(a.bn * ((a.R * a.C + kPx - 1) / kPx))), block(kWave);
  if (a.ew == 16) {
    if (p1) hipLaunchKernelGGL((render_pk_half_kernel<1, 16>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((render_pk_half_kernel<2, 16>), grid, block, 0, st, a);
  } else {
    if (p1) hipLaunchKernelGGL((render_pk_half_kernel<1, 32>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((render_pk_half_kernel<2, 32>), grid, block, 0, st, a);
  }
  return (int)hipGetLastError();
}

template <bool FROM_SG, bool WRITE_ENV, bool DO_RENDER>
static int fwd_launch(const Args& a, hipStream_t st) {
  if (!FROM_SG && DO_RENDER && fast_ok(a) && !sgr_generic_forced()) return render_fast_launch(a, st);
  if (FROM_SG && fast_ok(a) && a.K <= 24 && !sgr_generic_forced()) return fwd_fast_launch<WRITE_ENV, DO_RENDER>(a, st);
  if (!DO_RENDER || (a.imH == a.R && a.imW == a.C)) return fwd_launch_k<1, FROM_SG, WRITE_ENV, DO_RENDER>(a, st);
  return fwd_launch_k<2, FROM_SG, WRITE_ENV, DO_RENDER>(a, st);
}

// premap == 3 (the decoder heads as a prologue) is implemented in the packed kernels' lobe loader (sgr_pk.inl) only: the shapes
// the default dispatch above sends there
static inline bool fwd_heads_ok(const Args& a) {
  return fast_ok(a) && !sgr_generic_forced() && a.K > 6 && a.K <= 24;
}

static inline int check_pool(int R, int C, int imH, int imW, const char* who) {
  const bool ok = (imH == R && imW == C) || (imH == 2 * R && imW == 2 * C);
  SGR_SUPPORTED(ok, who);
  return SGR_OK;
}

static inline void set_dims(Args& a, int bn, int K, int R, int C, int eh, int ew, int imH, int imW) {
  a.bn = bn; a.K = K; a.R = R; a.C = C; a.J = eh * ew; a.Jpad = sgr_dirs_padded(a.J); a.imH = imH; a.imW = imW;
  a.eh = eh; a.ew = ew;
  a.rows = reinterpret_cast<const float*>(a.dirs) + 4 * (size_t)a.Jpad;
  a.cols = a.rows + 8 * (size_t)((eh + 1) / 2 * 2);
}

}  // namespace sgr
