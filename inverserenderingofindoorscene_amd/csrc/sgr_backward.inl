// Backward kernel templates (hand-derived adjoints).
// gfx950 (MI355X) only.  Same lanes<->pixels decomposition as the forward pass.
//
//   sg_bwd_kernel      d/d{axis, lamb, weight} of the SG mixture, given the per-direction
//                      cotangent  g[c,j] = gEnv[c,j]  (+)  omega_j ndl_j (gD_c A_c/pi + gS_c spec_j)
//                      -- the second term is the adjoint of the quadrature (models.py:511-520)
//                      recomputed on the fly, so the fused backward never materialises dL/dEnv.
//   render_genv_kernel dL/dEnv of forwardEnv alone (the un-fused drop-in path).
//
// With E_kj = exp(lam_k (a_k.l_j - 1)),  s_kj = sum_c g_cj w_kc,  T_kj = s_kj E_kj:
//   dL/dw_kc  = sum_j g_cj E_kj          dL/dlam_k = sum_j T_kj (a_k.l_j - 1)
//   dL/da_k   = lam_k sum_j T_kj l_j     and through the pre-map y = tan(pi/2 * 0.999 x):
//   dL/dx     = dL/dy * 0.999 * pi/2 * (1 + y^2)                         (models.py:396-400)
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

// KP lobes per register group (K > KP loops over groups, re-reading the cotangent tile).
template <int KP, int POOL, bool HAS_GENV, bool HAS_RENDER, bool VEC>
__global__ __launch_bounds__(kWave, 1) void sg_bwd_kernel(const Args a) {
  constexpr int TJ = SGR_TJ;
  __shared__ __attribute__((aligned(16))) float tile[HAS_GENV ? Tile<TJ>::kFloats : 4];

  const Pix x = locate(a);
  const int lane = x.lane, b = x.b, p = x.p;
  const int RC = a.R * a.C;
  const int K = a.K;

  Frame f;
  float gd0 = 0.f, gd1 = 0.f, gd2 = 0.f, gs0 = 0.f, gs1 = 0.f, gs2 = 0.f;
  if (HAS_RENDER) {
    float alb[3];
    f = load_frame<POOL>(a, x, alb);
    const size_t o = (size_t)b * 3 * RC;
    const unsigned up = (unsigned)p;
    gd0 = (a.g_diffuse + o)[up] * (alb[0] * kInvPi);
    gd1 = (a.g_diffuse + o + RC)[up] * (alb[1] * kInvPi);
    gd2 = (a.g_diffuse + o + 2 * (size_t)RC)[up] * (alb[2] * kInvPi);
    gs0 = (a.g_spec + o)[up];
    gs1 = (a.g_spec + o + RC)[up];
    gs2 = (a.g_spec + o + 2 * (size_t)RC)[up];
  }
  const DirTable dirs = as_dir_table(a.dirs);
  const size_t img = (size_t)b * 3 * RC * a.J;

  for (int kg = 0; kg < K; kg += KP) {
    // ---- this group's lobes -> registers ---------------------------------------------------
    float ax[KP], ay[KP], az[KP], lam[KP], w0[KP], w1[KP], w2[KP];
    float gax[KP], gay[KP], gaz[KP], glam[KP], gw0[KP], gw1[KP], gw2[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      ax[k] = ay[k] = az[k] = lam[k] = w0[k] = w1[k] = w2[k] = 0.0f;
      gax[k] = gay[k] = gaz[k] = glam[k] = gw0[k] = gw1[k] = gw2[k] = 0.0f;
      if (kg + k < K) {
        const size_t ab = ((size_t)(b * K + kg + k) * 3) * RC;
        const unsigned up = (unsigned)p;
        ax[k] = (a.axis + ab)[up];
        ay[k] = (a.axis + ab + RC)[up];
        az[k] = (a.axis + ab + 2 * (size_t)RC)[up];
        float l = (a.lamb + (size_t)(b * K + kg + k) * RC)[up];
        float t0 = (a.weight + ab)[up], t1 = (a.weight + ab + RC)[up], t2 = (a.weight + ab + 2 * (size_t)RC)[up];
        if (a.premap == 1) {
          l = premap(l);
          t0 = premap(t0); t1 = premap(t1); t2 = premap(t2);
        }
        lam[k] = l; w0[k] = t0; w1[k] = t1; w2[k] = t2;
      }
    }

    for (int j0 = 0; j0 < a.Jpad; j0 += TJ) {
      if (HAS_GENV) {
        tile_load_global<TJ, VEC>(tile, a.g_env + img, x.p0, RC, a.J, j0, lane);
        __syncthreads();
      }
#pragma unroll 1
      for (int jj = 0; jj < TJ; jj += 4) {
        float g0[4] = {0.f, 0.f, 0.f, 0.f}, g1[4] = {0.f, 0.f, 0.f, 0.f}, g2[4] = {0.f, 0.f, 0.f, 0.f};
        if (HAS_GENV) tile_row_read<TJ>(tile, lane, jj, g0, g1, g2);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const f32x4 dir = dirs[j0 + jj + u];
          float c0 = g0[u], c1 = g1[u], c2 = g2[u];
          if (HAS_RENDER) {
            float ndl, sp;
            brdf_dir(f, dir.x, dir.y, dir.z, a.F0, ndl, sp);
            const float wt = ndl * dir.w;
            c0 = fmaf(wt, fmaf(gs0, sp, gd0), c0);
            c1 = fmaf(wt, fmaf(gs1, sp, gd1), c1);
            c2 = fmaf(wt, fmaf(gs2, sp, gd2), c2);
          }
#pragma unroll
          for (int k = 0; k < KP; ++k) {
            float t = fmaf(az[k], dir.z, -1.0f);
            t = fmaf(ay[k], dir.y, t);
            t = fmaf(ax[k], dir.x, t);
            const float ex = fexp2((lam[k] * kLog2e) * t);
            gw0[k] = fmaf(c0, ex, gw0[k]);
            gw1[k] = fmaf(c1, ex, gw1[k]);
            gw2[k] = fmaf(c2, ex, gw2[k]);
            const float s = fmaf(c2, w2[k], fmaf(c1, w1[k], c0 * w0[k]));
            const float T = s * ex;
            glam[k] = fmaf(T, t, glam[k]);
            gax[k] = fmaf(T, dir.x, gax[k]);
            gay[k] = fmaf(T, dir.y, gay[k]);
            gaz[k] = fmaf(T, dir.z, gaz[k]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (HAS_GENV) __syncthreads();
    }

    // ---- write this group's gradients --------------------------------------------------------
    if (x.active) {
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        if (kg + k < K) {
          const size_t ab = ((size_t)(b * K + kg + k) * 3) * RC;
          const size_t lb = (size_t)(b * K + kg + k) * RC;
          const unsigned up = (unsigned)p;
          (a.g_axis + ab)[up] = lam[k] * gax[k];
          (a.g_axis + ab + RC)[up] = lam[k] * gay[k];
          (a.g_axis + ab + 2 * (size_t)RC)[up] = lam[k] * gaz[k];
          float gl = glam[k], q0 = gw0[k], q1 = gw1[k], q2 = gw2[k];
          if (a.premap) {
            gl *= premap_grad(lam[k]);
            q0 *= premap_grad(w0[k]); q1 *= premap_grad(w1[k]); q2 *= premap_grad(w2[k]);
          }
          (a.g_lamb + lb)[up] = gl;
          (a.g_weight + ab)[up] = q0;
          (a.g_weight + ab + RC)[up] = q1;
          (a.g_weight + ab + 2 * (size_t)RC)[up] = q2;
        }
      }
    }
  }
}

// dL/dEnv[c,j] = omega_j ndl_j (gD_c A_c/pi + gS_c spec_j)      (adjoint of models.py:511-520)
template <int POOL, bool VEC>
__global__ __launch_bounds__(kWave, 2) void render_genv_kernel(const Args a) {
  constexpr int TJ = SGR_TJ;
  __shared__ __attribute__((aligned(16))) float tile[Tile<TJ>::kFloats];
  const Pix x = locate(a);
  const int lane = x.lane, b = x.b, p = x.p;
  const int RC = a.R * a.C;
  float alb[3];
  const Frame f = load_frame<POOL>(a, x, alb);
  const size_t o = (size_t)b * 3 * RC;
  const unsigned up = (unsigned)p;
  const float gd0 = (a.g_diffuse + o)[up] * (alb[0] * kInvPi);
  const float gd1 = (a.g_diffuse + o + RC)[up] * (alb[1] * kInvPi);
  const float gd2 = (a.g_diffuse + o + 2 * (size_t)RC)[up] * (alb[2] * kInvPi);
  const float gs0 = (a.g_spec + o)[up], gs1 = (a.g_spec + o + RC)[up], gs2 = (a.g_spec + o + 2 * (size_t)RC)[up];
  const DirTable dirs = as_dir_table(a.dirs);
  const size_t img = (size_t)b * 3 * RC * a.J;
  for (int j0 = 0; j0 < a.Jpad; j0 += TJ) {
#pragma unroll 2
    for (int jj = 0; jj < TJ; jj += 4) {
      float g0[4], g1[4], g2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const f32x4 dir = dirs[j0 + jj + u];
        float ndl, sp;
        brdf_dir(f, dir.x, dir.y, dir.z, a.F0, ndl, sp);
        const float wt = ndl * dir.w;
        g0[u] = wt * fmaf(gs0, sp, gd0);
        g1[u] = wt * fmaf(gs1, sp, gd1);
        g2[u] = wt * fmaf(gs2, sp, gd2);
      }
      tile_row_write<TJ>(tile, lane, jj, g0, g1, g2);
    }
    __syncthreads();
    tile_store_global<TJ, VEC>(tile, a.g_env_out + img, x.p0, RC, a.J, j0, lane);
    __syncthreads();
  }
}

// ---- launch plumbing -------------------------------------------------------------------------
template <int KP, int POOL, bool HAS_GENV, bool HAS_RENDER>
static int sgbwd_launch_vec(const Args& a, hipStream_t st) {
  const dim3 grid = wave_grid(a.bn, a.R, a.C), block(kWave);
  if (a.J % 4 == 0)
    hipLaunchKernelGGL((sg_bwd_kernel<KP, POOL, HAS_GENV, HAS_RENDER, true>), grid, block, 0, st, a);
  else
    hipLaunchKernelGGL((sg_bwd_kernel<KP, POOL, HAS_GENV, HAS_RENDER, false>), grid, block, 0, st, a);
  return (int)hipGetLastError();
}

template <int POOL, bool HAS_GENV, bool HAS_RENDER>
static int sgbwd_launch_k(const Args& a, hipStream_t st) {
  // register groups: the smallest KP that covers K in the fewest passes
  if (a.K <= 4) return sgbwd_launch_vec<4, POOL, HAS_GENV, HAS_RENDER>(a, st);
  if (a.K <= 8 || (a.K > 12 && a.K <= 16)) return sgbwd_launch_vec<8, POOL, HAS_GENV, HAS_RENDER>(a, st);
  return sgbwd_launch_vec<12, POOL, HAS_GENV, HAS_RENDER>(a, st);
}

// packed-fp32 half-wave backward (envWidth 16 or 32; one workgroup per 32 pixels and group of 12 lobes), sgr_pk.inl
template <bool HAS_GENV, bool HAS_RENDER, int EW, bool HEADS = false>
static int sgbwd_pk_launch_ew(const Args& a, hipStream_t st) {
  const int ng = (a.K + 11) / 12;
  const unsigned tiles = (unsigned)(a.bn * ((a.R * a.C + kPx - 1) / kPx));
  const dim3 grid(ng == 1 ? tiles : ((tiles + 7) / 8) * 8 * (unsigned)ng), block(kWave);
  if (!HAS_RENDER || (a.imH == a.R && a.imW == a.C))
    hipLaunchKernelGGL((sg_bwd_pk_kernel<1, HAS_GENV, HAS_RENDER, EW, HEADS>), grid, block, 0, st, a);
  else
    hipLaunchKernelGGL((sg_bwd_pk_kernel<2, HAS_GENV, HAS_RENDER, EW, HEADS>), grid, block, 0, st, a);
  return (int)hipGetLastError();
}
template <bool HAS_GENV, bool HAS_RENDER>
static int sgbwd_pk_launch(const Args& a, hipStream_t st) {
  if constexpr (HAS_RENDER) {      // premap == 3: the same kernels built with the decoder heads as prologue / epilogue (bwd_heads_ok holds)
    if (a.premap == 3)
      return a.ew == 16 ? sgbwd_pk_launch_ew<HAS_GENV, HAS_RENDER, 16, true>(a, st) : sgbwd_pk_launch_ew<HAS_GENV, HAS_RENDER, 32, true>(a, st);
  }
  return a.ew == 16 ? sgbwd_pk_launch_ew<HAS_GENV, HAS_RENDER, 16>(a, st) : sgbwd_pk_launch_ew<HAS_GENV, HAS_RENDER, 32>(a, st);
}
template <bool HAS_GENV, bool HAS_RENDER>
static int sgbwd_launch(const Args& a, hipStream_t st) {
  // the reference's direction grids (16- and 32-wide): the packed half-wave kernel for every SGNum -- six lobes per half-wave,
  // one workgroup per 32 pixels and group of 12 lobes (SGNum <= 6 leaves the upper half's lobe slots empty)
  if (fast_ok(a) && !sgr_generic_forced()) return sgbwd_pk_launch<HAS_GENV, HAS_RENDER>(a, st);
  if (!HAS_RENDER || (a.imH == a.R && a.imW == a.C)) return sgbwd_launch_k<1, HAS_GENV, HAS_RENDER>(a, st);
  return sgbwd_launch_k<2, HAS_GENV, HAS_RENDER>(a, st);
}

// premap == 3: see fwd_heads_ok (sgr_forward.inl)
static inline bool bwd_heads_ok(const Args& a) {
  return fast_ok(a) && !sgr_generic_forced() && a.K > 6 && a.K <= 24;
}

static inline int check_pool_b(int R, int C, int imH, int imW, const char* who) {
  const bool ok = (imH == R && imW == C) || (imH == 2 * R && imW == 2 * C);
  SGR_SUPPORTED(ok, who);
  return SGR_OK;
}

static inline void set_dims_b(Args& a, int bn, int K, int R, int C, int eh, int ew, int imH, int imW) {
  a.bn = bn; a.K = K; a.R = R; a.C = C; a.J = eh * ew; a.Jpad = sgr_dirs_padded(a.J); a.imH = imH; a.imW = imW;
  a.eh = eh; a.ew = ew;
  a.rows = reinterpret_cast<const float*>(a.dirs) + 4 * (size_t)a.Jpad;
  a.cols = a.rows + 8 * (size_t)((eh + 1) / 2 * 2);
}

}  // namespace sgr
