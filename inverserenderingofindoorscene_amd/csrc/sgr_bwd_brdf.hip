// d/d{albedo, normal, rough} of renderingLayer.forwardEnv (autograd of models.py:461-522) on
// gfx950: the env image is either read (un-fused API) or re-evaluated from the SG lobes
// (fused API).  The per-direction / per-frame adjoints are in sgr_math.h.
#include <stdlib.h>
#include "sgr_common.h"
#include "sgr_launch.h"

#ifndef SGR_TJ
#define SGR_TJ 32
#endif

namespace sgr {

template <int KP, int POOL, bool FROM_SG, bool VEC>
__global__ __launch_bounds__(kWave, 1) void brdf_bwd_kernel(const Args a) {
  constexpr int TJ = SGR_TJ;
  __shared__ __attribute__((aligned(16))) float tile[FROM_SG ? 4 : Tile<TJ>::kFloats];

  const Pix x = locate(a);
  const int lane = x.lane, b = x.b, p = x.p;
  const int RC = a.R * a.C;

  float ax[KP], ay[KP], az[KP], lam[KP], w0[KP], w1[KP], w2[KP];
  if (FROM_SG) {
    const int K = a.K;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      ax[k] = ay[k] = az[k] = lam[k] = w0[k] = w1[k] = w2[k] = 0.0f;
      if (k < K) {
        const size_t ab = ((size_t)(b * K + k) * 3) * RC + p;
        ax[k] = a.axis[ab];
        ay[k] = a.axis[ab + RC];
        az[k] = a.axis[ab + 2 * (size_t)RC];
        float l = a.lamb[(size_t)(b * K + k) * RC + p];
        float t0 = a.weight[ab], t1 = a.weight[ab + RC], t2 = a.weight[ab + 2 * (size_t)RC];
        if (a.premap == 1) {
          l = premap(l);
          t0 = premap(t0); t1 = premap(t1); t2 = premap(t2);
        }
        lam[k] = l * kLog2e; w0[k] = t0; w1[k] = t1; w2[k] = t2;
      }
    }
  }

  float pooled[7];
  const Frame f = load_frame_pooled<POOL>(a, x, pooled);
  const size_t o = (size_t)b * 3 * RC + p;
  const float gD0 = a.g_diffuse[o], gD1 = a.g_diffuse[o + RC], gD2 = a.g_diffuse[o + 2 * (size_t)RC];
  const float gs0 = a.g_spec[o], gs1 = a.g_spec[o + RC], gs2 = a.g_spec[o + 2 * (size_t)RC];
  const float gd0 = gD0 * (pooled[0] * kInvPi), gd1 = gD1 * (pooled[1] * kInvPi), gd2 = gD2 * (pooled[2] * kInvPi);

  FrameGrad g;
  frame_grad_zero(g);
  float ds0 = 0.f, ds1 = 0.f, ds2 = 0.f;
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
        const f32x4 dir = dirs[j0 + jj + u];
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
        const float Ed = dir.w * (gd0 * e0[u] + gd1 * e1[u] + gd2 * e2[u]);
        const float Es = dir.w * (gs0 * e0[u] + gs1 * e1[u] + gs2 * e2[u]);
        const float ndl = brdf_dir_bwd(f, dir.x, dir.y, dir.z, a.F0, Ed, Es, g);
        const float wt = ndl * dir.w;
        ds0 = fmaf(wt, e0[u], ds0); ds1 = fmaf(wt, e1[u], ds1); ds2 = fmaf(wt, e2[u], ds2);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (!FROM_SG) __syncthreads();
  }

  float gpn[3], gprho;
  frame_bwd(pooled[3], pooled[4], pooled[5], pooled[6], f, g, gpn, gprho);
  if (x.active) {
    const unsigned off = pooled_offset<POOL>(p, a.C, a.imW);
    const size_t plane = (size_t)a.imH * a.imW;
    float* ga = a.g_albedo + (size_t)b * 3 * plane;
    float* gn = a.g_normal + (size_t)b * 3 * plane;
    float* gr = a.g_rough + (size_t)b * plane;
    scatter_pooled<POOL>(ga, off, a.imW, gD0 * kInvPi * ds0);
    scatter_pooled<POOL>(ga + plane, off, a.imW, gD1 * kInvPi * ds1);
    scatter_pooled<POOL>(ga + 2 * plane, off, a.imW, gD2 * kInvPi * ds2);
    scatter_pooled<POOL>(gn, off, a.imW, gpn[0]);
    scatter_pooled<POOL>(gn + plane, off, a.imW, gpn[1]);
    scatter_pooled<POOL>(gn + 2 * plane, off, a.imW, gpn[2]);
    scatter_pooled<POOL>(gr, off, a.imW, gprho);
  }
}

// env given, envWidth 16: the env rows arrive by double-buffered LDS-DMA (as in render_fast_kernel) instead of being
// staged through registers behind two workgroup barriers per tile; the arithmetic is the generic kernel's.
template <int POOL>
__global__ __launch_bounds__(kWave, 2) void brdf_bwd_dma_kernel(const Args a) {
  constexpr int EW = 16, HALF = 8;
  using D = DmaTile<EW>;
  __shared__ __attribute__((aligned(16))) float tile[2 * D::kFloats];

  const Pix x = locate(a);
  const int lane = x.lane, b = x.b, p = x.p;
  const int RC = a.R * a.C;
  float pooled[7];
  const Frame f = load_frame_pooled<POOL>(a, x, pooled);
  const size_t o = (size_t)b * 3 * RC + p;
  const float gD0 = a.g_diffuse[o], gD1 = a.g_diffuse[o + RC], gD2 = a.g_diffuse[o + 2 * (size_t)RC];
  const float gs0 = a.g_spec[o], gs1 = a.g_spec[o + RC], gs2 = a.g_spec[o + 2 * (size_t)RC];
  const float gd0 = gD0 * (pooled[0] * kInvPi), gd1 = gD1 * (pooled[1] * kInvPi), gd2 = gD2 * (pooled[2] * kInvPi);

  FrameGrad g;
  frame_grad_zero(g);
  float ds0 = 0.f, ds1 = 0.f, ds2 = 0.f;
  const DirTable dirs = as_dir_table(a.dirs);
  __amdgpu_buffer_rsrc_t eimg = env_rsrc(a.env_in + (size_t)b * 3 * RC * a.J, RC, a.J);
  const int eh = a.J / EW;

  tile_dma_issue<EW>(tile, eimg, x.p0, RC, a.J, 0, lane);
  for (int e = 0; e < eh; ++e) {
    const float* cur = tile + (e & 1) * D::kFloats;
    if (e + 1 < eh) {
      tile_dma_issue<EW>(tile + ((e + 1) & 1) * D::kFloats, eimg, x.p0, RC, a.J, (e + 1) * EW, lane);
      wait_vmcnt<D::kInstr>();
    } else {
      wait_vmcnt<0>();
    }
#pragma unroll 1
    for (int ap = 0; ap < HALF / 2; ++ap) {
      float ev[2][3][2];
      tile_dma_read_pairs<EW>(cur, lane, ap * 2, HALF + ap * 2, ev);
#pragma unroll
      for (int sg = 0; sg < 2; ++sg)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const f32x4 dir = dirs[e * EW + sg * HALF + ap * 2 + i];
          const float e0 = ev[sg][0][i], e1 = ev[sg][1][i], e2 = ev[sg][2][i];
          const float Ed = dir.w * (gd0 * e0 + gd1 * e1 + gd2 * e2);
          const float Es = dir.w * (gs0 * e0 + gs1 * e1 + gs2 * e2);
          const float ndl = brdf_dir_bwd(f, dir.x, dir.y, dir.z, a.F0, Ed, Es, g);
          const float wt = ndl * dir.w;
          ds0 = fmaf(wt, e0, ds0); ds1 = fmaf(wt, e1, ds1); ds2 = fmaf(wt, e2, ds2);
          __builtin_amdgcn_sched_barrier(0);
        }
    }
  }

  float gpn[3], gprho;
  frame_bwd(pooled[3], pooled[4], pooled[5], pooled[6], f, g, gpn, gprho);
  if (x.active) {
    const unsigned off = pooled_offset<POOL>(p, a.C, a.imW);
    const size_t plane = (size_t)a.imH * a.imW;
    float* ga = a.g_albedo + (size_t)b * 3 * plane;
    float* gn = a.g_normal + (size_t)b * 3 * plane;
    float* gr = a.g_rough + (size_t)b * plane;
    scatter_pooled<POOL>(ga, off, a.imW, gD0 * kInvPi * ds0);
    scatter_pooled<POOL>(ga + plane, off, a.imW, gD1 * kInvPi * ds1);
    scatter_pooled<POOL>(ga + 2 * plane, off, a.imW, gD2 * kInvPi * ds2);
    scatter_pooled<POOL>(gn, off, a.imW, gpn[0]);
    scatter_pooled<POOL>(gn + plane, off, a.imW, gpn[1]);
    scatter_pooled<POOL>(gn + 2 * plane, off, a.imW, gpn[2]);
    scatter_pooled<POOL>(gr, off, a.imW, gprho);
  }
}

template <int KP, int POOL, bool FROM_SG>
static int brdf_launch_vec(const Args& a, hipStream_t st) {
  const dim3 grid = wave_grid(a.bn, a.R, a.C), block(kWave);
  if (a.J % 4 == 0)
    hipLaunchKernelGGL((brdf_bwd_kernel<KP, POOL, FROM_SG, true>), grid, block, 0, st, a);
  else
    hipLaunchKernelGGL((brdf_bwd_kernel<KP, POOL, FROM_SG, false>), grid, block, 0, st, a);
  return (int)hipGetLastError();
}

template <int POOL>
static int brdf_launch(const Args& a, hipStream_t st) {
  if (a.K == 0 && a.ew == 16 && 3LL * a.R * a.C * a.J * 4 < (1LL << 31) && !sgr_generic_forced()) {
    hipLaunchKernelGGL((brdf_bwd_dma_kernel<POOL>), wave_grid(a.bn, a.R, a.C), dim3(kWave), 0, st, a);
    return (int)hipGetLastError();
  }
  if (a.K == 0) return brdf_launch_vec<1, POOL, false>(a, st);
  if (a.K <= 4) return brdf_launch_vec<4, POOL, true>(a, st);
  if (a.K <= 12) return brdf_launch_vec<12, POOL, true>(a, st);
  if (a.K <= 24) return brdf_launch_vec<24, POOL, true>(a, st);
  return brdf_launch_vec<32, POOL, true>(a, st);
}

}  // namespace sgr

using namespace sgr;

extern "C" int sgr_render_bwd_brdf(const float* g_diffuse, const float* g_spec, const float* albedo,
                                   const float* normal, const float* rough, const float* env, const float* axis,
                                   const float* lamb, const float* weight, const float* dirs, const float* view,
                                   float* g_albedo, float* g_normal, float* g_rough, int bn, int K, int R, int C,
                                   int eh, int ew, int imH, int imW, float F0, int premap, void* stream) {
  SGR_REQUIRE(g_diffuse && g_spec && albedo && normal && rough && dirs && view && g_albedo && g_normal && g_rough,
              "sgr_render_bwd_brdf: NULL tensor");
  SGR_REQUIRE(env || (axis && lamb && weight && K > 0), "sgr_render_bwd_brdf: need either env or the SG parameters");
  SGR_REQUIRE(bn > 0 && R > 0 && C > 0 && eh > 0 && ew > 0, "sgr_render_bwd_brdf: non-positive size");
  SGR_SUPPORTED(K <= SGR_MAX_LOBES, "sgr_render_bwd_brdf: SGNum > 32 is not supported");
  const bool ok = (imH == R && imW == C) || (imH == 2 * R && imW == 2 * C);
  SGR_SUPPORTED(ok, "sgr_render_bwd_brdf: BRDF-map / env-grid ratio must be 1 or 2 (pool first)");
  Args a{};
  a.g_diffuse = g_diffuse; a.g_spec = g_spec; a.albedo = albedo; a.normal = normal; a.rough = rough;
  a.env_in = env; a.axis = axis; a.lamb = lamb; a.weight = weight;
  a.dirs = reinterpret_cast<const float4*>(dirs); a.view = view;
  a.g_albedo = g_albedo; a.g_normal = g_normal; a.g_rough = g_rough;
  a.bn = bn; a.K = env ? 0 : K; a.R = R; a.C = C; a.J = eh * ew; a.Jpad = sgr_dirs_padded(a.J); a.imH = imH; a.imW = imW;
  a.F0 = F0; a.premap = premap == 1 ? 1 : 0; a.eh = eh; a.ew = ew;      // 2 = post-tan SG inputs: nothing to pre-map, no SG chain rule here
  const hipStream_t st = (hipStream_t)stream;
  return sgr_check(imH == R ? brdf_launch<1>(a, st) : brdf_launch<2>(a, st), "sgr_render_bwd_brdf");
}
