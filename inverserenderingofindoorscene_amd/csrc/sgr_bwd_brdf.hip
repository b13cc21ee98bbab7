// d/d{albedo, normal, rough} of renderingLayer.forwardEnv (autograd of models.py:461-522) on
// gfx950: the env image is either read (un-fused API) or re-evaluated from the SG lobes
// (fused API).  The per-direction / per-frame adjoints are in sgr_math.h.
#include <stdlib.h>
#include <string.h>
#include "sgr_pk.inl"
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


// ============================== round 3: the same adjoint, two directions per instruction ========================================
// brdf_dir_bwd (sgr_math.h) restated over the azimuth pair (a, a+1) of one table row and sign, the packing of sgr_pk.inl: every
// multiply-add of the world-space adjoint is one half of a v_pk_fma_f32; what stays per element are the clamps (v_med3), their
// gradient gates, the transcendentals and the Newton steps' seeds.  Directions come from the separable table
// (l = (ss ca_a, ss sa_a, c_e): the pair's (ca, sa) are SGPR pairs), the env rows by the double-buffered LDS-DMA of
// render_pk_half_kernel, the pairs read as ds_read_b64.  ~150 packed + ~40 scalar instructions per PAIR against ~170 scalar per
// direction.  World-space throughout, so degenerate frames need no separate path.
struct FrameGradPk {
  f32x2 gN[3], gcx[3], gcy[3], galpha2, gk, gndv;
};
__device__ __forceinline__ f32x2 sel2(bool c0, bool c1, f32x2 a, f32x2 b) { return f32x2{c0 ? a.x : b.x, c1 ? a.y : b.y}; }
__device__ __forceinline__ f32x2 clamp01_2(f32x2 x) { return f32x2{clamp01(x.x), clamp01(x.y)}; }
__device__ __forceinline__ f32x2 frsq_nr2(f32x2 x) {
  const f32x2 y = {__builtin_amdgcn_rsqf(x.x), __builtin_amdgcn_rsqf(x.y)};
  return pfma(pfma(splat2(-0.5f) * x * y, y, splat2(0.5f)), y, y);
}
__device__ __forceinline__ f32x2 frcp_nr2(f32x2 x) {
  const f32x2 y = {__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)};
  return pfma(pfma(-x, y, splat2(1.0f)), y, y);
}
// directions (lx, ly, lz): lx, ly pairs, lz wave-uniform.  Returns ndl (pair) and accumulates into g.
__device__ __forceinline__ f32x2 brdf_pair_bwd(const Frame& f, f32x2 lx, f32x2 ly, float lzs, float F0, f32x2 Ed, f32x2 Es, FrameGradPk& g) {
  const f32x2 lz = splat2(lzs);
  const f32x2 wx = pfma(lz, splat2(f.nx), pfma(ly, splat2(f.cyx), lx * splat2(f.cxx)));
  const f32x2 wy = pfma(lz, splat2(f.ny), pfma(ly, splat2(f.cyy), lx * splat2(f.cxy)));
  const f32x2 wz = pfma(lz, splat2(f.nz), pfma(ly, splat2(f.cyz), lx * splat2(f.cxz)));
  const f32x2 hsx = (splat2(f.vx) + wx) * splat2(0.5f), hsy = (splat2(f.vy) + wy) * splat2(0.5f), hsz = (splat2(f.vz) + wz) * splat2(0.5f);
  const f32x2 hh = pfma(hsz, hsz, pfma(hsy, hsy, hsx * hsx));
  const f32x2 hinv = frsq_nr2(f32x2{fmaxf(hh.x, 1e-6f), fmaxf(hh.y, 1e-6f)});
  const f32x2 hx = hsx * hinv, hy = hsy * hinv, hz = hsz * hinv;
  const f32x2 vdh = pfma(splat2(f.vz), hz, pfma(splat2(f.vy), hy, splat2(f.vx) * hx));
  const f32x2 pa = pfma(splat2(-5.55472f), vdh, splat2(-6.98316f)) * vdh;
  const f32x2 pw = {fexp2(pa.x), fexp2(pa.y)};
  const f32x2 fres = pfma(splat2(1.0f - F0), pw, splat2(F0));
  const f32x2 ndh_raw = pfma(splat2(f.nz), hz, pfma(splat2(f.ny), hy, splat2(f.nx) * hx));
  const f32x2 ndl_raw = pfma(splat2(f.nz), wz, pfma(splat2(f.ny), wy, splat2(f.nx) * wx));
  const f32x2 ndh = clamp01_2(ndh_raw), ndl = clamp01_2(ndl_raw);
  const float omk = 1.0f - f.k, am1 = f.alpha2 - 1.0f;
  // nom0 without the cancellation of 1 + ndh^2 (alpha^2 - 1) where the frame is regular (see brdf_dir_bwd)
  f32x2 nom0 = pfma(ndh * ndh, splat2(am1), splat2(1.0f));
  {
    const float nn = f.nx * f.nx + f.ny * f.ny + f.nz * f.nz;
    const f32x2 nh = pfma(splat2(f.nz), hsz, pfma(splat2(f.ny), hsy, splat2(f.nx) * hsx));
    const f32x2 tx = pfma(-nh, splat2(f.nx), hsx), ty = pfma(-nh, splat2(f.ny), hsy), tz = pfma(-nh, splat2(f.nz), hsz);
    const f32x2 tt = pfma(tz, tz, pfma(ty, ty, tx * tx));
    const f32x2 alt = pfma(splat2(f.alpha2), nh * nh, tt) * (hinv * hinv);
    const bool nreg = fabsf(nn - 1.0f) < 4e-7f;
    const bool r0 = nreg && hh.x >= 1e-6f && ndh_raw.x >= 0.0f && ndh_raw.x <= 1.0f;
    const bool r1 = nreg && hh.y >= 1e-6f && ndh_raw.y >= 0.0f && ndh_raw.y <= 1.0f;
    nom0 = sel2(r0, r1, alt, nom0);
  }
  const f32x2 nom2 = pfma(ndl, splat2(omk), splat2(f.k));
  const f32x2 n00 = nom0 * nom0;
  const f32x2 nomr = (splat2(kFourPi * f.nom1) * n00) * nom2;
  const f32x2 nom = {clampf(nomr.x, 1e-6f, kFourPi), clampf(nomr.y, 1e-6f, kFourPi)};
  const f32x2 rn = frcp_nr2(nom);
  const f32x2 frn = fres * rn;
  const f32x2 spec = splat2(f.alpha2) * frn;

  f32x2 gndl = pfma(spec, Es, Ed);                 // direct
  const f32x2 gsp = ndl * Es;
  g.galpha2 = pfma(gsp, frn, g.galpha2);
  const f32x2 gfres = (gsp * splat2(f.alpha2)) * rn;
  const f32x2 gnom_in = -(gsp * spec) * rn;
  const f32x2 zero = splat2(0.0f);
  const f32x2 gnom = sel2(nomr.x >= 1e-6f && nomr.x <= kFourPi, nomr.y >= 1e-6f && nomr.y <= kFourPi, gnom_in, zero);
  const f32x2 c4 = splat2(kFourPi);
  const f32x2 gnom0 = ((gnom * splat2(2.0f * kFourPi * f.nom1)) * nom0) * nom2;
  const f32x2 gnom1 = ((gnom * c4) * n00) * nom2;
  const f32x2 gnom2 = (gnom * splat2(kFourPi * f.nom1)) * n00;
  f32x2 gndh = ((gnom0 * splat2(2.0f * am1)) * ndh);
  g.galpha2 = pfma(gnom0, ndh * ndh, g.galpha2);
  g.gndv = pfma(gnom1, splat2(omk), g.gndv);
  g.gk = pfma(gnom1, splat2(1.0f - f.ndv), pfma(gnom2, splat2(1.0f) - ndl, g.gk));
  gndl = pfma(gnom2, splat2(omk), gndl);
  const f32x2 gvdh = ((gfres * splat2((1.0f - F0) * kLn2)) * pw) * pfma(splat2(-2.0f * 5.55472f), vdh, splat2(-6.98316f));
  gndh = sel2(ndh_raw.x >= 0.0f && ndh_raw.x <= 1.0f, ndh_raw.y >= 0.0f && ndh_raw.y <= 1.0f, gndh, zero);
  gndl = sel2(ndl_raw.x >= 0.0f && ndl_raw.x <= 1.0f, ndl_raw.y >= 0.0f && ndl_raw.y <= 1.0f, gndl, zero);
  // h: from N.h and v.h
  const f32x2 ghx = pfma(gvdh, splat2(f.vx), gndh * splat2(f.nx)), ghy = pfma(gvdh, splat2(f.vy), gndh * splat2(f.ny)),
              ghz = pfma(gvdh, splat2(f.vz), gndh * splat2(f.nz));
  // hs -> h = hs * rsqrt(max(hh, 1e-6))
  const f32x2 proj_in = pfma(ghz, hz, pfma(ghy, hy, ghx * hx));
  const f32x2 proj = sel2(hh.x >= 1e-6f, hh.y >= 1e-6f, proj_in, zero);
  const f32x2 ghsx = hinv * pfma(-proj, hx, ghx), ghsy = hinv * pfma(-proj, hy, ghy), ghsz = hinv * pfma(-proj, hz, ghz);
  // l: from N.l and hs = (v+l)/2
  const f32x2 hf = splat2(0.5f);
  const f32x2 glx = pfma(hf, ghsx, gndl * splat2(f.nx)), gly = pfma(hf, ghsy, gndl * splat2(f.ny)), glz = pfma(hf, ghsz, gndl * splat2(f.nz));
  // N: from N.h, N.l and l = lx camx + ly camy + lz N
  g.gN[0] = pfma(lz, glx, pfma(gndl, wx, pfma(gndh, hx, g.gN[0])));
  g.gN[1] = pfma(lz, gly, pfma(gndl, wy, pfma(gndh, hy, g.gN[1])));
  g.gN[2] = pfma(lz, glz, pfma(gndl, wz, pfma(gndh, hz, g.gN[2])));
  g.gcx[0] = pfma(lx, glx, g.gcx[0]); g.gcx[1] = pfma(lx, gly, g.gcx[1]); g.gcx[2] = pfma(lx, glz, g.gcx[2]);
  g.gcy[0] = pfma(ly, glx, g.gcy[0]); g.gcy[1] = pfma(ly, gly, g.gcy[1]); g.gcy[2] = pfma(ly, glz, g.gcy[2]);
  return ndl;
}
__device__ __forceinline__ void fold_frame_grad(const FrameGradPk& gp, FrameGrad& g) {
#pragma unroll
  for (int i = 0; i < 3; ++i) { g.gN[i] = gp.gN[i].x + gp.gN[i].y; g.gcx[i] = gp.gcx[i].x + gp.gcx[i].y; g.gcy[i] = gp.gcy[i].x + gp.gcy[i].y; }
  g.galpha2 = gp.galpha2.x + gp.galpha2.y; g.gk = gp.gk.x + gp.gk.y; g.gndv = gp.gndv.x + gp.gndv.y;
}


// env given, envWidth 16: the adjoint in azimuth pairs with the half-wave split of the other backward kernels: one wave = 32 pixels, lanes l and l + 32 own the same pixel
// and integrate one half row (sign) each -- 6 KB row tiles (double-buffered: 12 KB, so LDS no longer caps the CU at six waves),
// half as long work units; the two halves' sums meet once at the end (15 swaps), the lower half applies the frame adjoint
template <int POOL>
__global__ __launch_bounds__(kWave, 2) void brdf_bwd_pk_half_kernel(const Args a) {
  constexpr int EW = 16, HALF = 8;
  __shared__ __attribute__((aligned(16))) float tile[2 * kT32Floats];

  const int lane = threadIdx.x, half = lane >> 5, pl = lane & 31;
  const int own = 1 - half;                         // the half row (sign) this half-wave integrates
  const Pix x = locate_group32(a, (int)blockIdx.x);
  const int b = x.b, p = x.p;
  const int RC = a.R * a.C;
  float pooled[7];
  const Frame f = load_frame_pooled<POOL>(a, x, pooled);
  const size_t o = (size_t)b * 3 * RC + p;
  const float gD0 = a.g_diffuse[o], gD1 = a.g_diffuse[o + RC], gD2 = a.g_diffuse[o + 2 * (size_t)RC];
  const float gs0 = a.g_spec[o], gs1 = a.g_spec[o + RC], gs2 = a.g_spec[o + 2 * (size_t)RC];
  const float gd0 = gD0 * (pooled[0] * kInvPi), gd1 = gD1 * (pooled[1] * kInvPi), gd2 = gD2 * (pooled[2] * kInvPi);

  FrameGradPk gp;
#pragma unroll
  for (int i = 0; i < 3; ++i) gp.gN[i] = gp.gcx[i] = gp.gcy[i] = splat2(0.f);
  gp.galpha2 = gp.gk = gp.gndv = splat2(0.f);
  f32x2 ds[3] = {splat2(0.f), splat2(0.f), splat2(0.f)};
  const SepTable rows = as_sep_table(a.rows);
  const PairTable cpt = as_pair_table(a.cols, EW);
  __amdgpu_buffer_rsrc_t eimg = env_rsrc(a.env_in + (size_t)b * 3 * RC * a.J, RC, a.J);
  const int eh = a.eh;

  tile32_dma_issue(tile, eimg, x.p0, RC, a.J, 0, lane);
  for (int e = 0; e < eh; ++e) {
    const float* cur = tile + (e & 1) * kT32Floats;
    if (e + 1 < eh) {
      tile32_dma_issue(tile + ((e + 1) & 1) * kT32Floats, eimg, x.p0, RC, a.J, (e + 1) * EW, lane);
      wait_vmcnt<6>();
    } else {
      wait_vmcnt<0>();
    }
    const f32x8 row = rows[e];
    const float cr = row[1], om = row[2];
    const f32x2 ss = splat2(own ? -row[0] : row[0]);
#pragma unroll 1
    for (int ap = 0; ap < HALF / 2; ++ap) {
      float ev[3][2];
      tile32_read_pair(cur, pl, own * HALF + ap * 2, ev);
      const f32x4 cs = cpt[ap];
      const f32x2 ca = {cs[0], cs[1]}, sa = {cs[2], cs[3]};
      const f32x2 e0 = {ev[0][0], ev[0][1]}, e1 = {ev[1][0], ev[1][1]}, e2 = {ev[2][0], ev[2][1]};
      const f32x2 Ed = splat2(om) * pfma(splat2(gd2), e2, pfma(splat2(gd1), e1, splat2(gd0) * e0));
      const f32x2 Es = splat2(om) * pfma(splat2(gs2), e2, pfma(splat2(gs1), e1, splat2(gs0) * e0));
      const f32x2 ndl = brdf_pair_bwd(f, ss * ca, ss * sa, cr, a.F0, Ed, Es, gp);
      const f32x2 wt = ndl * splat2(om);
      ds[0] = pfma(wt, e0, ds[0]); ds[1] = pfma(wt, e1, ds[1]); ds[2] = pfma(wt, e2, ds[2]);
    }
  }

  FrameGrad g;
  fold_frame_grad(gp, g);
  float dsum[3] = {ds[0].x + ds[0].y, ds[1].x + ds[1].y, ds[2].x + ds[2].y};
  // the two half rows: every sum is linear in the directions
  auto both = [](float& v) { float d_ = v, s_ = v; swap32(d_, s_); v = d_ + s_; };
#pragma unroll
  for (int i = 0; i < 3; ++i) { both(g.gN[i]); both(g.gcx[i]); both(g.gcy[i]); both(dsum[i]); }
  both(g.galpha2); both(g.gk); both(g.gndv);
  float gpn[3], gprho;
  frame_bwd(pooled[3], pooled[4], pooled[5], pooled[6], f, g, gpn, gprho);
  if (x.active && half == 0) {
    const unsigned off = pooled_offset<POOL>(p, a.C, a.imW);
    const size_t plane = (size_t)a.imH * a.imW;
    float* ga = a.g_albedo + (size_t)b * 3 * plane;
    float* gn = a.g_normal + (size_t)b * 3 * plane;
    float* gr = a.g_rough + (size_t)b * plane;
    scatter_pooled<POOL>(ga, off, a.imW, gD0 * kInvPi * dsum[0]);
    scatter_pooled<POOL>(ga + plane, off, a.imW, gD1 * kInvPi * dsum[1]);
    scatter_pooled<POOL>(ga + 2 * plane, off, a.imW, gD2 * kInvPi * dsum[2]);
    scatter_pooled<POOL>(gn, off, a.imW, gpn[0]);
    scatter_pooled<POOL>(gn + plane, off, a.imW, gpn[1]);
    scatter_pooled<POOL>(gn + 2 * plane, off, a.imW, gpn[2]);
    scatter_pooled<POOL>(gr, off, a.imW, gprho);
  }
}

// no env image (the fused API with need_env = False): the radiance of the lane's half row is re-evaluated from the SG lobes -- all
// KP lobes in every lane, the packed exponent / accumulation code of the forward kernels for ONE sign (so U_ka is not shared
// between the half rows here: 6 packed + 2 v_exp per lobe and azimuth pair) -- and fed to the packed adjoint
template <int POOL, int KP>
__global__ __launch_bounds__(kWave, 2) void brdf_bwd_pk_sg_kernel(const Args a) {
  constexpr int EW = 16, HALF = 8;
  const int lane = threadIdx.x, half = lane >> 5;
  const int own = 1 - half;
  const Pix x = locate_group32(a, (int)blockIdx.x);
  const int b = x.b, p = x.p;
  const int RC = a.R * a.C;

  LobesPk<KP> P;
  load_lobes_pk<KP, true>(a, b, (unsigned)p, x.active, 0, P, false);
  float pooled[7];
  const Frame f = load_frame_pooled<POOL>(a, x, pooled);
  const size_t o = (size_t)b * 3 * RC + p;
  const float gD0 = a.g_diffuse[o], gD1 = a.g_diffuse[o + RC], gD2 = a.g_diffuse[o + 2 * (size_t)RC];
  const float gs0 = a.g_spec[o], gs1 = a.g_spec[o + RC], gs2 = a.g_spec[o + 2 * (size_t)RC];
  const float gd0 = gD0 * (pooled[0] * kInvPi), gd1 = gD1 * (pooled[1] * kInvPi), gd2 = gD2 * (pooled[2] * kInvPi);

  FrameGradPk gp;
#pragma unroll
  for (int i = 0; i < 3; ++i) gp.gN[i] = gp.gcx[i] = gp.gcy[i] = splat2(0.f);
  gp.galpha2 = gp.gk = gp.gndv = splat2(0.f);
  f32x2 ds[3] = {splat2(0.f), splat2(0.f), splat2(0.f)};
  const SepTable rows = as_sep_table(a.rows);
  const PairTable cpt = as_pair_table(a.cols, EW);
  const int eh = a.eh;

  for (int e = 0; e < eh; ++e) {
    const f32x8 row = rows[e];
    const float cr = row[1], om = row[2];
    const f32x2 ss = splat2(own ? -row[0] : row[0]);
    f32x2 Ck[KP / 2];
#pragma unroll
    for (int m = 0; m < KP / 2; ++m) Ck[m] = pfma(P.azp[m], splat2(cr), -P.lpp[m]);
#pragma unroll 1
    for (int ap = 0; ap < HALF / 2; ++ap) {
      fence_lobes<KP>(P);
#pragma unroll
      for (int m = 0; m < KP / 2; ++m) SGR_FENCE2(Ck[m]);
      const f32x4 cs = cpt[ap];
      const f32x2 ca = {cs[0], cs[1]}, sa = {cs[2], cs[3]};
      f32x2 e0 = splat2(0.f), e1 = splat2(0.f), e2 = splat2(0.f);
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        const f32x2 ck = half_of(Ck[k / 2], k & 1), w2 = half_of(P.w2p[k / 2], k & 1);
        const f32x2 U = pfma(SGR_HI(P.axy[k]), sa, SGR_LO(P.axy[k]) * ca);
        const f32x2 t = pfma(ss, U, ck);
        const f32x2 ex = {fexp2(t.x), fexp2(t.y)};
        e0 = pfma(SGR_LO(P.w01[k]), ex, e0);
        e1 = pfma(SGR_HI(P.w01[k]), ex, e1);
        e2 = pfma(w2, ex, e2);
      }
      const f32x2 Ed = splat2(om) * pfma(splat2(gd2), e2, pfma(splat2(gd1), e1, splat2(gd0) * e0));
      const f32x2 Es = splat2(om) * pfma(splat2(gs2), e2, pfma(splat2(gs1), e1, splat2(gs0) * e0));
      const f32x2 ndl = brdf_pair_bwd(f, ss * ca, ss * sa, cr, a.F0, Ed, Es, gp);
      const f32x2 wt = ndl * splat2(om);
      ds[0] = pfma(wt, e0, ds[0]); ds[1] = pfma(wt, e1, ds[1]); ds[2] = pfma(wt, e2, ds[2]);
    }
  }

  FrameGrad g;
  fold_frame_grad(gp, g);
  float dsum[3] = {ds[0].x + ds[0].y, ds[1].x + ds[1].y, ds[2].x + ds[2].y};
  auto both = [](float& v) { float d_ = v, s_ = v; swap32(d_, s_); v = d_ + s_; };
#pragma unroll
  for (int i = 0; i < 3; ++i) { both(g.gN[i]); both(g.gcx[i]); both(g.gcy[i]); both(dsum[i]); }
  both(g.galpha2); both(g.gk); both(g.gndv);
  float gpn[3], gprho;
  frame_bwd(pooled[3], pooled[4], pooled[5], pooled[6], f, g, gpn, gprho);
  if (x.active && half == 0) {
    const unsigned off = pooled_offset<POOL>(p, a.C, a.imW);
    const size_t plane = (size_t)a.imH * a.imW;
    float* ga = a.g_albedo + (size_t)b * 3 * plane;
    float* gn = a.g_normal + (size_t)b * 3 * plane;
    float* gr = a.g_rough + (size_t)b * plane;
    scatter_pooled<POOL>(ga, off, a.imW, gD0 * kInvPi * dsum[0]);
    scatter_pooled<POOL>(ga + plane, off, a.imW, gD1 * kInvPi * dsum[1]);
    scatter_pooled<POOL>(ga + 2 * plane, off, a.imW, gD2 * kInvPi * dsum[2]);
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
    hipLaunchKernelGGL((brdf_bwd_pk_half_kernel<POOL>), dim3((unsigned)(a.bn * ((a.R * a.C + kPx - 1) / kPx))), dim3(kWave), 0, st, a);
    return (int)hipGetLastError();
  }
  if (a.K > 0 && a.K <= 12 && a.ew == 16 && !sgr_generic_forced()) {
    const dim3 grid((unsigned)(a.bn * ((a.R * a.C + kPx - 1) / kPx))), block(kWave);
    if (a.K <= 6) hipLaunchKernelGGL((brdf_bwd_pk_sg_kernel<POOL, 6>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((brdf_bwd_pk_sg_kernel<POOL, 12>), grid, block, 0, st, a);
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
  a.rows = reinterpret_cast<const float*>(a.dirs) + 4 * (size_t)a.Jpad;      // separable form of the table (include/sgrender.h)
  a.cols = a.rows + 8 * (size_t)((eh + 1) / 2 * 2);
  SGR_REQUIRE(premap >= 0 && premap <= 2, "sgr_render_bwd_brdf: premap must be 0, 1 or 2");
  a.F0 = F0; a.premap = premap == 1 ? 1 : 0; a.eh = eh; a.ew = ew;      // 2 = post-tan SG inputs: nothing to pre-map, no SG chain rule here
  const hipStream_t st = (hipStream_t)stream;
  return sgr_check(imH == R ? brdf_launch<1>(a, st) : brdf_launch<2>(a, st), "sgr_render_bwd_brdf");
}
