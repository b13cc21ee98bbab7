// Fast-path kernels for the reference's direction grids (envWidth 16 or 32, any envHeight):
// the hemisphere table is a tensor product  l_j = (s_e ca_a, s_e sa_a, c_e),  j = e*EW + a, and
// the azimuths are antisymmetric (ca, sa)(a + EW/2) = -(ca, sa)(a).  Per lobe k and direction j
//
//     lam (a_k . l_j - 1) = s_e * U_ka + C_ke ,     U_ka = lam (ax ca_a + ay sa_a),  C_ke = lam (az c_e - 1)
//
// so one FMA (with +-s_e as an SGPR operand) replaces the 3-FMA dot product + scale, U_ka serves the
// 2*RPC directions (e, a), (e, a + EW/2) of a chunk and C_ke all EW azimuths of a row.  The
// microfacet terms use the same factorisation in the local frame (sgr_math.h: brdf_local_dir).
// Everything else (lanes <-> pixels, LDS-transposed env tiles) is as in sgr_common.h.
#pragma once
#include "sgr_common.h"
#include <type_traits>
#include "sgr_launch.h"

#ifndef SGR_TABLE_PREFETCH
#define SGR_TABLE_PREFETCH 1   // half-wave backward: scalar table entries requested one iteration ahead (324 -> 312 us; no gain in the forward)
#endif
#ifndef SGR_HALF_TD
#define SGR_HALF_TD 32   // half-wave forward: directions per flushed tile row (16: 64-byte segments, 32: 128-byte)
#endif
#ifndef SGR_FWD_DIRECT
#define SGR_FWD_DIRECT 0
#endif
#ifndef SGR_DIR_BARRIER
#define SGR_DIR_BARRIER 1   // scheduling fence between azimuths (A/B switch)
#endif

namespace sgr {

template <int KP>
struct Lobes {   // raw or folded SG parameters of the lane's pixel
  float ax[KP], ay[KP], az[KP], lp[KP], w0[KP], w1[KP], w2[KP];
};

// FOLD: ax,ay,az are pre-multiplied by lp = lam*log2e (forward); otherwise unit axes (backward).
// Two passes: every load of every lobe is in flight before anything consumes one -- one trip to memory per wave
// instead of one per lobe (the pre-map's conditional stores would otherwise fence the next lobe's loads behind
// them).  Lobes past K re-read lobe K-1 and get zero weights.
template <int KP, bool FOLD>
__device__ __forceinline__ void load_lobes(const Args& a, const Pix& x, int kg, Lobes<KP>& L, bool write_tan) {
  const int RC = a.R * a.C, K = a.K;
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    const int kk = min(kg + k, K - 1);
    const size_t ab = ((size_t)(x.b * K + kk) * 3) * RC;   // wave-uniform plane bases
    const size_t lb = (size_t)(x.b * K + kk) * RC;
    const unsigned up = (unsigned)x.p;                      // the lane's 32-bit offset
    L.ax[k] = (a.axis + ab)[up]; L.ay[k] = (a.axis + ab + RC)[up]; L.az[k] = (a.axis + ab + 2 * (size_t)RC)[up];
    L.lp[k] = (a.lamb + lb)[up];
    L.w0[k] = (a.weight + ab)[up]; L.w1[k] = (a.weight + ab + RC)[up]; L.w2[k] = (a.weight + ab + 2 * (size_t)RC)[up];
  }
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    const bool live = kg + k < K;
    float l = L.lp[k], t0 = L.w0[k], t1 = L.w1[k], t2 = L.w2[k];
    if (a.premap == 1) {
      l = premap(l);
      t0 = premap(t0); t1 = premap(t1); t2 = premap(t2);
      if (write_tan && live && x.active) {
        const size_t ab = ((size_t)(x.b * K + kg + k) * 3) * RC;
        const size_t lb = (size_t)(x.b * K + kg + k) * RC;
        const unsigned up = (unsigned)x.p;
        if (a.lamb_tan) (a.lamb_tan + lb)[up] = l;
        if (a.weight_tan) {
          (a.weight_tan + ab)[up] = t0; (a.weight_tan + ab + RC)[up] = t1; (a.weight_tan + ab + 2 * (size_t)RC)[up] = t2;
        }
      }
    }
    const float lp = l * kLog2e;
    L.lp[k] = lp;
    if (FOLD) { L.ax[k] *= lp; L.ay[k] *= lp; L.az[k] *= lp; }
    L.w0[k] = live ? t0 : 0.0f; L.w1[k] = live ? t1 : 0.0f; L.w2[k] = live ? t2 : 0.0f;
  }
}

// One direction's (quadrature weight * ndl, spec): orthonormal-frame path when the whole wave is
// non-degenerate (wave-uniform branch), Gram-matrix path otherwise.
// Per-azimuth terms built from these are row-invariant too; same LICM fence as for the lobe axes.
__device__ __forceinline__ void fence_row_invariants(PixLocal& q) {
  asm volatile("" : "+v"(q.vBx)); asm volatile("" : "+v"(q.vBy));
  asm volatile("" : "+v"(q.nBx)); asm volatile("" : "+v"(q.nBy));
  asm volatile("" : "+v"(q.Gxx)); asm volatile("" : "+v"(q.Gxy)); asm volatile("" : "+v"(q.Gyy));
  asm volatile("" : "+v"(q.Gxz)); asm volatile("" : "+v"(q.Gyz));
}
struct RowCtx {
  float sr, cr, om, s2r, scr;   // s_e, c_e, omega_e, s_e^2, 2 s_e c_e
  float Cv, Cn, Cz;             // general path: vBz c_e, nBz c_e, Gzz c_e^2
  RowOrtho ro;
};
__device__ __forceinline__ RowCtx make_row_ctx(const PixLocal& q, const f32x8 row, bool with_brdf) {
  RowCtx rc;
  rc.sr = row[0]; rc.cr = row[1]; rc.om = row[2]; rc.s2r = row[3]; rc.scr = row[4];
  rc.Cv = rc.Cn = rc.Cz = 0.0f;
  rc.ro.nw = rc.ro.rowc = rc.ro.c1n2 = rc.ro.wt = rc.ro.Cv = 0.0f;
  if (with_brdf) {
    rc.Cv = q.vBz * row[1];
    rc.Cn = q.nBz * row[1];
    rc.Cz = q.Gzz * row[5];
    rc.ro = make_row_ortho(q, row[1], row[2]);
  }
  return rc;
}
typedef const f32x4 __attribute__((address_space(4))) * XTable;   // per azimuth (ca^2, 2 ca sa, sa^2, 0)
template <bool ORTHO>
__device__ __forceinline__ void shade_dir(const PixLocal& q, const RowCtx& rc, int sg, float ca, float sa, XTable xt, int a, float& wt,
                                          float& sp) {
  const float ss = sg ? -rc.sr : rc.sr;
  const float Pv = fmaf(q.vBy, sa, q.vBx * ca);
  if (ORTHO) {
    sp = brdf_ortho_dir(q, rc.ro, ss, ca, sa, Pv);
    wt = rc.ro.wt;
  } else {
    const float sc = sg ? -rc.scr : rc.scr;
    const float Pn = fmaf(q.nBy, sa, q.nBx * ca);
    const f32x4 ex = xt[a];
    const float Qa = fmaf(q.Gyy, ex[2], fmaf(q.Gxy, ex[1], q.Gxx * ex[0]));
    const float Ra = fmaf(q.Gyz, sa, q.Gxz * ca);
    float ndl;
    brdf_local_dir(q, fmaf(ss, Pv, rc.Cv), fmaf(ss, Pn, rc.Cn), fmaf(rc.s2r, Qa, fmaf(sc, Ra, rc.Cz)), ndl, sp);
    wt = ndl * rc.om;
  }
}



// ============================== utils.predToShading (utils.py:156-195) ============================
// shading_c = max( sum_j env_c(l_j) cos(El_j) sin(El_j), 0 ): the SG mixture integrated against the
// cosine-weighted hemisphere measure (no microfacet terms, no env image) -- the forward inner loop with a
// row-constant weight.
template <int KP, int EW>
__global__ __launch_bounds__(kWave, 2) void shading_fast_kernel(const Args a) {
  constexpr int NQ = EW / 8;
  const Pix x = locate(a);
  const int b = x.b, p = x.p;
  const int RC = a.R * a.C;
  Lobes<KP> L;
  load_lobes<KP, true>(a, x, 0, L, false);
  const SepTable rows = as_sep_table(a.rows);
  const SepTable cst = as_sep_table(a.cols);
  float d0 = 0.f, d1 = 0.f, d2 = 0.f;
  for (int e = 0; e < a.eh; ++e) {
#pragma unroll
    for (int k = 0; k < KP; ++k) { asm volatile("" : "+v"(L.ax[k])); asm volatile("" : "+v"(L.ay[k])); }
    const f32x8 row = rows[e];
    const float sr = row[0];
    float Ck[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) Ck[k] = fmaf(L.az[k], row[1], -L.lp[k]);
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
#pragma unroll 1
    for (int aq = 0; aq < NQ; ++aq) {
      const f32x8 cs = cst[aq];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int k = 0; k < KP; ++k) {
          const float U = fmaf(L.ay[k], cs[2 * i + 1], L.ax[k] * cs[2 * i]);
          const float ee = fexp2(fmaf(sr, U, Ck[k])) + fexp2(fmaf(-sr, U, Ck[k]));
          r0 = fmaf(L.w0[k], ee, r0);
          r1 = fmaf(L.w1[k], ee, r1);
          r2 = fmaf(L.w2[k], ee, r2);
        }
      }
    }
    const float wrow = 0.5f * row[4];      // cos(El) sin(El)
    d0 = fmaf(wrow, r0, d0); d1 = fmaf(wrow, r1, d1); d2 = fmaf(wrow, r2, d2);
  }
  if (x.active) {
    const size_t o = (size_t)b * 3 * RC;
    const unsigned up = (unsigned)p;
    (a.diffuse + o)[up] = fmaxf(d0, 0.0f);
    (a.diffuse + o + RC)[up] = fmaxf(d1, 0.0f);
    (a.diffuse + o + 2 * (size_t)RC)[up] = fmaxf(d2, 0.0f);
  }
}

// fast path applies to the reference's direction grids; the LDS-DMA descriptors and the 32-bit lane offsets
// address one image's env tensor with signed 32-bit byte offsets
static inline bool fast_ok(const Args& a) {
  const long long env_bytes = 3LL * a.R * a.C * a.J * 4;
  return (a.ew == 16 || a.ew == 32) && env_bytes < (1LL << 31);
}

}  // namespace sgr
