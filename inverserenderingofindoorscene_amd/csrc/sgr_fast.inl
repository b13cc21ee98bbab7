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

// ============================== forward ==========================================================
//
// HAS_GT (fused objective, no env image): the ground-truth env rows stream in by LDS-DMA (one 12 KB tile; row
// e+1 is requested the moment row e's last quad is in registers and has a quad's worth of arithmetic to land --
// whole 64-byte segments, where 16-byte quad tiles made L2 fetch every segment twice: 219 -> 201 us)
// and every lane accumulates <pred, gt>, <pred, pred> and sum(gt) of its pixel on the fly -- the statistics
// behind the env mask and the LSregress scale (wrapperBRDFLight.py:172-176, models.py:7-21) -- so the
// predicted env never goes to memory.  Per-wave partials land in a.ws[(b*tiles + tile)*3 + {0,1,2}].
template <int KP, int POOL, int EW, int TJ, bool WRITE_ENV, bool DO_RENDER, bool HAS_GT = false>
__global__ __launch_bounds__(kWave, 2) void fwd_fast_kernel(const Args a) {
  static_assert(TJ % EW == 0, "a tile holds whole table rows");
  static_assert(!HAS_GT || (TJ == EW && EW == 16 && !WRITE_ENV), "fused statistics: one 16-direction row per tile, no env output");
  constexpr int RPC = TJ / EW;      // table rows per tile (TJ=32: 2 for EW=16, 1 for EW=32)
  constexpr int HALF = EW / 2;
  constexpr int NQ = HALF / 4;      // azimuth quads per half row
  __shared__ __attribute__((aligned(16))) float tile[WRITE_ENV ? Tile<TJ>::kFloats : 4];
  using GD = DmaTile<16>;          // ground-truth env, one whole table row (64-byte segments), single buffer
  __shared__ __attribute__((aligned(16))) float gtile[HAS_GT ? GD::kFloats : 4];

  const Pix x = locate(a);
  const int lane = x.lane, b = x.b, p = x.p;
  const int RC = a.R * a.C;

  Lobes<KP> L;
  load_lobes<KP, true>(a, x, 0, L, true);

  PixLocal q;
  float alb[3] = {0.f, 0.f, 0.f};
  bool ortho = true;
  if (DO_RENDER) {
    const Frame f = load_frame<POOL>(a, x, alb);
    q = make_local(f, a.F0);
    ortho = __all(frame_is_orthonormal(q));
  }
  float d0 = 0.f, d1 = 0.f, d2 = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f;
  const SepTable rows = as_sep_table(a.rows);
  const SepTable cst = as_sep_table(a.cols);                                   // [(EW/2)/4] x 4 x (ca, sa)
  const XTable xt = (XTable)(a.cols + EW);                                     // extras, general path only
  const size_t img = (size_t)b * 3 * RC * a.J;
  const int ehp = RPC == 2 ? ((a.eh + 1) & ~1) : a.eh;
  float s_pg = 0.f, s_pp = 0.f, s_g = 0.f;
  __amdgpu_buffer_rsrc_t gimg = env_rsrc(HAS_GT ? a.env_gt + img : a.view, RC, a.J);
  if (HAS_GT) tile_dma_issue<16>(gtile, gimg, x.p0, RC, a.J, 0, lane);

  // the wave-uniform frame test is hoisted out of the direction loops: two copies of the row loop
  auto row_loop = [&](auto ortho_c) {
  for (int e0 = 0; e0 < ehp; e0 += RPC) {
  #pragma unroll
      for (int k = 0; k < KP; ++k) { asm volatile("" : "+v"(L.ax[k])); asm volatile("" : "+v"(L.ay[k])); }
      if (DO_RENDER) fence_row_invariants(q);
      float sr[RPC], Ck[KP][RPC];
      RowCtx rc[RPC];
  #pragma unroll
      for (int r = 0; r < RPC; ++r) {
        const f32x8 row = rows[e0 + r];
        sr[r] = row[0];
  #pragma unroll
        for (int k = 0; k < KP; ++k) Ck[k][r] = fmaf(L.az[k], row[1], -L.lp[k]);
        rc[r] = make_row_ctx(q, row, DO_RENDER);
      }
  #pragma unroll 1
      for (int aq = 0; aq < NQ; ++aq) {
        const f32x8 cs = cst[aq];  // (ca, sa) of the quad's four azimuths: one scalar load per quad
        float acc[RPC][2][3][4];   // [row][sign][colour][azimuth in quad]
  #pragma unroll
        for (int r = 0; r < RPC; ++r)
  #pragma unroll
          for (int sg = 0; sg < 2; ++sg)
  #pragma unroll
            for (int c = 0; c < 3; ++c)
  #pragma unroll
              for (int i = 0; i < 4; ++i) acc[r][sg][c][i] = 0.0f;
  
  #pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float ca = cs[2 * i], sa = cs[2 * i + 1];
  #pragma unroll
          for (int k = 0; k < KP; ++k) {
            const float U = fmaf(L.ay[k], sa, L.ax[k] * ca);
  #pragma unroll
            for (int r = 0; r < RPC; ++r) {
              const float ep = fexp2(fmaf(sr[r], U, Ck[k][r]));
              const float em = fexp2(fmaf(-sr[r], U, Ck[k][r]));
              acc[r][0][0][i] = fmaf(L.w0[k], ep, acc[r][0][0][i]);
              acc[r][0][1][i] = fmaf(L.w1[k], ep, acc[r][0][1][i]);
              acc[r][0][2][i] = fmaf(L.w2[k], ep, acc[r][0][2][i]);
              acc[r][1][0][i] = fmaf(L.w0[k], em, acc[r][1][0][i]);
              acc[r][1][1][i] = fmaf(L.w1[k], em, acc[r][1][1][i]);
              acc[r][1][2][i] = fmaf(L.w2[k], em, acc[r][1][2][i]);
            }
          }
          if (DO_RENDER) {
  #pragma unroll
            for (int r = 0; r < RPC; ++r) {
  #pragma unroll
              for (int sg = 0; sg < 2; ++sg) {
                float wt, sp;
                shade_dir<decltype(ortho_c)::value>(q, rc[r], sg, ca, sa, xt, aq * 4 + i, wt, sp);
                const float sw = sp * wt;
                d0 = fmaf(wt, acc[r][sg][0][i], d0);
                d1 = fmaf(wt, acc[r][sg][1][i], d1);
                d2 = fmaf(wt, acc[r][sg][2][i], d2);
                s0 = fmaf(sw, acc[r][sg][0][i], s0);
                s1 = fmaf(sw, acc[r][sg][1][i], s1);
                s2 = fmaf(sw, acc[r][sg][2][i], s2);
              }
            }
          }
  #if SGR_DIR_BARRIER
          __builtin_amdgcn_sched_barrier(0);
  #endif
        }
        if (HAS_GT) {
          // whole rows in a single 12 KB tile: the row was requested when the previous row's last quad had been read
          // (a quad's worth of arithmetic ago); the next row is requested as soon as this row's last quad is in registers
          if (aq == 0) wait_vmcnt<0>();
  #pragma unroll
          for (int h = 0; h < 2; ++h) {
            float g[2][3][2];
            tile_dma_read_pairs<16>(gtile, lane, aq * 4 + 2 * h, HALF + aq * 4 + 2 * h, g);
            if (h == 1 && aq == NQ - 1 && e0 + 1 < ehp) tile_dma_issue<16>(gtile, gimg, x.p0, RC, a.J, (e0 + 1) * EW, lane);
  #pragma unroll
            for (int sg = 0; sg < 2; ++sg)
  #pragma unroll
              for (int c = 0; c < 3; ++c)
  #pragma unroll
                for (int i = 0; i < 2; ++i) {
                  const float pv = acc[0][sg][c][2 * h + i], gv = g[sg][c][i];
                  s_pg = fmaf(pv, gv, s_pg);
                  s_pp = fmaf(pv, pv, s_pp);
                  s_g += gv;
                }
          }
        }
        if (WRITE_ENV) {
  #if SGR_FWD_DIRECT
          // experiment: per-lane 16-byte stores straight from registers (no LDS transpose)
          if (x.active) {
  #pragma unroll
            for (int r = 0; r < RPC; ++r)
  #pragma unroll
              for (int sg = 0; sg < 2; ++sg)
  #pragma unroll
                for (int c = 0; c < 3; ++c) {
                  float* base = a.env_out + img + (size_t)c * RC * a.J + (size_t)((e0 + r) * EW + sg * HALF + aq * 4);   // uniform
                  f32x4 nv = {acc[r][sg][c][0], acc[r][sg][c][1], acc[r][sg][c][2], acc[r][sg][c][3]};
                  *reinterpret_cast<f32x4*>(base + (unsigned)(p * a.J)) = nv;
                }
          }
  #else
  #pragma unroll
          for (int r = 0; r < RPC; ++r)
  #pragma unroll
            for (int sg = 0; sg < 2; ++sg)
              tile_row_write<TJ>(tile, lane, r * EW + sg * HALF + aq * 4, acc[r][sg][0], acc[r][sg][1], acc[r][sg][2]);
  #endif
        }
      }
      if (WRITE_ENV && !SGR_FWD_DIRECT) {
        __syncthreads();
        tile_store_global<TJ, true>(tile, a.env_out + img, x.p0, RC, a.J, e0 * EW, lane);
        __syncthreads();
      }
    }
  };
  if (ortho) row_loop(std::true_type{}); else row_loop(std::false_type{});

  if (HAS_GT) {
    // env mask of the pixel (wrapperBRDFLight.py:172-174) and the wave's share of the per-image sums
    const float not_dark = (s_g / (3.0f * (float)a.J)) > 0.001f ? 1.0f : 0.0f;
    const float m = x.active ? seg_small_at(a, b, p) * a.env_ind[b] * not_dark : 0.0f;
    if (x.active) (a.mask + (size_t)b * RC)[(unsigned)p] = m;
    float r0 = m * m * s_pg, r1 = m * m * s_pp, r2 = m;
  #pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      r0 += __shfl_xor(r0, off, 64); r1 += __shfl_xor(r1, off, 64); r2 += __shfl_xor(r2, off, 64);
    }
    if (lane == 0) {
      float* w = a.ws + (size_t)blockIdx.x * 3;
      w[0] = r0; w[1] = r1; w[2] = r2;
    }
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

// ============================== forward, half-wave lobe split =====================================
// One wave = 32 pixels x 2 lobe groups (lanes l and l+32 own the same pixel; lobes 0..5 / 6..11), like the
// half-wave backward.  Each half accumulates its six lobes' share of all 8 directions of an azimuth quad;
// swap(D = share of half row 1, S = share of half row 0); D + S gives lanes 0..31 the radiance of half row 1 and
// lanes 32..63 that of half row 0, so each half then shades / stores / takes statistics for 4 of the 8 directions.
// Same arithmetic per pixel as fwd_fast_kernel plus 12 swaps + 12 adds per quad, but the work unit is half as long
// (9600 instead of 4800 waves at config 2: half the ramp/tail) and the register footprint allows 3 waves per SIMD.
// HAS_GT: the statistics of the fused objective (see fwd_fast_kernel), each half-wave against the ground truth of the half
// row it owns; the 32-pixel ground-truth row (6 KB) sits in a single early-requested LDS-DMA tile.
template <int POOL, bool WRITE_ENV, bool DO_RENDER, int OCC, bool HAS_GT = false>
__global__ __launch_bounds__(kWave, OCC) void fwd_half_kernel(const Args a) {
  static_assert(!(HAS_GT && WRITE_ENV), "the statistics variant does not write the env image");
  constexpr int EW = 16, HALF = 8, NQ = 2, KPW = 6;
  constexpr int TD = SGR_HALF_TD;                   // directions per flushed tile row: two table rows -> 128-byte segments
  __shared__ __attribute__((aligned(16))) float tile[WRITE_ENV ? T32Out<TD>::kFloats : 4];
  __shared__ __attribute__((aligned(16))) float gtile[HAS_GT ? kT32Floats : 4];

  const int lane = threadIdx.x, half = lane >> 5, pl = lane & 31;
  const int own = 1 - half;                         // the half row (sign) whose totals this half-wave ends up holding
  const int RC = a.R * a.C, K = a.K;
  Pix x;
  x.lane = lane;
  {
    const int tiles = (RC + kPx - 1) / kPx;
    x.b = blockIdx.x / tiles;
    x.p0 = (blockIdx.x - x.b * tiles) * kPx;
    x.active = (x.p0 + pl) < RC;
    x.p = x.active ? (x.p0 + pl) : (RC - 1);
  }
  const int b = x.b, p = x.p;

  __amdgpu_buffer_rsrc_t gimg = env_rsrc(HAS_GT ? a.env_gt + (size_t)b * 3 * RC * a.J : a.view, RC, a.J);
  if (HAS_GT) tile32_dma_issue(gtile, gimg, x.p0, RC, a.J, 0, lane);
  float s_pg = 0.f, s_pp = 0.f, s_g = 0.f;

  // this half's lobes, folded (axis pre-multiplied by lam * log2e)
  Lobes<KPW> L;
  {
    const float* axis_b = a.axis + (size_t)b * K * 3 * RC;
    const float* lamb_b = a.lamb + (size_t)b * K * RC;
    const float* weight_b = a.weight + (size_t)b * K * 3 * RC;
#pragma unroll
    for (int k = 0; k < KPW; ++k) {
      const int kc = min(half * KPW + k, K - 1);
      const unsigned o3 = (unsigned)(kc * 3 * RC + p), o1 = (unsigned)(kc * RC + p);
      L.ax[k] = axis_b[o3]; L.ay[k] = axis_b[o3 + RC]; L.az[k] = axis_b[o3 + 2 * RC];
      L.lp[k] = lamb_b[o1];
      L.w0[k] = weight_b[o3]; L.w1[k] = weight_b[o3 + RC]; L.w2[k] = weight_b[o3 + 2 * RC];
    }
#pragma unroll
    for (int k = 0; k < KPW; ++k) {
      const int kk = half * KPW + k;
      const bool live = kk < K;
      float l = L.lp[k], t0 = L.w0[k], t1 = L.w1[k], t2 = L.w2[k];
      if (a.premap == 1) {
        l = premap(l); t0 = premap(t0); t1 = premap(t1); t2 = premap(t2);
        if (live && x.active) {
          const unsigned o3 = (unsigned)(kk * 3 * RC + p), o1 = (unsigned)(kk * RC + p);
          if (a.lamb_tan) (a.lamb_tan + (size_t)b * K * RC)[o1] = l;
          if (a.weight_tan) {
            float* wt_b = a.weight_tan + (size_t)b * K * 3 * RC;
            wt_b[o3] = t0; wt_b[o3 + RC] = t1; wt_b[o3 + 2 * RC] = t2;
          }
        }
      }
      const float lp = l * kLog2e;
      L.lp[k] = lp;
      L.ax[k] *= lp; L.ay[k] *= lp; L.az[k] *= lp;
      L.w0[k] = live ? t0 : 0.0f; L.w1[k] = live ? t1 : 0.0f; L.w2[k] = live ? t2 : 0.0f;
    }
  }

  PixLocal q;
  float alb[3] = {0.f, 0.f, 0.f};
  bool ortho = true;
  if (DO_RENDER) {
    const Frame f = load_frame<POOL>(a, x, alb);
    q = make_local(f, a.F0);
    ortho = __all(frame_is_orthonormal(q));
  }
  float d0 = 0.f, d1 = 0.f, d2 = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f;
  const SepTable rows = as_sep_table(a.rows);
  const SepTable cst = as_sep_table(a.cols);                                   // [(EW/2)/4] x 4 x (ca, sa)
  const XTable xt = (XTable)(a.cols + EW);
  const size_t img = (size_t)b * 3 * RC * a.J;
  const int eh = a.eh;

  auto row_loop = [&](auto ortho_c) {
    for (int e = 0; e < eh; ++e) {
#pragma unroll
      for (int k = 0; k < KPW; ++k) { asm volatile("" : "+v"(L.ax[k])); asm volatile("" : "+v"(L.ay[k])); }   // no LICM of U_ka
      if (DO_RENDER) fence_row_invariants(q);
      const f32x8 row = rows[e];
      const float sr = row[0];
      float Ck[KPW];
#pragma unroll
      for (int k = 0; k < KPW; ++k) Ck[k] = fmaf(L.az[k], row[1], -L.lp[k]);
      const RowCtx rc = make_row_ctx(q, row, DO_RENDER);
#pragma unroll 1
      for (int aq = 0; aq < NQ; ++aq) {
        const f32x8 cs = cst[aq];
        float acc[2][3][4];
#pragma unroll
        for (int sg = 0; sg < 2; ++sg)
#pragma unroll
          for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[sg][c][i] = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float ca = cs[2 * i], sa = cs[2 * i + 1];
#pragma unroll
          for (int k = 0; k < KPW; ++k) {
            const float U = fmaf(L.ay[k], sa, L.ax[k] * ca);
            const float ep = fexp2(fmaf(sr, U, Ck[k]));
            const float em = fexp2(fmaf(-sr, U, Ck[k]));
            acc[0][0][i] = fmaf(L.w0[k], ep, acc[0][0][i]);
            acc[0][1][i] = fmaf(L.w1[k], ep, acc[0][1][i]);
            acc[0][2][i] = fmaf(L.w2[k], ep, acc[0][2][i]);
            acc[1][0][i] = fmaf(L.w0[k], em, acc[1][0][i]);
            acc[1][1][i] = fmaf(L.w1[k], em, acc[1][1][i]);
            acc[1][2][i] = fmaf(L.w2[k], em, acc[1][2][i]);
          }
        }
        // radiance of the half row this half-wave owns
        float tot[3][4];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float d_ = acc[1][c][i], s_ = acc[0][c][i];
            swap32(d_, s_);
            tot[c][i] = d_ + s_;
          }
        if (DO_RENDER) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float wt, sp;
            shade_dir<decltype(ortho_c)::value>(q, rc, own, cs[2 * i], cs[2 * i + 1], xt, aq * 4 + i, wt, sp);
            const float sw = sp * wt;
            d0 = fmaf(wt, tot[0][i], d0);
            d1 = fmaf(wt, tot[1][i], d1);
            d2 = fmaf(wt, tot[2][i], d2);
            s0 = fmaf(sw, tot[0][i], s0);
            s1 = fmaf(sw, tot[1][i], s1);
            s2 = fmaf(sw, tot[2][i], s2);
          }
        }
        if (WRITE_ENV) tile32_write4<TD>(tile, pl, (e % (TD / EW)) * EW + own * HALF + aq * 4, tot[0], tot[1], tot[2]);
        if (HAS_GT) {
          if (aq == 0) wait_vmcnt<0>();       // the row was requested a quad's worth of arithmetic ago
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            float g[3][2];
            tile32_read_pair(gtile, pl, own * HALF + aq * 4 + 2 * h, g);
            if (h == 1 && aq == NQ - 1 && e + 1 < eh) tile32_dma_issue(gtile, gimg, x.p0, RC, a.J, (e + 1) * EW, lane);
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
              for (int i = 0; i < 2; ++i) {
                const float pv = tot[c][2 * h + i], gv = g[c][i];
                s_pg = fmaf(pv, gv, s_pg);
                s_pp = fmaf(pv, pv, s_pp);
                s_g += gv;
              }
          }
        }
      }
      if (WRITE_ENV && ((e + 1) % (TD / EW) == 0 || e + 1 == eh)) {
        const int rows_in_tile = e % (TD / EW) + 1;
        __syncthreads();
        tile32_store_global<TD>(tile, a.env_out + img, x.p0, RC, a.J, (e + 1 - rows_in_tile) * EW, rows_in_tile * EW, lane);
        __syncthreads();
      }
    }
  };
  if (ortho) row_loop(std::true_type{}); else row_loop(std::false_type{});

  if (HAS_GT) {
    // both halves' shares of the pixel's sums, then the env mask (wrapperBRDFLight.py:172-174) and the wave's partials
    float v[3] = {s_pg, s_pp, s_g};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float d_ = v[i], s_ = v[i];
      swap32(d_, s_);
      v[i] = d_ + s_;
    }
    const float not_dark = (v[2] / (3.0f * (float)a.J)) > 0.001f ? 1.0f : 0.0f;
    const float m = (x.active && half == 0) ? seg_small_at(a, b, p) * a.env_ind[b] * not_dark : 0.0f;
    if (x.active && half == 0) (a.mask + (size_t)b * RC)[(unsigned)p] = m;
    float r0 = m * m * v[0], r1 = m * m * v[1], r2 = m;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      r0 += __shfl_xor(r0, off, 64); r1 += __shfl_xor(r1, off, 64); r2 += __shfl_xor(r2, off, 64);
    }
    if (lane == 0) {
      float* w = a.ws + (size_t)blockIdx.x * 3;
      w[0] = r0; w[1] = r1; w[2] = r2;
    }
  }
  if (DO_RENDER) {
    // each half integrated one half row: add the two
    float v[6] = {d0, d1, d2, s0, s1, s2};
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      float d_ = v[i], s_ = v[i];
      swap32(d_, s_);
      v[i] = d_ + s_;
    }
    if (x.active && half == 0) {
      const size_t o = (size_t)b * 3 * RC;
      const unsigned up = (unsigned)p;
      (a.diffuse + o)[up] = (alb[0] * kInvPi) * v[0];
      (a.diffuse + o + RC)[up] = (alb[1] * kInvPi) * v[1];
      (a.diffuse + o + 2 * (size_t)RC)[up] = (alb[2] * kInvPi) * v[2];
      (a.spec + o)[up] = v[3];
      (a.spec + o + RC)[up] = v[4];
      (a.spec + o + 2 * (size_t)RC)[up] = v[5];
    }
  }
}

// ============================== backward w.r.t. the SG parameters ================================
// g[c,j] = gEnv[c,j] (+) omega_j ndl_j (gD_c A_c/pi + gS_c spec_j);  per lobe, with T = (g . w) E:
//   dL/dw_c = sum g_c E,   dL/dlam = sum T t,   dL/da = lam (ca A, sa A, sum T c_e),  A_a = sum_e (+-s_e) T
// The env cotangent arrives one table row (EW directions) at a time by LDS-DMA, double-buffered for
// EW = 16 (2 x 12 KB): the next row's 12 DMA instructions are in flight while this row is consumed.
// More lobes than KP (config 5: 24): one workgroup per (pixel group, register group of KP lobes) instead of one workgroup
// walking the groups one after the other -- half as long work units (-12 % at config 5), and the workgroups of a pixel
// group sit 8 ids apart, i.e. are dispatched back to back to the SAME XCD, so that part of the second one's cotangent rows
// comes out of that XCD's L2 instead of HBM (PMC at config 5: 3.47 GB fetched for 2.34 GB algorithmic; the sequential form
// fetched 4.36 GB; two waves of ONE workgroup with a tile each measured worse: 2.88 vs 2.50 ms, 3.9 GB).
template <int KP, int POOL, int EW, bool HAS_GENV, bool HAS_RENDER>
__global__ __launch_bounds__(kWave, 2) void sg_bwd_fast_kernel(const Args a) {
  constexpr int TJ = EW;
  constexpr int HALF = EW / 2;
  constexpr int NP = HALF / 2;      // azimuth pairs per half row
  constexpr int NBUF = (EW == 16) ? 2 : 1;
  using D = DmaTile<TJ>;
  __shared__ __attribute__((aligned(16))) float tile[HAS_GENV ? NBUF * D::kFloats : 4];

  const int RC = a.R * a.C, K = a.K;
  // workgroup id -> (pixel group t, lobe group): ids [16 m, 16 m + 8) are lobe group 0 of pixel groups 8 m .. 8 m + 7,
  // ids [16 m + 8, 16 m + 16) lobe group 1 of the same pixel groups (for two groups; ng in general)
  const int ng = (K + KP - 1) / KP;
  const int chunk = (int)blockIdx.x / (8 * ng), within = (int)blockIdx.x - chunk * (8 * ng);
  const int grp = within >> 3, t = chunk * 8 + (within & 7);
  const int tiles = (RC + kWave - 1) / kWave;
  if (t >= a.bn * tiles) return;
  Pix x;
  x.lane = threadIdx.x;
  x.b = t / tiles;
  x.p0 = (t - x.b * tiles) * kWave;
  x.active = (x.p0 + x.lane) < RC;
  x.p = x.active ? (x.p0 + x.lane) : (RC - 1);
  const int lane = x.lane, b = x.b, p = x.p;

  PixLocal q;
  bool ortho = true;
  float gd0 = 0.f, gd1 = 0.f, gd2 = 0.f, gs0 = 0.f, gs1 = 0.f, gs2 = 0.f;
  if (HAS_RENDER) {
    float alb[3];
    const Frame f = load_frame<POOL>(a, x, alb);
    q = make_local(f, a.F0);
    ortho = __all(frame_is_orthonormal(q));
    const size_t o = (size_t)b * 3 * RC;
    const unsigned up = (unsigned)p;
    gd0 = (a.g_diffuse + o)[up] * (alb[0] * kInvPi);
    gd1 = (a.g_diffuse + o + RC)[up] * (alb[1] * kInvPi);
    gd2 = (a.g_diffuse + o + 2 * (size_t)RC)[up] * (alb[2] * kInvPi);
    gs0 = (a.g_spec + o)[up];
    gs1 = (a.g_spec + o + RC)[up];
    gs2 = (a.g_spec + o + 2 * (size_t)RC)[up];
  }
  const SepTable rows = as_sep_table(a.rows);
  const XTable cst = (XTable)(a.cols);                                         // [(EW/2)/2] x 2 x (ca, sa)
  const XTable xt = (XTable)(a.cols + EW);
  __amdgpu_buffer_rsrc_t gimg = env_rsrc(HAS_GENV ? a.g_env + (size_t)b * 3 * RC * a.J : a.view, RC, a.J);
  const int eh = a.eh;

  {
    const int kg = grp * KP;
    Lobes<KP> L;
    load_lobes<KP, false>(a, x, kg, L, false);
    float gax[KP], gay[KP], gaz[KP], glam[KP], gw0[KP], gw1[KP], gw2[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) gax[k] = gay[k] = gaz[k] = glam[k] = gw0[k] = gw1[k] = gw2[k] = 0.0f;

    if (HAS_GENV) tile_dma_issue<TJ>(tile, gimg, x.p0, RC, a.J, 0, lane);

    // the wave-uniform frame test is hoisted out of the direction loops: two copies of the row loop
    auto row_loop = [&](auto ortho_c) {
    for (int e = 0; e < eh; ++e) {
        const float* cur = tile + (NBUF == 2 ? (e & 1) * D::kFloats : 0);
        if (HAS_GENV) {
          if (NBUF == 2 && e + 1 < eh) {
            tile_dma_issue<TJ>(tile + ((e + 1) & 1) * D::kFloats, gimg, x.p0, RC, a.J, (e + 1) * EW, lane);
            wait_vmcnt<D::kInstr>();     // row e has landed; row e+1 stays in flight
          } else {
            wait_vmcnt<0>();
          }
        }
  #pragma unroll
        for (int k = 0; k < KP; ++k) { asm volatile("" : "+v"(L.ax[k])); asm volatile("" : "+v"(L.ay[k])); }   // no LICM of u_ka
        if (HAS_RENDER) fence_row_invariants(q);
        const f32x8 row = rows[e];
        const float sr = row[0], cr = row[1];
        const RowCtx rc = make_row_ctx(q, row, HAS_RENDER);
  #pragma unroll 1
        for (int ap = 0; ap < NP; ++ap) {
          const f32x4 cs = cst[ap];   // (ca, sa) of the pair's two azimuths
          float g[2][3][2];   // [sign][colour][azimuth in pair]
          if (HAS_GENV) {
            tile_dma_read_pairs<TJ>(cur, lane, ap * 2, HALF + ap * 2, g);
          } else {
  #pragma unroll
            for (int sg = 0; sg < 2; ++sg)
  #pragma unroll
              for (int c = 0; c < 3; ++c) g[sg][c][0] = g[sg][c][1] = 0.0f;
          }
          float ca[2], sa[2];
  #pragma unroll
          for (int i = 0; i < 2; ++i) {
            ca[i] = cs[2 * i]; sa[i] = cs[2 * i + 1];
            if (HAS_RENDER) {
  #pragma unroll
              for (int sg = 0; sg < 2; ++sg) {
                float wt, sp;
                shade_dir<decltype(ortho_c)::value>(q, rc, sg, ca[i], sa[i], xt, ap * 2 + i, wt, sp);
                g[sg][0][i] = fmaf(wt, fmaf(gs0, sp, gd0), g[sg][0][i]);
                g[sg][1][i] = fmaf(wt, fmaf(gs1, sp, gd1), g[sg][1][i]);
                g[sg][2][i] = fmaf(wt, fmaf(gs2, sp, gd2), g[sg][2][i]);
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
          for (int k = 0; k < KP; ++k) {
            const float czr = fmaf(L.az[k], cr, -1.0f);
  #pragma unroll
            for (int i = 0; i < 2; ++i) {
              const float u = fmaf(L.ay[k], sa[i], L.ax[k] * ca[i]);
              float A = 0.0f;
  #pragma unroll
              for (int sg = 0; sg < 2; ++sg) {
                const float ss = sg ? -sr : sr;
                const float t = fmaf(ss, u, czr);
                const float ex = fexp2(L.lp[k] * t);
                const float c0 = g[sg][0][i], c1 = g[sg][1][i], c2 = g[sg][2][i];
                gw0[k] = fmaf(c0, ex, gw0[k]);
                gw1[k] = fmaf(c1, ex, gw1[k]);
                gw2[k] = fmaf(c2, ex, gw2[k]);
                const float T = fmaf(c2, L.w2[k], fmaf(c1, L.w1[k], c0 * L.w0[k])) * ex;
                glam[k] = fmaf(T, t, glam[k]);
                A = fmaf(ss, T, A);
                gaz[k] = fmaf(cr, T, gaz[k]);
              }
              gax[k] = fmaf(ca[i], A, gax[k]);
              gay[k] = fmaf(sa[i], A, gay[k]);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (HAS_GENV && NBUF == 1 && e + 1 < eh) tile_dma_issue<TJ>(tile, gimg, x.p0, RC, a.J, (e + 1) * EW, lane);
      }
    };
    if (ortho) row_loop(std::true_type{}); else row_loop(std::false_type{});

    if (x.active) {
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        if (kg + k < K) {
          const size_t ab = ((size_t)(b * K + kg + k) * 3) * RC;
          const size_t lb = (size_t)(b * K + kg + k) * RC;
          const unsigned up = (unsigned)p;
          const float lam = L.lp[k] * kLn2;
          (a.g_axis + ab)[up] = lam * gax[k];
          (a.g_axis + ab + RC)[up] = lam * gay[k];
          (a.g_axis + ab + 2 * (size_t)RC)[up] = lam * gaz[k];
          float gl = glam[k], q0 = gw0[k], q1 = gw1[k], q2 = gw2[k];
          if (a.premap) {
            gl *= premap_grad(lam);
            q0 *= premap_grad(L.w0[k]); q1 *= premap_grad(L.w1[k]); q2 *= premap_grad(L.w2[k]);
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

// ============================== backward, lobes split over two waves ==============================
// Same math as sg_bwd_fast_kernel, but the 12 lobes of a 64-pixel group are split over the two waves of
// a 128-thread workgroup (6 each): ~140 VGPRs per wave instead of ~250, i.e. 3 waves/SIMD instead of 2,
// with the 24 KB double-buffered cotangent tile shared by both waves.  Per table row:
//   1. each wave issues its half of the next row's DMA, waits for its half of this row (counted vmcnt),
//      barrier;
//   2. the quadrature's contribution to the cotangent is evaluated once -- each wave does half of the
//      row's directions -- and added in place into the LDS tile; barrier;
//   3. each wave consumes all 16 directions of the row for its own lobes; barrier (buffer reuse).
template <int KPW, int POOL, bool HAS_GENV, bool HAS_RENDER>
__global__ __launch_bounds__(2 * kWave, 3) void sg_bwd_split_kernel(const Args a) {
  constexpr int EW = 16, TJ = 16, HALF = 8, NP = 4;
  using D = DmaTile<TJ>;
  __shared__ __attribute__((aligned(16))) float tile[(HAS_GENV ? 2 : 1) * D::kFloats];

  const int wave = threadIdx.x >> 6;
  Pix x;
  x.lane = threadIdx.x & 63;
  const int RC = a.R * a.C, K = a.K;
  {
    const int tiles = (RC + kWave - 1) / kWave;
    x.b = blockIdx.x / tiles;
    x.p0 = (blockIdx.x - x.b * tiles) * kWave;
    x.active = (x.p0 + x.lane) < RC;
    x.p = x.active ? (x.p0 + x.lane) : (RC - 1);
  }
  const int lane = x.lane, b = x.b, p = x.p;

  PixLocal q;
  bool ortho = true;
  float gd0 = 0.f, gd1 = 0.f, gd2 = 0.f, gs0 = 0.f, gs1 = 0.f, gs2 = 0.f;
  if (HAS_RENDER) {
    float alb[3];
    const Frame f = load_frame<POOL>(a, x, alb);
    q = make_local(f, a.F0);
    ortho = __all(frame_is_orthonormal(q));
    const size_t o = (size_t)b * 3 * RC;
    const unsigned up = (unsigned)p;
    gd0 = (a.g_diffuse + o)[up] * (alb[0] * kInvPi);
    gd1 = (a.g_diffuse + o + RC)[up] * (alb[1] * kInvPi);
    gd2 = (a.g_diffuse + o + 2 * (size_t)RC)[up] * (alb[2] * kInvPi);
    gs0 = (a.g_spec + o)[up];
    gs1 = (a.g_spec + o + RC)[up];
    gs2 = (a.g_spec + o + 2 * (size_t)RC)[up];
  }
  const SepTable rows = as_sep_table(a.rows);
  const XTable cst = (XTable)(a.cols);
  const XTable xt = (XTable)(a.cols + EW);
  __amdgpu_buffer_rsrc_t gimg = env_rsrc(HAS_GENV ? a.g_env + (size_t)b * 3 * RC * a.J : a.view, RC, a.J);
  const int eh = a.eh;

  for (int kg = 0; kg < K; kg += 2 * KPW) {
    Lobes<KPW> L;
    load_lobes<KPW, false>(a, x, kg + wave * KPW, L, false);
    float gax[KPW], gay[KPW], gaz[KPW], glam[KPW], gw0[KPW], gw1[KPW], gw2[KPW];
#pragma unroll
    for (int k = 0; k < KPW; ++k) gax[k] = gay[k] = gaz[k] = glam[k] = gw0[k] = gw1[k] = gw2[k] = 0.0f;

    if (HAS_GENV) tile_dma_issue_part<TJ>(tile, gimg, x.p0, RC, a.J, 0, lane, wave, 2);

    // the wave-uniform frame test is hoisted out of the direction loops: two copies of the row loop
    auto row_loop = [&](auto ortho_c) {
    for (int e = 0; e < eh; ++e) {
        float* cur = tile + (HAS_GENV ? (e & 1) * D::kFloats : 0);
        if (HAS_GENV) {
          if (e + 1 < eh) {
            tile_dma_issue_part<TJ>(tile + ((e + 1) & 1) * D::kFloats, gimg, x.p0, RC, a.J, (e + 1) * EW, lane, wave, 2);
            wait_vmcnt<6>();        // this wave's half of row e has landed; its half of row e+1 stays in flight
          } else {
            wait_vmcnt<0>();
          }
          barrier_lds_only();       // ... and so has the other wave's half
        }
  #pragma unroll
        for (int k = 0; k < KPW; ++k) { asm volatile("" : "+v"(L.ax[k])); asm volatile("" : "+v"(L.ay[k])); }   // no LICM of u_ka
        if (HAS_RENDER) fence_row_invariants(q);
        const f32x8 row = rows[e];
        const float sr = row[0], cr = row[1];
        const RowCtx rc = make_row_ctx(q, row, HAS_RENDER);
  
        // ---- 2. quadrature contribution, this wave's half of the directions, added in place ------------
        if (HAS_RENDER) {
  #pragma unroll 1
          for (int h = 0; h < 2; ++h) {
            const int ap = wave * 2 + h;
            const f32x4 cs = cst[ap];
            float g[2][3][2];
            if (HAS_GENV) {
              tile_dma_read_pairs<TJ>(cur, lane, ap * 2, HALF + ap * 2, g);
            } else {
  #pragma unroll
              for (int sg = 0; sg < 2; ++sg)
  #pragma unroll
                for (int c = 0; c < 3; ++c) g[sg][c][0] = g[sg][c][1] = 0.0f;
            }
  #pragma unroll
            for (int i = 0; i < 2; ++i) {
  #pragma unroll
              for (int sg = 0; sg < 2; ++sg) {
                float wt, sp;
                shade_dir<decltype(ortho_c)::value>(q, rc, sg, cs[2 * i], cs[2 * i + 1], xt, ap * 2 + i, wt, sp);
                g[sg][0][i] = fmaf(wt, fmaf(gs0, sp, gd0), g[sg][0][i]);
                g[sg][1][i] = fmaf(wt, fmaf(gs1, sp, gd1), g[sg][1][i]);
                g[sg][2][i] = fmaf(wt, fmaf(gs2, sp, gd2), g[sg][2][i]);
              }
            }
            tile_dma_write_pairs<TJ>(cur, lane, ap * 2, HALF + ap * 2, g);
          }
          barrier_lds_only();
        }
  
        // ---- 3. all directions of the row, this wave's lobes ---------------------------------------------
  #pragma unroll 1
        for (int ap = 0; ap < NP; ++ap) {
          const f32x4 cs = cst[ap];
          float g[2][3][2];
          tile_dma_read_pairs<TJ>(cur, lane, ap * 2, HALF + ap * 2, g);
          float ca[2] = {cs[0], cs[2]}, sa[2] = {cs[1], cs[3]};
  #pragma unroll
          for (int k = 0; k < KPW; ++k) {
            const float czr = fmaf(L.az[k], cr, -1.0f);
  #pragma unroll
            for (int i = 0; i < 2; ++i) {
              const float u = fmaf(L.ay[k], sa[i], L.ax[k] * ca[i]);
              float A = 0.0f;
  #pragma unroll
              for (int sg = 0; sg < 2; ++sg) {
                const float ss = sg ? -sr : sr;
                const float t = fmaf(ss, u, czr);
                const float ex = fexp2(L.lp[k] * t);
                const float c0 = g[sg][0][i], c1 = g[sg][1][i], c2 = g[sg][2][i];
                gw0[k] = fmaf(c0, ex, gw0[k]);
                gw1[k] = fmaf(c1, ex, gw1[k]);
                gw2[k] = fmaf(c2, ex, gw2[k]);
                const float T = fmaf(c2, L.w2[k], fmaf(c1, L.w1[k], c0 * L.w0[k])) * ex;
                glam[k] = fmaf(T, t, glam[k]);
                A = fmaf(ss, T, A);
                gaz[k] = fmaf(cr, T, gaz[k]);
              }
              gax[k] = fmaf(ca[i], A, gax[k]);
              gay[k] = fmaf(sa[i], A, gay[k]);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        barrier_lds_only();   // both waves are done with `cur` before it is refilled / rewritten
      }
    };
    if (ortho) row_loop(std::true_type{}); else row_loop(std::false_type{});

    if (x.active) {
#pragma unroll
      for (int k = 0; k < KPW; ++k) {
        const int kk = kg + wave * KPW + k;
        if (kk < K) {
          const size_t ab = ((size_t)(b * K + kk) * 3) * RC;
          const size_t lb = (size_t)(b * K + kk) * RC;
          const unsigned up = (unsigned)p;
          const float lam = L.lp[k] * kLn2;
          (a.g_axis + ab)[up] = lam * gax[k];
          (a.g_axis + ab + RC)[up] = lam * gay[k];
          (a.g_axis + ab + 2 * (size_t)RC)[up] = lam * gaz[k];
          float gl = glam[k], q0 = gw0[k], q1 = gw1[k], q2 = gw2[k];
          if (a.premap) {
            gl *= premap_grad(lam);
            q0 *= premap_grad(L.w0[k]); q1 *= premap_grad(L.w1[k]); q2 *= premap_grad(L.w2[k]);
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

// ============================== backward, half-wave lobe split ====================================
// One wave = 32 pixels x 2 lobe groups: lanes l and l+32 own the same pixel, lanes 0..31 hold lobes 0..5 and
// lanes 32..63 lobes 6..11 (12 lobes + 12 gradient sets per lane do not fit the register file).  What the two
// halves have to share -- the render term of the cotangent, of which each half evaluates the BRDF for one
// half row -- is traded with v_permlane32_swap_b32 (gfx950: upper 32 lanes of one VGPR <-> lower 32 of
// another): swap(D = r, S = r) leaves lanes 0..31's value in D and lanes 32..63's in S, in all lanes.
// No LDS exchange, no barrier; the env cotangent rows arrive by LDS-DMA, 32 pixels x 16 directions at a time.
template <int POOL, bool HAS_GENV, bool HAS_RENDER, int OCC>
__global__ __launch_bounds__(kWave, OCC) void sg_bwd_half_kernel(const Args a) {
  constexpr int EW = 16, HALF = 8, NP = 4, KPW = 6;
  __shared__ __attribute__((aligned(16))) float tile[HAS_GENV ? 2 * kT32Floats : 4];

  const int lane = threadIdx.x, half = lane >> 5, pl = lane & 31;
  const int own = 1 - half;                         // the half row (sign) whose BRDF terms this half-wave evaluates
  const int RC = a.R * a.C, K = a.K;
  Pix x;
  x.lane = lane;
  {
    const int tiles = (RC + kPx - 1) / kPx;
    x.b = blockIdx.x / tiles;
    x.p0 = (blockIdx.x - x.b * tiles) * kPx;
    x.active = (x.p0 + pl) < RC;
    x.p = x.active ? (x.p0 + pl) : (RC - 1);
  }
  const int b = x.b, p = x.p;

  __amdgpu_buffer_rsrc_t gimg = env_rsrc(HAS_GENV ? a.g_env + (size_t)b * 3 * RC * a.J : a.view, RC, a.J);
  if (HAS_GENV) tile32_dma_issue(tile, gimg, x.p0, RC, a.J, 0, lane);

  PixLocal q;
  bool ortho = true;
  float gd0 = 0.f, gd1 = 0.f, gd2 = 0.f, gs0 = 0.f, gs1 = 0.f, gs2 = 0.f;
  if (HAS_RENDER) {
    float alb[3];
    const Frame f = load_frame<POOL>(a, x, alb);
    q = make_local(f, a.F0);
    ortho = __all(frame_is_orthonormal(q));
    const size_t o = (size_t)b * 3 * RC;
    const unsigned up = (unsigned)p;
    gd0 = (a.g_diffuse + o)[up] * (alb[0] * kInvPi);
    gd1 = (a.g_diffuse + o + RC)[up] * (alb[1] * kInvPi);
    gd2 = (a.g_diffuse + o + 2 * (size_t)RC)[up] * (alb[2] * kInvPi);
    gs0 = (a.g_spec + o)[up];
    gs1 = (a.g_spec + o + RC)[up];
    gs2 = (a.g_spec + o + 2 * (size_t)RC)[up];
  }

  // this half's lobes: per-lane offsets into the image's SG block (the lobe index differs between the halves)
  Lobes<KPW> L;
  {
    const float* axis_b = a.axis + (size_t)b * K * 3 * RC;
    const float* lamb_b = a.lamb + (size_t)b * K * RC;
    const float* weight_b = a.weight + (size_t)b * K * 3 * RC;
#pragma unroll
    for (int k = 0; k < KPW; ++k) {
      const int kc = min(half * KPW + k, K - 1);
      const unsigned o3 = (unsigned)(kc * 3 * RC + p), o1 = (unsigned)(kc * RC + p);
      L.ax[k] = axis_b[o3]; L.ay[k] = axis_b[o3 + RC]; L.az[k] = axis_b[o3 + 2 * RC];
      L.lp[k] = lamb_b[o1];
      L.w0[k] = weight_b[o3]; L.w1[k] = weight_b[o3 + RC]; L.w2[k] = weight_b[o3 + 2 * RC];
    }
#pragma unroll
    for (int k = 0; k < KPW; ++k) {
      const bool live = half * KPW + k < K;
      float l = L.lp[k], t0 = L.w0[k], t1 = L.w1[k], t2 = L.w2[k];
      if (a.premap == 1) { l = premap(l); t0 = premap(t0); t1 = premap(t1); t2 = premap(t2); }
      L.lp[k] = l * kLog2e;
      L.w0[k] = live ? t0 : 0.0f; L.w1[k] = live ? t1 : 0.0f; L.w2[k] = live ? t2 : 0.0f;
    }
  }
  float gax[KPW], gay[KPW], gaz[KPW], glam[KPW], gw0[KPW], gw1[KPW], gw2[KPW];
#pragma unroll
  for (int k = 0; k < KPW; ++k) gax[k] = gay[k] = gaz[k] = glam[k] = gw0[k] = gw1[k] = gw2[k] = 0.0f;

  const SepTable rows = as_sep_table(a.rows);
  const XTable cst = (XTable)(a.cols);
  const XTable xt = (XTable)(a.cols + EW);
  const int eh = a.eh;

#if SGR_TABLE_PREFETCH
  f32x8 row_next = rows[0];
#endif
  auto row_loop = [&](auto ortho_c) {
    for (int e = 0; e < eh; ++e) {
      const float* cur = tile + (HAS_GENV ? (e & 1) * kT32Floats : 0);
      if (HAS_GENV) {
        if (e + 1 < eh) {
          tile32_dma_issue(tile + ((e + 1) & 1) * kT32Floats, gimg, x.p0, RC, a.J, (e + 1) * EW, lane);
          wait_vmcnt<6>();        // row e has landed; row e+1 stays in flight
        } else {
          wait_vmcnt<0>();
        }
      }
#pragma unroll
      for (int k = 0; k < KPW; ++k) { asm volatile("" : "+v"(L.ax[k])); asm volatile("" : "+v"(L.ay[k])); }   // no LICM of u_ka
      if (HAS_RENDER) fence_row_invariants(q);
#if SGR_TABLE_PREFETCH
      const f32x8 row = row_next;
      {
        int ne = e + 1 < eh ? e + 1 : e;
        asm volatile("" : "+s"(ne) : "s"(row));
        row_next = rows[ne];
      }
      f32x4 cs_next = cst[0];      // the next azimuth pair's scalar table entry is requested one iteration ahead
      __builtin_amdgcn_sched_barrier(0);
#else
      const f32x8 row = rows[e];
#endif
      const float sr = row[0], cr = row[1];
      const RowCtx rc = make_row_ctx(q, row, HAS_RENDER);

#pragma unroll 1
      for (int ap = 0; ap < NP; ++ap) {
#if SGR_TABLE_PREFETCH
        const f32x4 cs = cs_next;
#else
        const f32x4 cs = cst[ap];
#endif
        const float ca[2] = {cs[0], cs[2]}, sa[2] = {cs[1], cs[3]};
        float g[2][3][2];
        if (HAS_GENV) {
          float t0[3][2], t1[3][2];
          tile32_read_two_pairs(cur, pl, ap * 2, HALF + ap * 2, t0, t1);
#pragma unroll
          for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < 2; ++i) { g[0][c][i] = t0[c][i]; g[1][c][i] = t1[c][i]; }
        } else {
#pragma unroll
          for (int sg = 0; sg < 2; ++sg)
#pragma unroll
            for (int c = 0; c < 3; ++c) g[sg][c][0] = g[sg][c][1] = 0.0f;
        }
#if SGR_TABLE_PREFETCH
        {   // requested after the LDS reads (scalar loads and LDS share a counter), consumed an iteration later
          int nxt = (ap + 1) & (NP - 1);
          asm volatile("" : "+s"(nxt) : "s"(cs));  // `cs` is waited for here, not (together with the new request) at its first use
          cs_next = cst[nxt];
          __builtin_amdgcn_sched_barrier(0);      // ... and not sunk towards its use by the scheduler
        }
#endif
        if (HAS_RENDER) {
          // the render term of the half row this half-wave owns, then both halves' terms to all lanes
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            float wt, sp;
            shade_dir<decltype(ortho_c)::value>(q, rc, own, ca[i], sa[i], xt, ap * 2 + i, wt, sp);
            float r_[3] = {wt * fmaf(gs0, sp, gd0), wt * fmaf(gs1, sp, gd1), wt * fmaf(gs2, sp, gd2)};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              float d_ = r_[c], s_ = r_[c];
              swap32(d_, s_);
              g[1][c][i] += d_;     // evaluated by lanes 0..31
              g[0][c][i] += s_;     // evaluated by lanes 32..63
            }
          }
        }
#pragma unroll
        for (int k = 0; k < KPW; ++k) {
          const float czr = fmaf(L.az[k], cr, -1.0f);
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const float u = fmaf(L.ay[k], sa[i], L.ax[k] * ca[i]);
            float A = 0.0f;
#pragma unroll
            for (int sg = 0; sg < 2; ++sg) {
              const float ss = sg ? -sr : sr;
              const float t = fmaf(ss, u, czr);
              const float ex = fexp2(L.lp[k] * t);
              const float c0 = g[sg][0][i], c1 = g[sg][1][i], c2 = g[sg][2][i];
              gw0[k] = fmaf(c0, ex, gw0[k]);
              gw1[k] = fmaf(c1, ex, gw1[k]);
              gw2[k] = fmaf(c2, ex, gw2[k]);
              const float T = fmaf(c2, L.w2[k], fmaf(c1, L.w1[k], c0 * L.w0[k])) * ex;
              glam[k] = fmaf(T, t, glam[k]);
              A = fmaf(ss, T, A);
              gaz[k] = fmaf(cr, T, gaz[k]);
            }
            gax[k] = fmaf(ca[i], A, gax[k]);
            gay[k] = fmaf(sa[i], A, gay[k]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  if (ortho) row_loop(std::true_type{}); else row_loop(std::false_type{});

  if (x.active) {
    float* g_axis_b = a.g_axis + (size_t)b * K * 3 * RC;
    float* g_lamb_b = a.g_lamb + (size_t)b * K * RC;
    float* g_weight_b = a.g_weight + (size_t)b * K * 3 * RC;
#pragma unroll
    for (int k = 0; k < KPW; ++k) {
      const int kk = half * KPW + k;
      if (kk < K) {
        const unsigned o3 = (unsigned)(kk * 3 * RC + p), o1 = (unsigned)(kk * RC + p);
        const float lam = L.lp[k] * kLn2;
        g_axis_b[o3] = lam * gax[k];
        g_axis_b[o3 + RC] = lam * gay[k];
        g_axis_b[o3 + 2 * RC] = lam * gaz[k];
        float gl = glam[k], q0 = gw0[k], q1 = gw1[k], q2 = gw2[k];
        if (a.premap) {
          gl *= premap_grad(lam);
          q0 *= premap_grad(L.w0[k]); q1 *= premap_grad(L.w1[k]); q2 *= premap_grad(L.w2[k]);
        }
        g_lamb_b[o1] = gl;
        g_weight_b[o3] = q0;
        g_weight_b[o3 + RC] = q1;
        g_weight_b[o3 + 2 * RC] = q2;
      }
    }
  }
}

// ============================== forwardEnv alone (env image read) =================================
// The un-fused drop-in call renderingLayer.forwardEnv (models.py:461-522): HBM-bound (1672 B/px in,
// ~45 VALU slots per direction), so the env rows stream in by double-buffered LDS-DMA exactly like the
// cotangent rows of the backward pass.
template <int POOL, int EW>
__global__ __launch_bounds__(kWave, 4) void render_fast_kernel(const Args a) {
  constexpr int TJ = EW;
  constexpr int HALF = EW / 2;
  constexpr int NP = HALF / 2;
  constexpr int NBUF = (EW == 16) ? 2 : 1;
  using D = DmaTile<TJ>;
  __shared__ __attribute__((aligned(16))) float tile[NBUF * D::kFloats];

  const Pix x = locate(a);
  const int lane = x.lane, b = x.b, p = x.p;
  const int RC = a.R * a.C;
  float alb[3];
  const Frame f = load_frame<POOL>(a, x, alb);
  PixLocal q = make_local(f, a.F0);
  const bool ortho = __all(frame_is_orthonormal(q));
  const SepTable rows = as_sep_table(a.rows);
  const XTable cst = (XTable)(a.cols);
  const XTable xt = (XTable)(a.cols + EW);
  __amdgpu_buffer_rsrc_t eimg = env_rsrc(a.env_in + (size_t)b * 3 * RC * a.J, RC, a.J);
  const int eh = a.eh;
  float d0 = 0.f, d1 = 0.f, d2 = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f;

  tile_dma_issue<TJ>(tile, eimg, x.p0, RC, a.J, 0, lane);
  // the wave-uniform frame test is hoisted out of the direction loops: two copies of the row loop
  auto row_loop = [&](auto ortho_c) {
  for (int e = 0; e < eh; ++e) {
      const float* cur = tile + (NBUF == 2 ? (e & 1) * D::kFloats : 0);
      if (NBUF == 2 && e + 1 < eh) {
        tile_dma_issue<TJ>(tile + ((e + 1) & 1) * D::kFloats, eimg, x.p0, RC, a.J, (e + 1) * EW, lane);
        wait_vmcnt<D::kInstr>();
      } else {
        wait_vmcnt<0>();
      }
      fence_row_invariants(q);
      const RowCtx rc = make_row_ctx(q, rows[e], true);
  #pragma unroll 1
      for (int ap = 0; ap < NP; ++ap) {
        const f32x4 cs = cst[ap];
        float g[2][3][2];
        tile_dma_read_pairs<TJ>(cur, lane, ap * 2, HALF + ap * 2, g);
  #pragma unroll
        for (int i = 0; i < 2; ++i) {
  #pragma unroll
          for (int sg = 0; sg < 2; ++sg) {
            float wt, sp;
            shade_dir<decltype(ortho_c)::value>(q, rc, sg, cs[2 * i], cs[2 * i + 1], xt, ap * 2 + i, wt, sp);
            const float sw = sp * wt;
            d0 = fmaf(wt, g[sg][0][i], d0);
            d1 = fmaf(wt, g[sg][1][i], d1);
            d2 = fmaf(wt, g[sg][2][i], d2);
            s0 = fmaf(sw, g[sg][0][i], s0);
            s1 = fmaf(sw, g[sg][1][i], s1);
            s2 = fmaf(sw, g[sg][2][i], s2);
          }
        }
      }
      if (NBUF == 1 && e + 1 < eh) tile_dma_issue<TJ>(tile, eimg, x.p0, RC, a.J, (e + 1) * EW, lane);
    }
  };
  if (ortho) row_loop(std::true_type{}); else row_loop(std::false_type{});
  if (x.active) {
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
