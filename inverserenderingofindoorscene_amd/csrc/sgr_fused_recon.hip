// The trainLight objective without the env image on gfx950.
//
// wrapperBRDFLight.py:172-207 needs the predicted env image twice: for the log-L2 reconstruction loss
// against the ground-truth env (:172-188, with the models.LSregress scale, models.py:7-21) and for the
// render (:194).  Both are reductions over the image, so neither it nor its cotangent has to exist in
// memory (1536 B per shaded pixel each way at 8x16 directions):
//
//   sgr_fused_fwd_recon   one pass of the fused forward kernel (sgr_fast.inl, HAS_GT): render + per-pixel
//                         <pred, gt>, <pred, pred>, sum gt with the ground-truth rows arriving by LDS-DMA
//                         -> env mask, per-image LSregress scale (deterministic fold).
//   sgr_fused_bwd_recon   one pass that recomputes the lobes' exponentials, forms the predicted radiance of
//                         four directions at a time, evaluates the reconstruction loss AND its cotangent
//                         there and then, adds the render cotangent, and accumulates the SG gradients.
//
// Backward work decomposition (same as sg_bwd_split_kernel): a 128-thread workgroup = two waves over the
// same 64 pixels, wave w owning lobes [6w, 6w+6).  The radiance of a direction needs all 12 lobes, so per
// chunk of 4 directions (two azimuths x both half-rows) each wave publishes its 6-lobe partial radiance (12
// floats) and the BRDF terms of the two directions it shaded (4 floats) in LDS; one barrier per chunk
// (exchange buffers alternate), after which both waves hold the full radiance and cotangent.
#include "sgr_forward.inl"
#include "sgr_recon_fold.h"

namespace sgr {

// exchange area: [buf][wave][16 values][64 lanes]; inline-asm LDS ops for the same reason as
// tile_dma_read_pairs (a ds_read the compiler can see drains all outstanding LDS-DMA first)
__device__ __forceinline__ void xch_write(unsigned addr, const float (&v)[16]) {
#pragma unroll
  for (int i = 0; i < 16; ++i) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(addr), "v"(v[i]), "n"(i * 256) : "memory");
}
__device__ __forceinline__ void xch_read(unsigned addr, float (&v)[16]) {
#pragma unroll
  for (int i = 0; i < 16; ++i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[i]) : "v"(addr), "n"(i * 256) : "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

template <int POOL>
__global__ __launch_bounds__(2 * kWave, 2) void sg_bwd_recon_kernel(const Args a) {
  constexpr int EW = 16, TJ = 16, HALF = 8, NP = 4, KPW = 6;
  using D = DmaTile<TJ>;
  __shared__ __attribute__((aligned(16))) float tile[2 * D::kFloats];          // ground-truth rows, double-buffered
  __shared__ __attribute__((aligned(16))) float xch[2 * 2 * 16 * kWave];       // partial radiance + BRDF terms

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

  float alb[3];
  const Frame f = load_frame<POOL>(a, x, alb);
  PixLocal q = make_local(f, a.F0);
  const bool ortho = __all(frame_is_orthonormal(q));
  float gd0, gd1, gd2, gs0, gs1, gs2;
  {
    const size_t o = (size_t)b * 3 * RC;
    const unsigned up = (unsigned)p;
    gd0 = (a.g_diffuse + o)[up] * (alb[0] * kInvPi);
    gd1 = (a.g_diffuse + o + RC)[up] * (alb[1] * kInvPi);
    gd2 = (a.g_diffuse + o + 2 * (size_t)RC)[up] * (alb[2] * kInvPi);
    gs0 = (a.g_spec + o)[up];
    gs1 = (a.g_spec + o + RC)[up];
    gs2 = (a.g_spec + o + 2 * (size_t)RC)[up];
  }
  // reconstruction side: x = cf p + off;  dnum/dp = 2 m (ln x - ln(gt + off)) cf / x        (coef is a constant)
  const float cf = a.coef[b], off = a.offset;
  const float m = x.active ? (a.mask_in + (size_t)b * RC)[(unsigned)p] : 0.0f;
  const float grec = 2.0f * m * a.rec_scale[0] * cf * kLn2;     // times dl (in log2 units) / x
  float loss = 0.0f;

  const SepTable rows = as_sep_table(a.rows);
  const XTable cst = (XTable)(a.cols);
  const XTable xt = (XTable)(a.cols + EW);
  __amdgpu_buffer_rsrc_t gimg = env_rsrc(a.env_gt + (size_t)b * 3 * RC * a.J, RC, a.J);
  const int eh = a.eh;
  const unsigned xmine = lds_addr(xch) + (unsigned)((wave * 16 * kWave + lane) * 4);
  const unsigned xother = lds_addr(xch) + (unsigned)(((wave ^ 1) * 16 * kWave + lane) * 4);
  constexpr unsigned kBufBytes = 2 * 16 * kWave * 4;

  Lobes<KPW> L;
  load_lobes<KPW, false>(a, x, wave * KPW, L, false);
  float gax[KPW], gay[KPW], gaz[KPW], glam[KPW], gw0[KPW], gw1[KPW], gw2[KPW];
#pragma unroll
  for (int k = 0; k < KPW; ++k) gax[k] = gay[k] = gaz[k] = glam[k] = gw0[k] = gw1[k] = gw2[k] = 0.0f;

  tile_dma_issue_part<TJ>(tile, gimg, x.p0, RC, a.J, 0, lane, wave, 2);

  auto row_loop = [&](auto ortho_c) {
    unsigned par = 0;
    for (int e = 0; e < eh; ++e) {
      const float* cur = tile + (e & 1) * D::kFloats;
      if (e + 1 < eh) {
        tile_dma_issue_part<TJ>(tile + ((e + 1) & 1) * D::kFloats, gimg, x.p0, RC, a.J, (e + 1) * EW, lane, wave, 2);
        wait_vmcnt<6>();        // this wave's half of row e has landed; its half of row e+1 stays in flight
      } else {
        wait_vmcnt<0>();
      }
      barrier_lds_only();       // ... and so has the other wave's half
#pragma unroll
      for (int k = 0; k < KPW; ++k) { asm volatile("" : "+v"(L.ax[k])); asm volatile("" : "+v"(L.ay[k])); }   // no LICM of u_ka
      fence_row_invariants(q);
      const f32x8 row = rows[e];
      const float sr = row[0], cr = row[1];
      const RowCtx rc = make_row_ctx(q, row, true);

#pragma unroll 1
      for (int ap = 0; ap < NP; ++ap) {
        const f32x4 cs = cst[ap];
        const float ca[2] = {cs[0], cs[2]}, sa[2] = {cs[1], cs[3]};
        // ---- 1. this wave's lobes: exponentials and partial radiance of the 4 directions ---------------
        float ex[KPW][2][2], u[KPW][2];
        float v[16];
#pragma unroll
        for (int i = 0; i < 12; ++i) v[i] = 0.0f;
#pragma unroll
        for (int k = 0; k < KPW; ++k) {
          const float czr = fmaf(L.az[k], cr, -1.0f);
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            u[k][i] = fmaf(L.ay[k], sa[i], L.ax[k] * ca[i]);
#pragma unroll
            for (int sg = 0; sg < 2; ++sg) {
              const float t = fmaf(sg ? -sr : sr, u[k][i], czr);
              const float e_ = fexp2(L.lp[k] * t);
              ex[k][i][sg] = e_;
              v[(sg * 3 + 0) * 2 + i] = fmaf(L.w0[k], e_, v[(sg * 3 + 0) * 2 + i]);
              v[(sg * 3 + 1) * 2 + i] = fmaf(L.w1[k], e_, v[(sg * 3 + 1) * 2 + i]);
              v[(sg * 3 + 2) * 2 + i] = fmaf(L.w2[k], e_, v[(sg * 3 + 2) * 2 + i]);
            }
          }
        }
        // ---- 2. BRDF terms of the two directions of half-row `wave`; publish; fetch the other wave's -----
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float wt, sp;
          shade_dir<decltype(ortho_c)::value>(q, rc, wave, ca[i], sa[i], xt, ap * 2 + i, wt, sp);
          v[12 + 2 * i] = wt;
          v[13 + 2 * i] = sp;
        }
        xch_write(xmine + par * kBufBytes, v);
        barrier_lds_only();
        float o[16];
        xch_read(xother + par * kBufBytes, o);
        par ^= 1u;
        float wts[2][2], sps[2][2];      // [sg][i]
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          wts[0][i] = wave ? o[12 + 2 * i] : v[12 + 2 * i];
          sps[0][i] = wave ? o[13 + 2 * i] : v[13 + 2 * i];
          wts[1][i] = wave ? v[12 + 2 * i] : o[12 + 2 * i];
          sps[1][i] = wave ? v[13 + 2 * i] : o[13 + 2 * i];
        }
        // ---- 3. cotangent of the radiance: reconstruction term (and loss) + render term -----------------
        float gt[2][3][2];
        tile_dma_read_pairs<TJ>(cur, lane, ap * 2, HALF + ap * 2, gt);
        float g[2][3][2];
#pragma unroll
        for (int sg = 0; sg < 2; ++sg)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const float wt = wts[sg][i], sp = sps[sg][i];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const float pf = v[(sg * 3 + c) * 2 + i] + o[(sg * 3 + c) * 2 + i];
              const float xx = fmaf(cf, pf, off);
              const float r = __builtin_amdgcn_rcpf(xx);
              const float dl = -__builtin_amdgcn_logf((gt[sg][c][i] + off) * r);   // log2(x / (gt + off))
              loss = fmaf(dl, dl, loss);
              const float gr = c == 0 ? fmaf(gs0, sp, gd0) : (c == 1 ? fmaf(gs1, sp, gd1) : fmaf(gs2, sp, gd2));
              g[sg][c][i] = fmaf(grec * dl, r, wt * gr);
            }
          }
        // ---- 4. this wave's lobes: gradient accumulation ------------------------------------------------
#pragma unroll
        for (int k = 0; k < KPW; ++k) {
          const float czr = fmaf(L.az[k], cr, -1.0f);
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            float A = 0.0f;
#pragma unroll
            for (int sg = 0; sg < 2; ++sg) {
              const float ss = sg ? -sr : sr;
              const float t = fmaf(ss, u[k][i], czr);
              const float e_ = ex[k][i][sg];
              const float c0 = g[sg][0][i], c1 = g[sg][1][i], c2 = g[sg][2][i];
              gw0[k] = fmaf(c0, e_, gw0[k]);
              gw1[k] = fmaf(c1, e_, gw1[k]);
              gw2[k] = fmaf(c2, e_, gw2[k]);
              const float T = fmaf(c2, L.w2[k], fmaf(c1, L.w1[k], c0 * L.w0[k])) * e_;
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
      barrier_lds_only();   // both waves are done with `cur` before it is refilled
    }
  };
  if (ortho) row_loop(std::true_type{}); else row_loop(std::false_type{});

  // loss partial of the tile: sum_p m_p sum_{c,j} (ln x - ln(gt+off))^2   (both waves hold it; wave 0 writes)
  {
    float r0 = m * loss * (kLn2 * kLn2);
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) r0 += __shfl_xor(r0, s, 64);
    if (wave == 0 && lane == 0) a.ws[blockIdx.x] = r0;
  }

  if (x.active) {
#pragma unroll
    for (int k = 0; k < KPW; ++k) {
      const int kk = wave * KPW + k;
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

// d objective / d num = rec_weight / (3 J max(den, 1e-5)), with den = sum of the env mask over the (global) batch
__global__ void recon_scale_kernel(const float* __restrict__ den_img, const float* __restrict__ den_global, float* __restrict__ scale,
                                   int bn, float rec_weight_over_3J) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float den;
    if (den_global) {
      den = den_global[0];
    } else {
      double s = 0.0;
      for (int i = 0; i < bn; ++i) s += (double)den_img[i];
      den = (float)s;
    }
    scale[0] = rec_weight_over_3J / fmaxf(den, 1e-5f);
  }
}

}  // namespace sgr

using namespace sgr;

static int recon_tiles(int RC) { return (RC + kWave - 1) / kWave; }
static bool fused_recon_ok(int K, int R, int C, int eh, int ew) {
  const long long env_bytes = 3LL * R * C * eh * ew * 4;
  return ew == 16 && K <= 12 && K >= 1 && env_bytes < (1LL << 31);
}

extern "C" int sgr_fused_recon_supported(int K, int R, int C, int eh, int ew) { return fused_recon_ok(K, R, C, eh, ew) ? 1 : 0; }

// workspace: [bn,tiles,3] forward partials | [bn,tiles] loss partials | [bn] per-image mask sums | [1] scale
extern "C" int sgr_fused_recon_workspace_floats(int bn, int R, int C) { return bn * recon_tiles(R * C) * 4 + bn + 4; }

extern "C" int sgr_fused_fwd_recon(const float* albedo, const float* normal, const float* rough, const float* axis, const float* lamb,
                                   const float* weight, const float* dirs, const float* view, const float* env_gt,
                                   const float* seg_small, const float* env_ind, float* diffuse, float* spec, float* mask,
                                   float* coef, float* parts, float* workspace, int bn, int K, int R, int C, int eh, int ew, int imH,
                                   int imW, float F0, int premap, void* stream) {
  SGR_REQUIRE(albedo && normal && rough && axis && lamb && weight && dirs && view && env_gt && seg_small && env_ind && diffuse &&
                  spec && mask && coef && parts && workspace,
              "sgr_fused_fwd_recon: NULL tensor");
  SGR_REQUIRE(bn > 0 && K > 0 && R > 0 && C > 0 && eh > 0 && ew > 0, "sgr_fused_fwd_recon: non-positive size");
  SGR_SUPPORTED(fused_recon_ok(K, R, C, eh, ew), "sgr_fused_fwd_recon: needs envWidth 16, SGNum <= 12 (use the unfused calls)");
  if (int rc = check_pool(R, C, imH, imW, "sgr_fused_fwd_recon: BRDF-map / env-grid ratio must be 1 or 2 (pool first)")) return rc;
  Args a{};
  a.albedo = albedo; a.normal = normal; a.rough = rough; a.axis = axis; a.lamb = lamb; a.weight = weight;
  a.dirs = reinterpret_cast<const float4*>(dirs); a.view = view; a.diffuse = diffuse; a.spec = spec;
  a.env_gt = env_gt; a.seg_small = seg_small; a.env_ind = env_ind; a.mask = mask;
  set_dims(a, bn, K, R, C, eh, ew, imH, imW);
  a.F0 = F0; a.premap = premap;
  const int tiles = recon_tiles(R * C);
  float* ws0 = workspace;
  float* den_img = workspace + (size_t)bn * tiles * 4;
  a.ws = ws0;
  const hipStream_t st = (hipStream_t)stream;
  const dim3 grid = wave_grid(bn, R, C), block(kWave);
  const bool p1 = (imH == R && imW == C);
  if (K <= 6) {
    if (p1) hipLaunchKernelGGL((fwd_fast_kernel<6, 1, 16, 16, false, true, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((fwd_fast_kernel<6, 2, 16, 16, false, true, true>), grid, block, 0, st, a);
  } else {
    if (p1) hipLaunchKernelGGL((fwd_fast_kernel<12, 1, 16, 16, false, true, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((fwd_fast_kernel<12, 2, 16, 16, false, true, true>), grid, block, 0, st, a);
  }
  hipLaunchKernelGGL(recon_fold0, dim3(bn), dim3(kRThreads), 0, st, ws0, coef, den_img, tiles);
  hipLaunchKernelGGL(recon_fold1, dim3(1), dim3(kRThreads), 0, st, ws0, den_img, parts, bn, 0);   // parts = (0, local sum of the env mask)
  return sgr_check((int)hipGetLastError(), "sgr_fused_fwd_recon");
}

extern "C" int sgr_fused_bwd_recon(const float* albedo, const float* normal, const float* rough, const float* axis, const float* lamb,
                                   const float* weight, const float* dirs, const float* view, const float* env_gt, const float* mask,
                                   const float* coef, const float* den_global, const float* g_diffuse, const float* g_spec,
                                   float* g_axis, float* g_lamb, float* g_weight, float* parts, float* workspace, int bn, int K, int R,
                                   int C, int eh, int ew, int imH, int imW, float F0, int premap, float offset, float rec_weight,
                                   void* stream) {
  SGR_REQUIRE(albedo && normal && rough && axis && lamb && weight && dirs && view && env_gt && mask && coef && g_diffuse && g_spec &&
                  g_axis && g_lamb && g_weight && parts && workspace,
              "sgr_fused_bwd_recon: NULL tensor");
  SGR_REQUIRE(bn > 0 && K > 0 && R > 0 && C > 0 && eh > 0 && ew > 0, "sgr_fused_bwd_recon: non-positive size");
  SGR_SUPPORTED(fused_recon_ok(K, R, C, eh, ew), "sgr_fused_bwd_recon: needs envWidth 16, SGNum <= 12 (use the unfused calls)");
  if (int rc = check_pool(R, C, imH, imW, "sgr_fused_bwd_recon: BRDF-map / env-grid ratio must be 1 or 2 (pool first)")) return rc;
  Args a{};
  a.albedo = albedo; a.normal = normal; a.rough = rough; a.axis = axis; a.lamb = lamb; a.weight = weight;
  a.dirs = reinterpret_cast<const float4*>(dirs); a.view = view; a.g_diffuse = g_diffuse; a.g_spec = g_spec;
  a.g_axis = g_axis; a.g_lamb = g_lamb; a.g_weight = g_weight;
  a.env_gt = env_gt; a.mask_in = mask; a.coef = coef; a.offset = offset;
  set_dims(a, bn, K, R, C, eh, ew, imH, imW);
  a.F0 = F0; a.premap = premap;
  const int tiles = recon_tiles(R * C);
  float* ws1 = workspace + (size_t)bn * tiles * 3;
  float* den_img = workspace + (size_t)bn * tiles * 4;
  float* scale = den_img + bn;
  a.ws = ws1; a.rec_scale = scale;
  const hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(recon_scale_kernel, dim3(1), dim3(64), 0, st, den_img, den_global, scale, bn, rec_weight / (3.0f * (float)(eh * ew)));
  const dim3 grid = wave_grid(bn, R, C), block(2 * kWave);
  if (imH == R && imW == C) hipLaunchKernelGGL((sg_bwd_recon_kernel<1>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((sg_bwd_recon_kernel<2>), grid, block, 0, st, a);
  hipLaunchKernelGGL(recon_fold1, dim3(1), dim3(kRThreads), 0, st, ws1, den_img, parts, bn, tiles);     // parts = (loss numerator, local sum of the env mask)
  return sgr_check((int)hipGetLastError(), "sgr_fused_bwd_recon");
}
