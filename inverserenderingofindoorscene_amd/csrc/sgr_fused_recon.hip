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
// Backward work decomposition: one wave = 32 pixels x 2 lobe groups.  Lanes l and l+32 own the same pixel;
// lanes 0..31 hold lobes 0..5, lanes 32..63 lobes 6..11 (the register budget of 12 lobes + 12 gradient sets
// per lane would not fit).  The radiance of a direction needs all 12 lobes and its cotangent is needed by
// both halves, so per chunk of 4 directions (two azimuths x both half rows) the halves trade values with
// v_permlane32_swap_b32 -- gfx950's swap of the upper 32 lanes of one VGPR with the lower 32 of another --
// arranged so that no select is needed:
//   swap(D = partial of half row 1, S = partial of half row 0); D + S = total of half row 1 in lanes 0..31,
//                                                                       total of half row 0 in lanes 32..63
//   each half evaluates loss, reconstruction and render cotangent for the half row it now holds (6 values)
//   swap(D = g, S = g):  D = cotangent of half row 1 in all lanes, S = cotangent of half row 0 in all lanes
// 12 VALU swaps + 6 adds per chunk; no LDS traffic, no barrier, no duplicated transcendental.
#include <string.h>
#include "sgr_forward.inl"
#include "sgr_recon_fold.h"

namespace sgr {

// ---- the pass in azimuth pairs (sgr_pk.inl): v_pk_fma_f32 over the directions (e, a), (e, a+1) ----
// Per azimuth pair and lobe: 10 packed instructions + 4 v_exp for the exponentials and the partial radiance, 18 packed for the
// gradient accumulation (the exponentials are kept; round 6: no sum T t, see sharpness_grad); per pair 6 swaps + 3 packed adds for the
// radiance, 6 v_rcp + 6 v_log for loss and cotangent, 6 swaps to hand the cotangents round.
//
// Round 3: EW = 32 walks a table row as two virtual rows of 8 + 8 directions (tile32_dma_issue_vrow, as sg_bwd_pk_kernel), and
// NG = 4 lane groups per pixel (one wave = 16 pixels x 4 groups of 6 lobes: lanes pl, pl + 16, pl + 32, pl + 48) carry up to 24
// lobes -- twelve lobes plus their gradient sets do not fit one lane, and unlike the layer's backward this pass cannot split the
// lobes over workgroups: the loss cotangent of a direction needs the radiance of ALL lobes.  The groups trade values in two
// stages, v_permlane32_swap between the halves (as with NG = 2) and gfx950's v_permlane16_swap between the 16-lane rows of a
// half (swap(D, S): D's odd rows <-> S's even rows), so that group (half, sub) ends up with the total radiance of ONE direction
// of the azimuth pair -- sign 1 - half, azimuth component 1 - sub -- evaluates loss, reconstruction and render cotangent for
// it (scalar: the transcendental count per pixel is what it was), and the four cotangents travel back the same way:
// 6 + 3 swaps for the reduce-scatter, 3 + 6 for the all-gather.
// Round 4, measured and NOT adopted (profiles/r04a_objective_bwd_variants.txt, one box): keeping the 24 exponentials of an azimuth
// pair in LDS between the radiance half and the gradient half of the loop body (+6 KB per wave) -- 333 vs 318 us; trading the halves'
// partial radiance / cotangents with ds_bpermute_b32 (the idle LDS crossbar, select-free by giving each half-wave the sign of the half
// row it owns) instead of v_permlane32_swap -- 293 instead of 317 VALU instructions per azimuth pair, but 334 vs 318 us: two LDS round
// trips per pair in the dependency chain cost more latency than the 12 swaps cost issue slots at two waves per SIMD.  The hot
// (orthonormal-frame) loop has no scratch access: the spilled registers (private segment 140-170 B) belong to the degenerate-frame
// loop and the code around the row loop, so register relief has nothing to buy (tools/loop_mix.py).
#ifndef SGR_RECON_FENCES
#define SGR_RECON_FENCES 1      // loop-body register fences of sg_bwd_recon_pk_kernel: 1 = lobes + row constants (default), 2 = also the BRDF / cotangent constants (round 2: 36 more bytes of scratch, +1 %), 0 = none (op_sel broadcasts get hoisted into register pairs: 543 vs 330 us)
#endif
__device__ __forceinline__ void swap16(float& d, float& s) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(d), __float_as_uint(s), false, false);
  d = __uint_as_float(r[0]);
  s = __uint_as_float(r[1]);
}
// virtual row `vr` of a 16-pixel tile: [3][16 px][16 floats] (3 KB), 16-byte slots XOR-swizzled by (row >> 2) & 3; three DMA instructions
template <int AUX, int EW>
__device__ __forceinline__ void tile16_dma_issue_vrow(float* tile, __amdgpu_buffer_rsrc_t rsrc, int p0, int RC, int J, int vr, int lane) {
  const int e = EW == 16 ? vr : (vr >> 1), q = EW == 16 ? 0 : (vr & 1);
  const int row = lane >> 2, slot = lane & 3;
  const int ls = slot ^ ((row >> 2) & 3);                                            // logical 16-byte slot this lane fills
  const int col = EW == 16 ? 4 * ls : 4 * ls + 8 * q + (ls >= 2 ? 8 : 0);            // slots 0,1: half row 0; slots 2,3: half row 1
  const int voff = (row * J + col) * 4;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int soff = (int)((((size_t)c * RC + p0) * J + e * EW) * 4);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LdsPtr)(tile + c * 16 * 16), 16, voff, soff, 0, AUX);
  }
}
template <> __device__ __forceinline__ void wait_vmcnt<3>() { asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }

// GRADS = false (round 4): the same pass without its gradient half -- the reconstruction-loss VALUE alone, for forward-only callers
// of the objective (torch.no_grad(): testLight.py-style evaluation): lobes -> exponentials -> radiance of the azimuth pair -> the
// log-L2 term.  No shading frame, no cotangents, no accumulators, no all-gather; the first 6 (+3) swaps stay.
// Six lobes per lane group, two resident waves per SIMD.  (Round 5 carried the lobes per group and the occupancy as template parameters for
// the NG = 4 x 3-lobe experiment -- measured slower, see fused_bwd_recon_impl -- and left them and an odd-lobe-count path without an
// instantiation; ADVICE round 5: removed, the record of the experiment is profiles/r05a_* and DESIGN_HISTORY.md.)
template <int POOL, int EW = 16, int NG = 2, bool HEADS = false, bool GRADS = true>
__global__ __launch_bounds__(kWave, 2) void sg_bwd_recon_pk_kernel(const Args a) {
  constexpr int KPW = 6, HALF = 8, NP = 4, KH = KPW / 2, Q = EW / 16, PXW = kWave / NG;       // PXW pixels per wave
  constexpr int kTile = 3 * PXW * 16;                                           // floats per ground-truth virtual-row tile
  // ground-truth rows: double-buffered one-row tiles, row vr+1 requested while row vr is consumed.  PMC (round 4, profiles/r04b_pmc_traffic_
  // config2_batch16_objective.txt): 826 MB fetched where ~610 MB are read -- a row is one 64-byte half of each 128-byte line, and the other
  // half is requested a row of arithmetic (~5 us) later, after the XCD's L2 has turned over.  A ring of three tiles requested two rows at a
  // time (sg_bwd_pk_kernel's scheme: both halves of a line back to back) brought the fetch down to 671 MB but cost 18 KB of LDS and ran
  // SLOWER, 342 vs 321 us (r04c_kbench.txt): this kernel moves 3 TB/s and is bound by VALU issue, the re-fetches come out of the Infinity
  // Cache, and the 12-instruction DMA bursts sit in front of the loads the next row's arithmetic is waiting for.  Not adopted.
  // Round 5 (profiles/r05c_objective_bwd_gt_stream.txt, one box, three alternations; kbench = the bench loop for this kernel): what the stream
  // costs is not the WAIT -- with the requests issued and never waited for (tools/ablate.sh 32) 316 us against 321-325 -- but the requests
  // themselves: without them (ablation 1) 268-271 us.  So nothing that hides latency better can pay: a ring of three one-row tiles with row
  // vr+2 requested (twice the latency tolerance) 333-336 us, the non-temporal policy on the rows 335-337, both 347-352.
  __shared__ __attribute__((aligned(16))) float tile[2 * kTile];
  static_assert(NG == 2 || NG == 4, "two or four lane groups per pixel");

  const int lane = threadIdx.x, half = lane >> 5, sub = (lane >> 4) & 1, pl = lane & (PXW - 1);
  const int grp = NG == 2 ? half : (lane >> 4);       // this lane's group of six lobes
  const int own = 1 - half;                            // the half row (sign) this lane evaluates the cotangent of
  const int oaz = 1 - sub;                             // NG = 4: and the azimuth component (0: x, 1: y) of the pair
  const int RC = a.R * a.C, K = a.K;
  Pix x;
  x.lane = lane;
  {
    const int tiles = (RC + PXW - 1) / PXW;
    x.b = (int)blockIdx.x / tiles;
    x.p0 = ((int)blockIdx.x - x.b * tiles) * PXW;
    x.active = (x.p0 + pl) < RC;
    x.p = x.active ? (x.p0 + pl) : (RC - 1);
  }
  const int b = x.b, p = x.p;
  const int nvr = a.eh * Q;

  __amdgpu_buffer_rsrc_t gimg = env_rsrc(a.env_gt + (size_t)b * 3 * RC * a.J, RC, a.J);
  // Round 6 (the review's ISA finding, tools/loop_scratch.py): at 256 VGPRs the requests' lane offsets were spilled and RELOADED FROM SCRATCH in
  // front of every row's requests -- scratch loads count in vmcnt like the LDS-DMA requests, so the reload's s_waitcnt vmcnt(0) between the
  // two halves of the burst waited for the three requests just issued: a full memory latency per row and wave (base vs the loop without those
  // reloads, kbench: 314-324 vs 304-310 us).  The lane id now comes from v_mbcnt each time (a one-wave workgroup: lane == threadIdx.x; asm
  // volatile, so that it is not hoisted and spilled again) and the offset is re-formed from it: 7 VALU instructions per row, no live register.
  auto issue = [&](float* dst, int vr) {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    if constexpr (NG == 2) tile32_dma_issue_vrow<SGR_DMA_AUX, EW>(dst, gimg, x.p0, RC, a.J, vr, l);
    else tile16_dma_issue_vrow<SGR_DMA_AUX, EW>(dst, gimg, x.p0, RC, a.J, vr, l);
  };
  if (!(SGR_ABLATE & 1)) issue(tile, 0);

  PixLocal q{};
  OrthoPix oq{};
  bool ortho = true;
  f32x2 gds[3] = {splat2(0.f), splat2(0.f), splat2(0.f)};      // (gD_c A_c / pi, gS_c)
  if constexpr (GRADS) {
    float alb[3];
    const Frame f = load_frame<POOL>(a, x, alb);
    q = make_local(f, a.F0);
    oq = make_ortho_pix(q);
    ortho = __all(frame_is_orthonormal(q));
    const size_t o = (size_t)b * 3 * RC;
    const unsigned up = (unsigned)p;
#pragma unroll
    for (int c = 0; c < 3; ++c)
      gds[c] = f32x2{(a.g_diffuse + o + (size_t)c * RC)[up] * (alb[c] * kInvPi), (a.g_spec + o + (size_t)c * RC)[up]};
  }
  // reconstruction side: x = cf p + off;  dnum/dp = 2 m (ln x - ln(gt + off)) cf / x        (coef is a constant)
  const float cf = a.coef[b], off = a.offset;
  const float m = x.active ? (a.mask_in + (size_t)b * RC)[(unsigned)p] : 0.0f;
  float den;
  if (a.den_global) {
    den = a.den_global[0];
  } else {
    double sden = 0.0;
    for (int i = 0; i < a.bn; ++i) sden += (double)a.den_img[i];
    den = (float)sden;
  }
  const float rec_scale = a.rec_w3j / fmaxf(den, 1e-5f);
  // times dl (in log2 units) / x.  Negated: the loop evaluates log2((gt + off) / x) = -dl as it comes out of v_log_f32 -- the loss term
  // is its square, and the sign rides in this factor instead of six v_xor per azimuth pair
  f32x2 grec = splat2(-2.0f * m * rec_scale * cf * kLn2);
  f32x2 lossp = splat2(0.0f);

  LobesPk<KPW> P;      // axes pre-multiplied by lp = lam * log2e (floored), as in sg_bwd_pk_kernel
  load_lobes_pk<KPW, true, HEADS>(a, b, (unsigned)p, x.active, grp * KPW, P, false);
  // (az lp) of the lobe pairs is read once per ROW (czr below): parked in lane-private LDS slots (1.5 KB of the 8 KB a wave has left at two
  // waves per SIMD) instead of six registers -- the compiler's own choice was to spill one pair to scratch and reload it behind the row's
  // DMA requests (vmcnt(0) again: the decoder-heads instantiation), ds_read counts in lgkmcnt and waits for nothing but itself
  // -- and lp itself likewise (once per row in czr, once in the epilogue; with the heads as prologue the compiler spilled THAT pair next).
  // Only where the gradient accumulators make registers scarce: the loss-value-only instantiation (GRADS = false, 97 VGPRs) keeps them.
  constexpr bool PARK = GRADS;
  __shared__ __attribute__((aligned(8))) f32x2 az_slots[PARK ? KH : 1][PARK ? kWave : 1];
  __shared__ __attribute__((aligned(8))) f32x2 lp_slots[PARK ? KH : 1][PARK ? kWave : 1];
  if constexpr (PARK) {
#pragma unroll
    for (int mm = 0; mm < KH; ++mm) { az_slots[mm][lane] = P.azp[mm]; lp_slots[mm][lane] = P.lpp[mm]; }
  }
  auto az_of = [&](int mm) -> f32x2 { if constexpr (PARK) return az_slots[mm][lane]; else return P.azp[mm]; };
  auto lp_of = [&](int mm) -> f32x2 { if constexpr (PARK) return lp_slots[mm][lane]; else return P.lpp[mm]; };
  f32x2 gw0[KPW], gw1[KPW], gw2[KPW], gz[KPW], gx[KPW], gy[KPW];
#pragma unroll
  for (int k = 0; k < KPW; ++k) gw0[k] = gw1[k] = gw2[k] = gz[k] = gx[k] = gy[k] = splat2(0.f);

  const SepTable rows = as_sep_table(a.rows);
  const PairTable cpt = as_pair_table(a.cols, EW);
  const XTable xt = (XTable)(a.cols + EW);

  auto row_loop = [&](auto ortho_c, PixLocal& q) {      // q: the caller's frame (the degenerate-frame path brings its own, see below)
    constexpr bool ORTHO = decltype(ortho_c)::value;
    for (int vr = 0; vr < nvr; ++vr) {
      const int e = Q == 1 ? vr : (vr >> 1), aoff = Q == 1 ? 0 : (vr & 1) * NP;      // table row; first azimuth pair of this virtual row
      constexpr int kRow = NG == 2 ? 6 : 3;      // LDS-DMA instructions per virtual-row tile
      const float* cur = tile + (vr & 1) * kTile;
      if (SGR_ABLATE & 1) {
        // ablation: no ground-truth rows requested or waited for
      } else if (vr + 1 < nvr) {
        issue(tile + ((vr + 1) & 1) * kTile, vr + 1);
        if (!(SGR_ABLATE & 32)) wait_vmcnt<kRow>();      // this virtual row has landed; the next stays in flight
      } else {
        if (!(SGR_ABLATE & 32)) wait_vmcnt<0>();
      }
      if (GRADS && !ORTHO) fence_row_invariants(q);
      const f32x8 row = rows[e];
      const float sr = row[0], cr = row[1];
      f32x2 czr[KH];
#pragma unroll
      for (int mm = 0; mm < KH; ++mm) czr[mm] = pfma(az_of(mm), splat2(cr), -lp_of(mm));      // lp (az c_e - 1)
      const RowCtx rc = make_row_ctx(q, row, GRADS);
      OrthoRow orow = make_ortho_row(rc.ro, q.vv);

#pragma unroll 1
      for (int ap = 0; ap < NP; ++ap) {
#if SGR_RECON_FENCES >= 1
        fence_lobes<KPW>(P);
#pragma unroll
        for (int mm = 0; mm < KH; ++mm) { SGR_FENCE2(czr[mm]); }
#endif
#if SGR_RECON_FENCES >= 2
#pragma unroll
        for (int mm = 0; mm < KH; ++mm) { SGR_FENCE2(P.lpp[mm]); }
#pragma unroll
        for (int c = 0; c < 3; ++c) SGR_FENCE2(gds[c]);
        SGR_FENCE2(grec);
        if (ORTHO) { SGR_FENCE2(oq.vB); SGR_FENCE2(oq.ff); SGR_FENCE2(orow.n2c); SGR_FENCE2(orow.cvc); }
#endif
        const f32x4 cs = cpt[aoff + ap];
        const f32x2 ca = {cs[0], cs[1]}, sa = {cs[2], cs[3]};
        const f32x2 srv = splat2(sr);
        // ---- 1. this group's lobes: exponentials and partial radiance of the 4 directions -----------------
        f32x2 ep[KPW], em[KPW];
        f32x2 v[2][3];        // [half row][colour], the azimuth pair
#pragma unroll
        for (int sg = 0; sg < 2; ++sg)
#pragma unroll
          for (int c = 0; c < 3; ++c) v[sg][c] = splat2(0.f);
#pragma unroll
        for (int k = 0; k < KPW; ++k) {
          const f32x2 cz = half_of(czr[k / 2], k & 1), w2 = half_of(P.w2p[k / 2], k & 1);
          const f32x2 u = pfma(SGR_HI(P.axy[k]), sa, SGR_LO(P.axy[k]) * ca);
          const f32x2 xp = pfma(srv, u, cz), xm = pfma(-srv, u, cz);      // lp t
          ep[k] = f32x2{fexp2(xp.x), fexp2(xp.y)};
          em[k] = f32x2{fexp2(xm.x), fexp2(xm.y)};
          v[0][0] = pfma(SGR_LO(P.w01[k]), ep[k], v[0][0]); v[0][1] = pfma(SGR_HI(P.w01[k]), ep[k], v[0][1]); v[0][2] = pfma(w2, ep[k], v[0][2]);
          v[1][0] = pfma(SGR_LO(P.w01[k]), em[k], v[1][0]); v[1][1] = pfma(SGR_HI(P.w01[k]), em[k], v[1][1]); v[1][2] = pfma(w2, em[k], v[1][2]);
        }
        // ---- 2. full radiance of the half row this half-wave owns (lanes 0..31: half row 1, 32..63: half row 0)
        f32x2 tot[3];
        {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            float dx = v[1][c].x, sx = v[0][c].x, dy = v[1][c].y, sy = v[0][c].y;
            swap32(dx, sx);
            swap32(dy, sy);
            tot[c] = f32x2{dx, dy} + f32x2{sx, sy};      // one v_pk_add_f32
          }
        }
        f32x2 g[2][3];
        if constexpr (NG == 2) {
          // ---- 3. its cotangent: reconstruction term (and loss) + render term --------------------------------
          float gt[3][2];
#if SGR_ABLATE & 2
          for (int c = 0; c < 3; ++c) { gt[c][0] = 0.5f + 1e-3f * (float)(ap + c + lane); gt[c][1] = 0.7f; }
          (void)cur;
#else
          tile32_read_pair(cur, pl, own * HALF + ap * 2, gt);
#endif
          f32x2 wt = splat2(0.f), sp = splat2(0.f);
          if constexpr (GRADS) {
            const f32x2 Pv = pfma(SGR_HI(oq.vB), sa, SGR_LO(oq.vB) * ca);
            shade_pair<ORTHO>(q, oq, rc, orow, own, ca, sa, Pv, xt, (aoff + ap) * 2, wt, sp);
          }
          f32x2 go[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const f32x2 xx = pfma(splat2(cf), tot[c], splat2(off));
            const f32x2 r = {__builtin_amdgcn_rcpf(xx.x), __builtin_amdgcn_rcpf(xx.y)};
            const f32x2 ar = (f32x2{gt[c][0], gt[c][1]} + splat2(off)) * r;
            const f32x2 dl = {__builtin_amdgcn_logf(ar.x), __builtin_amdgcn_logf(ar.y)};   // -log2(x / (gt + off)); see grec
            lossp = pfma(dl, dl, lossp);
            if constexpr (GRADS) {
              const f32x2 gr = pfma(SGR_HI(gds[c]), sp, SGR_LO(gds[c]));
              go[c] = pfma(grec * dl, r, wt * gr);
            }
          }
          // ---- 4. both half rows' cotangents to all lanes --------------------------------------------------------
          if constexpr (GRADS) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              float dx = go[c].x, sx = go[c].x, dy = go[c].y, sy = go[c].y;
#if !(SGR_ABLATE & 4)
              swap32(dx, sx);
              swap32(dy, sy);
#endif
              g[1][c] = f32x2{dx, dy};     // from lanes 0..31
              g[0][c] = f32x2{sx, sy};     // from lanes 32..63
            }
          }
        } else {
          // ---- 2b. second stage of the reduce-scatter: rows of 16 lanes.  swap16(D = y totals, S = x totals); D + S leaves
          // even rows (sub 0) with the full radiance of azimuth y and odd rows with that of azimuth x
          float t1[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            float d_ = tot[c].y, s_ = tot[c].x;
            swap16(d_, s_);
            t1[c] = d_ + s_;
          }
          // ---- 3. cotangent of this group's ONE direction (sign own, azimuth 2 (aoff + ap) + oaz): loss, reconstruction, render
          const int jj = own * HALF + ap * 2 + oaz;
          const unsigned addr = lds_addr(cur) + (unsigned)(pl * 64) + (unsigned)((((jj >> 2) ^ ((pl >> 2) & 3)) * 4 + (jj & 3)) * 4);
          float gt[3];
          asm volatile("ds_read_b32 %0, %1" : "=v"(gt[0]) : "v"(addr) : "memory");
          asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(gt[1]) : "v"(addr), "n"(1 * 16 * 64) : "memory");
          asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(gt[2]) : "v"(addr), "n"(2 * 16 * 64) : "memory");
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          const float ca1 = oaz ? ca.y : ca.x, sa1 = oaz ? sa.y : sa.x;
          float wt1 = 0.f, sp1 = 0.f;
          if constexpr (GRADS) shade_dir<ORTHO>(q, rc, own, ca1, sa1, xt, (aoff + ap) * 2 + oaz, wt1, sp1);
          float go1[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float xx = fmaf(cf, t1[c], off);
            const float r = __builtin_amdgcn_rcpf(xx);
            const float dl = __builtin_amdgcn_logf((gt[c] + off) * r);      // -log2(x / (gt + off)); see grec
            lossp.x = fmaf(dl, dl, lossp.x);
            if constexpr (GRADS) go1[c] = fmaf(grec.x * dl, r, wt1 * fmaf(gds[c].y, sp1, gds[c].x));
          }
          // ---- 4. all-gather: rows first (swap16(D = go, S = go): D = the even row's value = azimuth y, S = the odd row's =
          // azimuth x, in both rows of the half), then the halves
          if constexpr (GRADS) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              float d_ = go1[c], s_ = go1[c];
              swap16(d_, s_);
              float dx = s_, sx = s_, dy = d_, sy = d_;
              swap32(dx, sx);
              swap32(dy, sy);
              g[1][c] = f32x2{dx, dy};     // from lanes 0..31
              g[0][c] = f32x2{sx, sy};     // from lanes 32..63
            }
          }
        }
        // ---- 5. this group's lobes: gradient accumulation (sg_bwd_pk_kernel's inner loop with the kept exponentials)
        const f32x2 sca = srv * ca, ssa = srv * sa;
        if constexpr (GRADS) {
#pragma unroll
        for (int k = 0; k < KPW; ++k) {
          // (round 6: no sum T t accumulator -- dL/dlam = a . S - w . q in the epilogue, sgr_pk.inl: sharpness_grad -- so u and lp t are
          // not needed here and T enters through its sum and difference alone: 18 packed instructions per lobe)
          const f32x2 w2 = half_of(P.w2p[k / 2], k & 1);
          gw0[k] = pfma(g[0][0], ep[k], gw0[k]); gw1[k] = pfma(g[0][1], ep[k], gw1[k]); gw2[k] = pfma(g[0][2], ep[k], gw2[k]);
          gw0[k] = pfma(g[1][0], em[k], gw0[k]); gw1[k] = pfma(g[1][1], em[k], gw1[k]); gw2[k] = pfma(g[1][2], em[k], gw2[k]);
          const f32x2 Sp = pfma(g[0][2], w2, pfma(g[0][1], SGR_HI(P.w01[k]), g[0][0] * SGR_LO(P.w01[k])));
          const f32x2 Sm = pfma(g[1][2], w2, pfma(g[1][1], SGR_HI(P.w01[k]), g[1][0] * SGR_LO(P.w01[k])));
          const f32x2 Tm = Sm * em[k];
          const f32x2 Ts = pfma(Sp, ep[k], Tm), Td = pfma(Sp, ep[k], -Tm);
          gz[k] = pfma(splat2(cr), Ts, gz[k]);
          gx[k] = pfma(sca, Td, gx[k]);
          gy[k] = pfma(ssa, Td, gy[k]);
        }
        }
      }
    }
  };
  // The degenerate-frame loop (N parallel to up, |N|^2 off 1 by more than 2e-6: a handful of pixels in real data, none in most waves) reads
  // nine more frame terms than the orthonormal one.  Kept live from the prologue they were what pushed the kernel over 256 VGPRs -- spill
  // stores in the prologue and reloads in the epilogue of EVERY wave (tools/ablate.sh 64: that loop compiled out = 12 B of scratch and
  // 304-310 us against 314-324).  Round 6: that path re-derives its frame from memory (the pixel index passes through an opaque asm, so the
  // loads are not merged with the prologue's), and nothing but the orthonormal path's own terms lives across the branch.
  if ((SGR_ABLATE & 64) || ortho) {
    row_loop(std::true_type{}, q);
  } else {
    PixLocal q2 = q;
    if constexpr (GRADS) {
      Pix x2 = x;
      asm volatile("" : "+v"(x2.p));
      float alb2[3];
      const Frame f2 = load_frame<POOL>(a, x2, alb2);
      q2 = make_local(f2, a.F0);
    }
    row_loop(std::false_type{}, q2);
  }

  // loss partial of the tile: sum_p m_p sum_{c,j} (ln x - ln(gt+off))^2   (each group holds its directions' share)
  {
    float r0 = m * (lossp.x + lossp.y) * (kLn2 * kLn2);
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) r0 += __shfl_xor(r0, s, 64);
    if (lane == 0) a.ws[blockIdx.x] = r0;
  }

  if (GRADS && x.active) {
    // wave-uniform plane base of lobe slot k + the lane's 32-bit byte offset (see load_lobes_pk)
    const unsigned o3_own = ((unsigned)(grp * KPW * 3 * RC) + (unsigned)p) * 4u, o1_own = ((unsigned)(grp * KPW * RC) + (unsigned)p) * 4u;
    char* g_axis_b = reinterpret_cast<char*>(a.g_axis + (size_t)b * K * 3 * RC);
    char* g_lamb_b = reinterpret_cast<char*>(a.g_lamb + (size_t)b * K * RC);
    char* g_weight_b = reinterpret_cast<char*>(a.g_weight + (size_t)b * K * 3 * RC);
#pragma unroll
    for (int k = 0; k < KPW; ++k) {
      const int kk = grp * KPW + k;
      if (kk < K) {
        const f32x2 lp2 = lp_of(k / 2);
        const float lpk = (k & 1) ? lp2.y : lp2.x;
        const float w0 = P.w01[k].x, w1 = P.w01[k].y, w2 = (k & 1) ? P.w2p[k / 2].y : P.w2p[k / 2].x;
        const float lam = fabsf(lpk) <= kLpFloor ? 0.0f : lpk * kLn2;      // the floor stands for lam == 0
        float q0 = gw0[k].x + gw0[k].y, q1 = gw1[k].x + gw1[k].y, q2 = gw2[k].x + gw2[k].y;
        const float sx = gx[k].x + gx[k].y, sy = gy[k].x + gy[k].y, sz = gz[k].x + gz[k].y;
        const f32x2 az2 = az_of(k / 2);
        float glk = sharpness_grad(P.axy[k].x, P.axy[k].y, (k & 1) ? az2.y : az2.x, lpk, sx, sy, sz, w0, w1, w2, q0, q1, q2);
        if (HEADS || a.premap) {
          glk *= premap_grad(lam);
          q0 *= premap_grad(w0); q1 *= premap_grad(w1); q2 *= premap_grad(w2);
        }
        float gax = lam * sx, gay = lam * sy, gaz = lam * sz;
        if (HEADS)      // decoder heads as a prologue: their chain rule as the epilogue
          heads_bwd_lobe(reinterpret_cast<const char*>(a.axis + (size_t)b * K * 3 * RC) + (size_t)k * 3 * RC * 4,
                         reinterpret_cast<const char*>(a.lamb + (size_t)b * K * RC) + (size_t)k * RC * 4,
                         reinterpret_cast<const char*>(a.weight + (size_t)b * K * 3 * RC) + (size_t)k * 3 * RC * 4, o3_own, o1_own,
                         (size_t)RC * 4, gax, gay, gaz, glk, q0, q1, q2);
        char* pa = g_axis_b + (size_t)k * 3 * RC * 4;
        char* pw = g_weight_b + (size_t)k * 3 * RC * 4;
        *reinterpret_cast<float*>(pa + o3_own) = gax;
        *reinterpret_cast<float*>(pa + (size_t)RC * 4 + o3_own) = gay;
        *reinterpret_cast<float*>(pa + (size_t)RC * 8 + o3_own) = gaz;
        *reinterpret_cast<float*>(g_lamb_b + (size_t)k * RC * 4 + o1_own) = glk;
        *reinterpret_cast<float*>(pw + o3_own) = q0;
        *reinterpret_cast<float*>(pw + (size_t)RC * 4 + o3_own) = q1;
        *reinterpret_cast<float*>(pw + (size_t)RC * 8 + o3_own) = q2;
      }
    }
  }
}

// x *= s / applied, skipped entirely when the two are equal (the usual cotangent of a scalar objective is 1)
struct RescaleArgs { float* x[4]; long long n[4]; };
__global__ __launch_bounds__(256) void rescale_kernel(RescaleArgs r, const float* __restrict__ s, const float* __restrict__ applied) {
  const float f = s[0] / applied[0];
  if (f == 1.0f) return;                 // the usual case: a handful of workgroups look and leave (grid-stride: the grid is small)
  float* __restrict__ x = r.x[blockIdx.y];
  const long long n = r.n[blockIdx.y];
  const long long stride = (long long)gridDim.x * 256 * 4;
  for (long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i0 < n; i0 += stride) {
    if (i0 + 3 < n) {
      f32x4 v = *reinterpret_cast<f32x4*>(x + i0);
      v *= f;
      *reinterpret_cast<f32x4*>(x + i0) = v;
    } else {
      for (long long i = i0; i < n; ++i) x[i] *= f;
    }
  }
}
__global__ void set_scalar_kernel(float* dst, const float* src) { dst[0] = src[0]; }
// the same in one launch: `applied` is a pair of slots, read at [parity] by every workgroup and written at [1 - parity] by the first
// one (no workgroup reads the slot that is written); the caller flips the parity after each call
__global__ __launch_bounds__(256) void rescale_flip_kernel(RescaleArgs r, const float* __restrict__ s, float* __restrict__ applied2, int parity) {
  const float sv = s[0], f = sv / applied2[parity];
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) applied2[1 - parity] = sv;
  if (f == 1.0f) return;
  float* __restrict__ x = r.x[blockIdx.y];
  const long long n = r.n[blockIdx.y];
  const long long stride = (long long)gridDim.x * 256 * 4;
  for (long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i0 < n; i0 += stride) {
    if (i0 + 3 < n) {
      f32x4 v = *reinterpret_cast<f32x4*>(x + i0);
      v *= f;
      *reinterpret_cast<f32x4*>(x + i0) = v;
    } else {
      for (long long i = i0; i < n; ++i) x[i] *= f;
    }
  }
}

}  // namespace sgr

using namespace sgr;

static int recon_tiles(int RC) { return (RC + kWave - 1) / kWave; }
// fused objective kernels: 8x16-style (envWidth 16) and 16x32-style (envWidth 32) grids, up to 24 lobes
static bool fused_recon_ok(int K, int R, int C, int eh, int ew) {
  const long long env_bytes = 3LL * R * C * eh * ew * 4;
  return (ew == 16 || ew == 32) && K <= 24 && K >= 1 && env_bytes < (1LL << 31);
}

extern "C" int sgr_fused_recon_supported(int K, int R, int C, int eh, int ew) { return fused_recon_ok(K, R, C, eh, ew) ? 1 : 0; }

static int recon_tiles32(int RC) { return (RC + kPx - 1) / kPx; }
static int recon_tiles16(int RC) { return (RC + 15) / 16; }
// workspace: [bn] per-image mask sums | [4] scale | [bn,tiles32,3] forward partials | [bn,tiles16] loss partials
extern "C" int sgr_fused_recon_workspace_floats(int bn, int R, int C) {
  return bn + 4 + bn * recon_tiles32(R * C) * 3 + bn * recon_tiles16(R * C);
}

static int fused_fwd_recon_impl(const float* albedo, const float* normal, const float* rough, const float* axis, const float* lamb,
                               const float* weight, const float* dirs, const float* view, const float* env_gt,
                               const float* seg_small, int seg_pool2, const float* env_ind, float* lamb_tan, float* weight_tan,
                               float* diffuse, float* spec, float* mask, float* coef, float* parts, float* workspace, int bn, int K, int R,
                               int C, int eh, int ew, int imH, int imW, float F0, int premap, void* stream,
                               FoldJob* deferred = nullptr /* given: the per-image fold is not launched but described here */) {
  SGR_REQUIRE(albedo && normal && rough && axis && lamb && weight && dirs && view && env_gt && seg_small && env_ind && diffuse &&
                  spec && mask && coef && workspace,
              "sgr_fused_fwd_recon: NULL tensor");
  SGR_REQUIRE(bn > 0 && K > 0 && R > 0 && C > 0 && eh > 0 && ew > 0, "sgr_fused_fwd_recon: non-positive size");
  SGR_SUPPORTED(fused_recon_ok(K, R, C, eh, ew), "sgr_fused_fwd_recon: needs envWidth 16 or 32 and SGNum <= 24 (use the unfused calls)");
  if (int rc = check_pool(R, C, imH, imW, "sgr_fused_fwd_recon: BRDF-map / env-grid ratio must be 1 or 2 (pool first)")) return rc;
  Args a{};
  a.albedo = albedo; a.normal = normal; a.rough = rough; a.axis = axis; a.lamb = lamb; a.weight = weight;
  a.dirs = reinterpret_cast<const float4*>(dirs); a.view = view; a.diffuse = diffuse; a.spec = spec;
  a.env_gt = env_gt; a.seg_small = seg_small; a.seg_pool2 = seg_pool2; a.env_ind = env_ind; a.mask = mask;
  a.lamb_tan = lamb_tan; a.weight_tan = weight_tan;
  set_dims(a, bn, K, R, C, eh, ew, imH, imW);
  a.F0 = F0; a.premap = premap == 1 ? 1 : (premap == 3 ? 3 : 0);
  SGR_REQUIRE(premap >= 0 && premap <= 3, "sgr_fused_fwd_recon: premap must be 0..3");
  SGR_SUPPORTED(premap != 3 || K > 6, "sgr_fused_fwd_recon: premap 3 (decoder heads as a prologue) needs 6 < SGNum <= 24");
  // More than six lobes, or the 16x32 grid: the packed half-wave statistics kernel, whatever the pre-map -- at three waves per SIMD the 42
  // tanh per lane of the decoder heads disappear behind the other waves' row loops (165 us with or without them at config 2).  Rounds 2-4 ran
  // 7..12 lobes with premap <= 2 through the one-pixel-per-lane form (fwd_pk_kernel<12, .., HAS_GT>: single-buffered ground-truth tile, 893 MB
  // fetched per launch for 617 algorithmic, 88 B of scratch -- the round-4 review's finding); same-box A/B in the bench loop, round 5
  // (profiles/r05c_bench_ab.txt): objective step 0.500-0.513 vs 0.503-0.504 ms, with standalone heads 0.776 vs 0.781-0.782 -- a tie on warm
  // data, at 1.04-1.1x the algorithmic traffic instead of 1.46x, so the instantiation is gone.  Up to six lobes: one pixel per lane.
  const bool wide = K > 6 || ew == 32;
  const int tiles = wide ? recon_tiles32(R * C) : recon_tiles(R * C);
  float* den_img = workspace;
  float* ws0 = workspace + bn + 4;
  a.ws = ws0;
  const hipStream_t st = (hipStream_t)stream;
  const bool p1 = (imH == R && imW == C);
  const bool heads = premap == 3;
  if (wide) {
    const dim3 grid((unsigned)(bn * tiles)), block(kWave);
#define SGR_LAUNCH_GT(KPW_, EW_, OCC_)                                                                        \
    do {                                                                                                      \
      if (heads) {                                                                                            \
        if (p1) hipLaunchKernelGGL((fwd_pk_half_gt_kernel<1, KPW_, EW_, OCC_, true>), grid, block, 0, st, a);  \
        else hipLaunchKernelGGL((fwd_pk_half_gt_kernel<2, KPW_, EW_, OCC_, true>), grid, block, 0, st, a);     \
      } else {                                                                                                \
        if (p1) hipLaunchKernelGGL((fwd_pk_half_gt_kernel<1, KPW_, EW_, OCC_>), grid, block, 0, st, a);        \
        else hipLaunchKernelGGL((fwd_pk_half_gt_kernel<2, KPW_, EW_, OCC_>), grid, block, 0, st, a);           \
      }                                                                                                       \
    } while (0)
    if (K <= 12 && ew == 16) SGR_LAUNCH_GT(6, 16, 3);
    else if (K <= 12) SGR_LAUNCH_GT(6, 32, 2);
    else if (ew == 16) SGR_LAUNCH_GT(12, 16, 2);
    else SGR_LAUNCH_GT(12, 32, 2);
#undef SGR_LAUNCH_GT
  } else {
    const dim3 grid = wave_grid(bn, R, C), block(kWave);
    if (p1) hipLaunchKernelGGL((fwd_pk_kernel<6, 1, false, true, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((fwd_pk_kernel<6, 2, false, true, true>), grid, block, 0, st, a);
  }
  if (deferred) {
    *deferred = FoldJob{ws0, coef, den_img, tiles};
    return sgr_check((int)hipGetLastError(), "sgr_fused_fwd_recon");
  }
  hipLaunchKernelGGL(recon_fold0, dim3(bn), dim3(kRThreads), 0, st, ws0, coef, den_img, tiles);
  if (parts)      // (0, local sum of the env mask): what a sharded caller all-reduces before the backward pass; nobody else needs it
    hipLaunchKernelGGL(recon_fold1, dim3(1), dim3(kFold1Threads), 0, st, ws0, den_img, parts, bn, 0, ObjectiveTail{});
  return sgr_check((int)hipGetLastError(), "sgr_fused_fwd_recon");
}

extern "C" int sgr_fused_fwd_recon(const float* albedo, const float* normal, const float* rough, const float* axis, const float* lamb,
                                   const float* weight, const float* dirs, const float* view, const float* env_gt,
                                   const float* seg_small, const float* env_ind, float* diffuse, float* spec, float* mask,
                                   float* coef, float* parts, float* workspace, int bn, int K, int R, int C, int eh, int ew, int imH,
                                   int imW, float F0, int premap, void* stream) {
  return fused_fwd_recon_impl(albedo, normal, rough, axis, lamb, weight, dirs, view, env_gt, seg_small, 0, env_ind, nullptr, nullptr, diffuse,
                              spec, mask, coef, parts, workspace, bn, K, R, C, eh, ew, imH, imW, F0, premap, stream);
}
// also returns the post-tan sharpness / intensity for sgr_fused_bwd_recon(premap = 2)
extern "C" int sgr_fused_fwd_recon_tan(const float* albedo, const float* normal, const float* rough, const float* axis,
                                       const float* lamb, const float* weight, const float* dirs, const float* view,
                                       const float* env_gt, const float* seg_small, const float* env_ind, float* lamb_tan,
                                       float* weight_tan, float* diffuse, float* spec, float* mask, float* coef, float* parts,
                                       float* workspace, int bn, int K, int R, int C, int eh, int ew, int imH, int imW, float F0,
                                       int premap, void* stream) {
  return fused_fwd_recon_impl(albedo, normal, rough, axis, lamb, weight, dirs, view, env_gt, seg_small, 0, env_ind, lamb_tan, weight_tan,
                              diffuse, spec, mask, coef, parts, workspace, bn, K, R, C, eh, ew, imH, imW, F0, premap, stream);
}
// the same with the object mask at its own resolution: [bn,1,segH,segW] with (segH, segW) = (R, C) or (2R, 2C) -- pooled 2x2 by the
// kernel (wrapperBRDFLight.py:171), no separate pooling pass
extern "C" int sgr_fused_fwd_recon_seg(const float* albedo, const float* normal, const float* rough, const float* axis,
                                       const float* lamb, const float* weight, const float* dirs, const float* view,
                                       const float* env_gt, const float* seg, int segH, int segW, const float* env_ind, float* lamb_tan,
                                       float* weight_tan, float* diffuse, float* spec, float* mask, float* coef, float* parts,
                                       float* workspace, int bn, int K, int R, int C, int eh, int ew, int imH, int imW, float F0,
                                       int premap, void* stream) {
  const bool same = segH == R && segW == C, twice = segH == 2 * R && segW == 2 * C;
  SGR_SUPPORTED(same || twice, "sgr_fused_fwd_recon_seg: object mask / env-grid ratio must be 1 or 2 (pool first)");
  return fused_fwd_recon_impl(albedo, normal, rough, axis, lamb, weight, dirs, view, env_gt, seg, twice ? 1 : 0, env_ind, lamb_tan, weight_tan,
                              diffuse, spec, mask, coef, parts, workspace, bn, K, R, C, eh, ew, imH, imW, F0, premap, stream);
}

// The forward half of the light objective on ONE rank in four launches (ABI 5): the statistics kernel, then the render loss's three passes
// with (i) the per-image fold of the env statistics as an extra workgroup per image of the first pass -- nothing between the two reads
// it -- and (ii) ren_weight * d renderErr / d{diffuse, spec} written by the third (g_diffuse / g_spec given) -- what
// sgr_fused_fwd_recon_seg + sgr_render_loss_fwd_total + sgr_render_loss_bwd_scaled did in six.  seg [bn,1,imH,imW] or [bn,1,R,C].
extern "C" int sgr_light_objective_fwd(const float* albedo, const float* normal, const float* rough, const float* axis, const float* lamb,
                                       const float* weight, const float* dirs, const float* view, const float* env_gt, const float* im,
                                       const float* seg, const float* env_ind, float* lamb_tan, float* weight_tan, float* diffuse, float* spec,
                                       float* mask, float* coef_env, float* im_small, float* seg_small, float* rendered, float* coef_ds,
                                       float* parts_r, float* render_err, float* scale_r, float ren_weight, float* g_diffuse, float* g_spec,
                                       float* recon_workspace, float* loss_workspace, int bn, int K, int R, int C, int eh, int ew, int imH,
                                       int imW, int brdfH, int brdfW, float F0, int premap, void* stream) {
  SGR_REQUIRE(im && seg && im_small && seg_small && rendered && coef_ds && parts_r && render_err && scale_r && loss_workspace,
              "sgr_light_objective_fwd: NULL tensor");
  SGR_REQUIRE((g_diffuse == nullptr) == (g_spec == nullptr), "sgr_light_objective_fwd: g_diffuse / g_spec come together");
  const bool same = imH == R && imW == C, twice = imH == 2 * R && imW == 2 * C;
  SGR_SUPPORTED(same || twice, "sgr_light_objective_fwd: image / env-grid ratio must be 1 or 2 (pool first)");
  FoldJob job{};
  if (int rc = fused_fwd_recon_impl(albedo, normal, rough, axis, lamb, weight, dirs, view, env_gt, seg, twice ? 1 : 0, env_ind, lamb_tan, weight_tan,
                                    diffuse, spec, mask, coef_env, nullptr, recon_workspace, bn, K, R, C, eh, ew, brdfH, brdfW, F0, premap, stream, &job))
    return rc;
  return render_loss_fwd_launch(diffuse, spec, im, seg, im_small, seg_small, rendered, coef_ds, parts_r, render_err, scale_r, 3.0f, ren_weight,
                                g_diffuse, g_spec, loss_workspace, bn, R, C, imH, imW, job, stream);
}

static int fused_bwd_recon_impl(const float* albedo, const float* normal, const float* rough, const float* axis, const float* lamb,
                               const float* weight, const float* dirs, const float* view, const float* env_gt, const float* mask,
                               const float* coef, const float* den_global, const float* g_diffuse, const float* g_spec,
                               float* g_axis, float* g_lamb, float* g_weight, float* parts, float* workspace, int bn, int K, int R,
                               int C, int eh, int ew, int imH, int imW, float F0, int premap, float offset, float rec_weight,
                               ObjectiveTail tail, void* stream) {
  const bool grads = g_axis != nullptr;      // all three or none: none = the loss value alone (forward-only callers)
  SGR_REQUIRE(albedo && normal && rough && axis && lamb && weight && dirs && view && env_gt && mask && coef && parts && workspace,
              "sgr_fused_bwd_recon: NULL tensor");
  SGR_REQUIRE((g_axis && g_lamb && g_weight && g_diffuse && g_spec) || (!g_axis && !g_lamb && !g_weight),
              "sgr_fused_bwd_recon: the gradient outputs come all three (with g_diffuse / g_spec) or not at all");
  SGR_REQUIRE(bn > 0 && K > 0 && R > 0 && C > 0 && eh > 0 && ew > 0, "sgr_fused_bwd_recon: non-positive size");
  SGR_REQUIRE(premap >= 0 && premap <= 3, "sgr_fused_bwd_recon: premap must be 0..3");
  SGR_SUPPORTED(premap != 3 || K > 6, "sgr_fused_bwd_recon: premap 3 (decoder heads as a prologue) needs 6 < SGNum <= 24");
  SGR_SUPPORTED(fused_recon_ok(K, R, C, eh, ew), "sgr_fused_bwd_recon: needs envWidth 16 or 32 and SGNum <= 24 (use the unfused calls)");
  if (int rc = check_pool(R, C, imH, imW, "sgr_fused_bwd_recon: BRDF-map / env-grid ratio must be 1 or 2 (pool first)")) return rc;
  Args a{};
  a.albedo = albedo; a.normal = normal; a.rough = rough; a.axis = axis; a.lamb = lamb; a.weight = weight;
  a.dirs = reinterpret_cast<const float4*>(dirs); a.view = view; a.g_diffuse = g_diffuse; a.g_spec = g_spec;
  a.g_axis = g_axis; a.g_lamb = g_lamb; a.g_weight = g_weight;
  a.env_gt = env_gt; a.mask_in = mask; a.coef = coef; a.offset = offset;
  set_dims(a, bn, K, R, C, eh, ew, imH, imW);
  a.F0 = F0; a.premap = premap;
  const int tiles32 = recon_tiles32(R * C);
  // Round 5, measured and NOT adopted (profiles/r05a_objective_bwd_lane_groups_kbench.txt, r05a_sq_config2_batch16_objective_k12ng4.txt; one box,
  // three alternations): 7..12 lobes on the 8x16 grid as FOUR lane groups of THREE lobes (a quarter of the accumulators per lane; the
  // template parameters it needed are gone again).  Asked for three waves per SIMD: 168 VGPRs, 56 B of scratch outside the hot loop, 2.78 resident
  // waves per SIMD, VALU-busy 0.92 at 2.05 GHz (1888 busy cycles per SIMD and microsecond against 1544) -- and 356-364 us against 324-328, because
  // the wave now covers 16 pixels: per azimuth pair 213 VALU instructions per 16 pixels against 311 per 32 (+37 %: the microfacet terms and the
  // loss / cotangent of a direction are scalar, one direction per lane group, and the exchange is 18 swaps per 16 pixels instead of 12 per 32).
  // Asked for four waves (128 VGPRs): 220 B of scratch, 5 scratch loads in the hot loop, 464-467 us.  Both pass the objective's parity tests.
  const bool four = K > 12;                          // four lane groups of six lobes per pixel: 16 pixels per wave
  const int tiles = four ? recon_tiles16(R * C) : tiles32;
  float* den_img = workspace;
  float* ws1 = workspace + bn + 4 + (size_t)bn * tiles32 * 3;
  a.ws = ws1; a.den_img = den_img; a.den_global = den_global; a.rec_w3j = rec_weight / (3.0f * (float)(eh * ew));
  const hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)(bn * tiles)), block(kWave);
  const bool p1 = (imH == R && imW == C);
  {
#define SGR_LAUNCH_BR(EW_, NG_)                                                                              \
    do {                                                                                                     \
      if (!grads) {                                                                                          \
        if (premap == 3) hipLaunchKernelGGL((sg_bwd_recon_pk_kernel<1, EW_, NG_, true, false>), grid, block, 0, st, a);   \
        else hipLaunchKernelGGL((sg_bwd_recon_pk_kernel<1, EW_, NG_, false, false>), grid, block, 0, st, a);              \
      } else if (premap == 3) {                                                                              \
        if (p1) hipLaunchKernelGGL((sg_bwd_recon_pk_kernel<1, EW_, NG_, true>), grid, block, 0, st, a);      \
        else hipLaunchKernelGGL((sg_bwd_recon_pk_kernel<2, EW_, NG_, true>), grid, block, 0, st, a);         \
      } else {                                                                                               \
        if (p1) hipLaunchKernelGGL((sg_bwd_recon_pk_kernel<1, EW_, NG_>), grid, block, 0, st, a);            \
        else hipLaunchKernelGGL((sg_bwd_recon_pk_kernel<2, EW_, NG_>), grid, block, 0, st, a);               \
      }                                                                                                      \
    } while (0)
    if (!four && ew == 16) SGR_LAUNCH_BR(16, 2);
    else if (!four) SGR_LAUNCH_BR(32, 2);
    else if (ew == 16) SGR_LAUNCH_BR(16, 4);
    else SGR_LAUNCH_BR(32, 4);
#undef SGR_LAUNCH_BR
  }
  hipLaunchKernelGGL(recon_fold1, dim3(1), dim3(kFold1Threads), 0, st, ws1, den_img, parts, bn, tiles, tail);     // parts = (loss numerator, local sum of the env mask)
  return sgr_check((int)hipGetLastError(), "sgr_fused_bwd_recon");
}

extern "C" int sgr_fused_bwd_recon(const float* albedo, const float* normal, const float* rough, const float* axis, const float* lamb,
                                   const float* weight, const float* dirs, const float* view, const float* env_gt, const float* mask,
                                   const float* coef, const float* den_global, const float* g_diffuse, const float* g_spec,
                                   float* g_axis, float* g_lamb, float* g_weight, float* parts, float* workspace, int bn, int K, int R,
                                   int C, int eh, int ew, int imH, int imW, float F0, int premap, float offset, float rec_weight,
                                   void* stream) {
  return fused_bwd_recon_impl(albedo, normal, rough, axis, lamb, weight, dirs, view, env_gt, mask, coef, den_global, g_diffuse, g_spec, g_axis,
                              g_lamb, g_weight, parts, workspace, bn, K, R, C, eh, ew, imH, imW, F0, premap, offset, rec_weight, ObjectiveTail{},
                              stream);
}

// one rank: the objective's scalar tail (sgr_objective_finalize) comes out of the same fold
extern "C" int sgr_fused_bwd_recon_total(const float* albedo, const float* normal, const float* rough, const float* axis, const float* lamb,
                                         const float* weight, const float* dirs, const float* view, const float* env_gt, const float* mask,
                                         const float* coef, const float* g_diffuse, const float* g_spec, float* g_axis, float* g_lamb,
                                         float* g_weight, float* parts, float* workspace, int bn, int K, int R, int C, int eh, int ew, int imH,
                                         int imW, float F0, int premap, float offset, float rec_weight, const float* render_err, float ren_weight,
                                         float* objective, float* recon_err, float* applied2, void* stream) {
  SGR_REQUIRE(render_err && objective && recon_err, "sgr_fused_bwd_recon_total: NULL scalar");
  ObjectiveTail tail{render_err, ren_weight, rec_weight, 3.0f * (float)(eh * ew), objective, recon_err, applied2};
  return fused_bwd_recon_impl(albedo, normal, rough, axis, lamb, weight, dirs, view, env_gt, mask, coef, nullptr, g_diffuse, g_spec, g_axis,
                              g_lamb, g_weight, parts, workspace, bn, K, R, C, eh, ew, imH, imW, F0, premap, offset, rec_weight, tail, stream);
}

// Cotangent scaling for gradients that were produced ahead of the backward call (sgr.light_objective):
// every x[i] (i = 0..count-1, n[i] floats, 16-byte aligned) is multiplied in place by scale / *applied, then
// *applied = scale; nothing is touched when the two are equal.  All on the stream, no host sync.
extern "C" int sgr_rescale_inplace(float* const* x, const long long* n, int count, const float* scale, float* applied, void* stream) {
  SGR_REQUIRE(x && n && scale && applied && count >= 0, "sgr_rescale_inplace: NULL argument");
  const hipStream_t st = (hipStream_t)stream;
  SGR_SUPPORTED(count <= 4, "sgr_rescale_inplace: at most 4 tensors per call");
  if (count > 0) {
    RescaleArgs r{};
    long long nmax = 0;
    for (int i = 0; i < count; ++i) {
      SGR_REQUIRE(x[i] && n[i] > 0, "sgr_rescale_inplace: empty tensor");
      r.x[i] = x[i]; r.n[i] = n[i];
      nmax = n[i] > nmax ? n[i] : nmax;
    }
    const long long blocks = (nmax + 1023) / 1024;
    hipLaunchKernelGGL(rescale_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024), (unsigned)count), dim3(256), 0, st, r, scale, applied);
  }
  hipLaunchKernelGGL(set_scalar_kernel, dim3(1), dim3(1), 0, st, applied, scale);
  return sgr_check((int)hipGetLastError(), "sgr_rescale_inplace");
}

// sgr_rescale_inplace in ONE launch: applied2 = two slots, the current factor in applied2[parity]; on return the new one is in
// applied2[1 - parity] (the caller flips its parity).  count in 1..4.
extern "C" int sgr_rescale_inplace_flip(float* const* x, const long long* n, int count, const float* scale, float* applied2, int parity,
                                        void* stream) {
  SGR_REQUIRE(x && n && scale && applied2 && count >= 1 && (parity == 0 || parity == 1), "sgr_rescale_inplace_flip: bad argument");
  SGR_SUPPORTED(count <= 4, "sgr_rescale_inplace_flip: at most 4 tensors per call");
  RescaleArgs r{};
  long long nmax = 0;
  for (int i = 0; i < count; ++i) {
    SGR_REQUIRE(x[i] && n[i] > 0, "sgr_rescale_inplace_flip: empty tensor");
    r.x[i] = x[i]; r.n[i] = n[i];
    nmax = n[i] > nmax ? n[i] : nmax;
  }
  const long long blocks = (nmax + 1023) / 1024;
  hipLaunchKernelGGL(rescale_flip_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024), (unsigned)count), dim3(256), 0, (hipStream_t)stream, r,
                     scale, applied2, parity);
  return sgr_check((int)hipGetLastError(), "sgr_rescale_inplace_flip");
}
