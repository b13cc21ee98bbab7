// Packed-fp32 kernels for the reference's 8x16 direction grid (envWidth 16), round 2.
//
// Why: on gfx950 a VALU instruction issued next to transcendentals costs ~4 cycles of its SIMD whatever it is
// and however many waves are resident (tools/ubench5: `1 v_exp + 6 v_fmac` = 31.6 cycles at 1..4 waves per SIMD,
// `1 v_exp + 3 v_pk_fma_f32` = 20.4; the two-wave dual issue that gives pure FMA streams 2.1 cycles per
// instruction is lost as soon as v_exp/v_rcp/v_rsq/v_permlane are in the stream, however they are grouped).
// So the lever is the instruction COUNT: every FMA of the inner loops is issued as one half of a v_pk_fma_f32.
//
// Packing axis: two neighbouring azimuths (a, a+1) of one table row and one sign.  With the separable table
//     lam (a_k . l_j - 1) = +-s_e U_ka + C_ke,   U_ka = lam (ax ca_a + ay sa_a),
// (U_k,a , U_k,a+1) is one v_pk_mul + one v_pk_fma from the SGPR pairs (ca_a, ca_a+1), (sa_a, sa_a+1), the two
// exponents of a sign are one v_pk_fma, the three colour accumulations one v_pk_fma each, and the accumulators
// come out as (env[.., a], env[.., a+1]) pairs -- the order the env image, its LDS tiles (ds_write_b128 /
// ds_read_b64) and the cotangent rows already use.  The microfacet terms are evaluated for the same two
// directions at a time (brdf_ortho_pair).  Degenerate shading frames (wave-uniform test) take the scalar
// Gram-matrix path of sgr_fast.inl.
#pragma once
#include "sgr_fast.inl"

namespace sgr {

// Wave priority of the prologue (loads, pre-map, shading frame) over the row loop: a wave that has just started takes the issue slots
// first and reaches its own row loop sooner -- the prologue is 25-40 % of a wave's life (tools/wavetrace) and runs next to one or two
// waves that are in their loops.  Measured A/B on one box (round 3): +0.9 % on the fwd+bwd step at priority 1 or 3, forward -2 us;
// raising the priority of the data-movement phases as well (env tile flush, cotangent row requests) changed nothing; issuing every
// prologue load before the first use (lobes, BRDF maps, cotangents in one burst) changed nothing either -- the latencies are hidden by
// the other waves already; and in the objective's backward kernel the same priority made the step 1-2 % SLOWER, so that one stays at
// the default.  Round 4: nor do the objective's forward (statistics) kernels -- without it the objective step is 3-6 us faster in the loop
// (profiles/r04q_prio_bench.txt) -- so the forward bodies raise the priority only when they are not the HAS_GT variant.  0 = hardware default.
// -DSGR_ABLATE=<bits> (development builds only, tools/ablate.sh): time a kernel with one of its data-movement components REMOVED -- results are
// wrong, the timing difference is that component's cost (an upper bound on what hiding it better could buy).  1: no LDS-DMA row requests / waits
// (cotangent rows, ground-truth rows); 2: no LDS tile reads; 4: no cotangent all-gather (objective backward); 8: lobe parameters synthesised in
// registers instead of loaded (the prologue's 42 / 84 loads per lane); 16: no env-image tile writes / stores (forward).
#ifndef SGR_ABLATE
#define SGR_ABLATE 0
#endif
#ifndef SGR_PROLOGUE_PRIO
#define SGR_PROLOGUE_PRIO 3
#endif
#if SGR_PROLOGUE_PRIO > 0
#define SGR_PRIO_PROLOGUE __builtin_amdgcn_s_setprio(SGR_PROLOGUE_PRIO);
#define SGR_PRIO_LOOP __builtin_amdgcn_s_setprio(0);
#else
#define SGR_PRIO_PROLOGUE
#define SGR_PRIO_LOOP
#endif
// -DSGR_TRACE (development builds only, tools/wavetrace): every wave of the packed kernels records when and where it ran
#ifdef SGR_TRACE
// t*: s_memrealtime (100 MHz, wall); c*: s_memtime (shader-clock cycles) -- round 5: (c1 - c0) / (t1 - t0) is the clock the wave actually ran at
struct TraceRec { unsigned long long t0, t1, tp, c0, c1; unsigned hw, xcc; };
static __device__ TraceRec* g_trace = nullptr;
#define SGR_TRACE_BEGIN const unsigned long long trace_t0_ = __builtin_amdgcn_s_memrealtime(), trace_c0_ = __builtin_amdgcn_s_memtime(); unsigned long long trace_tp_ = 0;
#define SGR_TRACE_MARK trace_tp_ = __builtin_amdgcn_s_memrealtime();
#define SGR_TRACE_END                                                                                  \
  if (threadIdx.x == 0 && g_trace) {                                                                   \
    unsigned hw_, xcc_;                                                                                \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                                  \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                                \
    TraceRec r_; r_.t0 = trace_t0_; r_.t1 = __builtin_amdgcn_s_memrealtime(); r_.tp = trace_tp_; r_.c0 = trace_c0_; r_.c1 = __builtin_amdgcn_s_memtime(); r_.hw = hw_; r_.xcc = xcc_; \
    g_trace[blockIdx.x] = r_;                                                                          \
  }
#else
#define SGR_TRACE_BEGIN SGR_PRIO_PROLOGUE
#define SGR_TRACE_MARK SGR_PRIO_LOOP
#define SGR_TRACE_END
#endif
// the forward bodies: priority (or trace point) unless `quiet` (a compile-time constant)
#ifdef SGR_TRACE
#define SGR_TRACE_BEGIN_UNLESS(quiet) SGR_TRACE_BEGIN
#define SGR_TRACE_MARK_UNLESS(quiet) SGR_TRACE_MARK
#else
#define SGR_TRACE_BEGIN_UNLESS(quiet) if (!(quiet)) { SGR_PRIO_PROLOGUE }
#define SGR_TRACE_MARK_UNLESS(quiet) if (!(quiet)) { SGR_PRIO_LOOP }
#endif

__device__ __forceinline__ f32x2 splat2(float x) { return f32x2{x, x}; }
__device__ __forceinline__ f32x2 pfma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
// One half of a register pair as both operands of a packed instruction: folds into the op_sel / op_sel_hi bits of the
// consuming v_pk_* instruction -- PROVIDED the shuffle is selected in the same basic block as its use.  Hoisted out of
// a loop (LICM) it becomes a materialised (x, x) pair, i.e. a second register per scalar.  Hence FENCE2 below: an
// empty asm that redefines the pair inside the loop body, at no cost in instructions.
#define SGR_LO(P) __builtin_shufflevector(P, P, 0, 0)
#define SGR_HI(P) __builtin_shufflevector(P, P, 1, 1)
#define SGR_FENCE2(P) asm volatile("" : "+v"(P))
__device__ __forceinline__ f32x2 half_of(f32x2 p, int i) { return i ? SGR_HI(p) : SGR_LO(p); }
typedef const f32x4 __attribute__((address_space(4))) * PairTable;   // per azimuth pair: (ca_a, ca_a+1, sa_a, sa_a+1)
__device__ __forceinline__ PairTable as_pair_table(const float* cols, int ew) { return (PairTable)(cols + 4 * ew); }

// the 32-pixel group `g` of a half-wave kernel's grid (lanes l and l + 32 own the same pixel)
__device__ __forceinline__ Pix locate_group32(const Args& a, int g) {
  Pix x;
  x.lane = threadIdx.x;
  const int RC = a.R * a.C, tiles = (RC + kPx - 1) / kPx, pl = x.lane & 31;
  x.b = g / tiles;
  x.p0 = (g - x.b * tiles) * kPx;
  x.active = (x.p0 + pl) < RC;
  x.p = x.active ? (x.p0 + pl) : (RC - 1);
  return x;
}

// the 64-pixel unit `u` of a one-pixel-per-lane launch (locate() with an explicit unit index)
__device__ __forceinline__ Pix locate_unit(const Args& a, int u) {
  Pix x;
  x.lane = threadIdx.x;
  const int RC = a.R * a.C, tiles = (RC + kWave - 1) / kWave;
  x.b = u / tiles;
  x.p0 = (u - x.b * tiles) * kWave;
  x.active = (x.p0 + x.lane) < RC;
  x.p = x.active ? (x.p0 + x.lane) : (RC - 1);
  return x;
}
// SG parameters of the lane's pixel in register pairs (KP even; lobes past K carry zero weights).
// FOLD: axis pre-multiplied by lp = lam * log2e (forward and sg_bwd_pk_kernel); unit axes otherwise (objective backward).
// The backward divides its sharpness gradient by lp again (see there), so a |lp| below kLpFloor = 2^-40 -- lam == 0 is
// what the decoders' clamp produces -- is replaced by the floor: exp2(2^-40 t) is exactly 1, like exp2(0 t).
template <int KP>
struct LobesPk {
  static constexpr int KH = KP / 2;      // lobe pairs (KP is even: load_lobes_pk asserts it)
  f32x2 axy[KP];        // (ax, ay)
  f32x2 w01[KP];        // (w0, w1)
  f32x2 w2p[KH];        // (w2 of lobe 2m, w2 of lobe 2m+1)
  f32x2 azp[KH];        // (az, az)  likewise
  f32x2 lpp[KH];        // (lp, lp)  likewise
};
// Two pre-maps at a time (round 3): tan(fl(fl(0.999 x) pi/2)) with the argument products, the Cody-Waite reduction and the
// polynomial of sgr_math.h's tan_f32 issued as v_pk_mul / v_pk_fma_f32 over a register pair; only the parity select and
// v_rcp_f32 stay per element.  n = rint(y 2/pi) comes out of the 1.5 * 2^23 magic-number add (one packed FMA + one packed
// subtract instead of a multiply and two v_rndne), which also leaves the parity of n in the low mantissa bit of q -- no
// v_cvt.  Valid for |y| < 2^22 pi/2 (the decoders produce y in [0, 1.5692]).  Against tan_f32 the quotient is rounded once
// instead of twice, so n can differ at exact ties of y 2/pi (z = +-pi/4: both branches are equally accurate there).
// 23 instructions per pair against 2 x 19: 48 pre-maps per pixel in the forward.
// Written for two pairs in lockstep: a packed fp32 instruction needs one wait state before a dependent one (the compiler
// pads a lone chain with an s_nop after every instruction), and the two independent chains fill each other's gaps.
__device__ __forceinline__ void premap2x2(f32x2& u, f32x2& v) {
  constexpr float kMagic = 12582912.0f;
  const f32x2 yu = (u * splat2(kPremapScale)) * splat2(kHalfPiHi);      // two rounded products, like torch (no add to contract with)
  const f32x2 yv = (v * splat2(kPremapScale)) * splat2(kHalfPiHi);
  f32x2 qu = pfma(yu, splat2(kTwoOverPi), splat2(kMagic));
  f32x2 qv = pfma(yv, splat2(kTwoOverPi), splat2(kMagic));
  asm("" : "+v"(qu));                                                    // (y c + M) - M must not be simplified
  asm("" : "+v"(qv));
  const f32x2 nu = qu - splat2(kMagic), nv = qv - splat2(kMagic);
  f32x2 zu = pfma(-nu, splat2(kPio2_1), yu), zv = pfma(-nv, splat2(kPio2_1), yv);
  zu = pfma(-nu, splat2(kPio2_2), zu); zv = pfma(-nv, splat2(kPio2_2), zv);
  zu = pfma(-nu, splat2(kPio2_3), zu); zv = pfma(-nv, splat2(kPio2_3), zv);
  const f32x2 su = zu * zu, sv = zv * zv;
  f32x2 pu = pfma(splat2(9.38540185543e-3f), su, splat2(3.11992232697e-3f)), pv = pfma(splat2(9.38540185543e-3f), sv, splat2(3.11992232697e-3f));
  pu = pfma(pu, su, splat2(2.44301354525e-2f)); pv = pfma(pv, sv, splat2(2.44301354525e-2f));
  pu = pfma(pu, su, splat2(5.34112807005e-2f)); pv = pfma(pv, sv, splat2(5.34112807005e-2f));
  pu = pfma(pu, su, splat2(1.33387994085e-1f)); pv = pfma(pv, sv, splat2(1.33387994085e-1f));
  pu = pfma(pu, su, splat2(3.33331568548e-1f)); pv = pfma(pv, sv, splat2(3.33331568548e-1f));
  const f32x2 tu = pfma(pu * su, zu, zu), tv = pfma(pv * sv, zv, zv);
  const bool u0 = (__float_as_uint(qu.x) & 1u) != 0, u1 = (__float_as_uint(qu.y) & 1u) != 0;
  const bool v0 = (__float_as_uint(qv.x) & 1u) != 0, v1 = (__float_as_uint(qv.y) & 1u) != 0;
  u = f32x2{u0 ? -frcp(tu.x) : tu.x, u1 ? -frcp(tu.y) : tu.y};
  v = f32x2{v0 ? -frcp(tv.x) : tv.x, v1 ? -frcp(tv.y) : tv.y};
}

// ---- the light decoders' output activations as a prologue (premap == 3; models.py:336-346) ----
//   axis = normalize3(1.01 tanh x),   lamb, weight = clamp(0.5 (1.01 tanh x + 1), 0, 1)   -> then the tan pre-map as for premap == 1
// tanh two at a time as 1 - 2 / (e^{2x} + 1): accurate to ~1e-7 ABSOLUTE for every x (e^{2x} = inf and 0 give +-1), which is all
// the normalisation and the affine map in front of a [0, 1] clamp can see; the standalone pass (sgr_heads.hip, bit-pinned
// against the reference's clamp decisions) adds a polynomial below |x| = 0.625 for relative accuracy.  7 instructions per pair.
__device__ __forceinline__ f32x2 tanh2_abs(f32x2 x) {
  const f32x2 y = x * splat2(2.8853900817779268f);
  const f32x2 d = f32x2{fexp2(y.x), fexp2(y.y)} + splat2(1.0f);
  return pfma(splat2(-2.0f), f32x2{frcp(d.x), frcp(d.y)}, splat2(1.0f));
}
// 0.5 (1.01 t + 1) with torch's op-by-op rounding (the clamp kinks sit on these bits), two at a time
__device__ __forceinline__ f32x2 unit_pre2(f32x2 t) {
#pragma clang fp contract(off)
  f32x2 u = t * splat2(1.01f);
  u = u + splat2(1.0f);
  return u * splat2(0.5f);
}
// two lobes (14 raw decoder outputs) -> unit axes and [0, 1] sharpness / intensity, in place
__device__ __forceinline__ void heads_two_lobes(float (&ax)[2], float (&ay)[2], float (&az)[2], float (&lp)[2], float (&w0)[2], float (&w1)[2],
                                                float (&w2)[2]) {
  const f32x2 a0 = tanh2_abs(f32x2{ax[0], ay[0]}) * splat2(1.01f), a1 = tanh2_abs(f32x2{ax[1], ay[1]}) * splat2(1.01f);
  const f32x2 az2 = tanh2_abs(f32x2{az[0], az[1]}) * splat2(1.01f);
  const f32x2 tl = tanh2_abs(f32x2{lp[0], lp[1]});
  const f32x2 t0 = tanh2_abs(f32x2{w0[0], w1[0]}), t1 = tanh2_abs(f32x2{w0[1], w1[1]}), t2 = tanh2_abs(f32x2{w2[0], w2[1]});
  // a / max(|a|, 1e-6) as a * min(rsq(|a|^2), 1e6)   (|a| = 0: 0 * 1e6 = 0, like 0 / 1e-6)
  const f32x2 n2 = pfma(az2, az2, pfma(f32x2{a0.y, a1.y}, f32x2{a0.y, a1.y}, f32x2{a0.x, a1.x} * f32x2{a0.x, a1.x}));
  const f32x2 inv = {fminf(frsq(n2.x), 1e6f), fminf(frsq(n2.y), 1e6f)};
  const f32x2 y0 = a0 * SGR_LO(inv), y1 = a1 * SGR_HI(inv), yz = az2 * inv;
  ax[0] = y0.x; ay[0] = y0.y; ax[1] = y1.x; ay[1] = y1.y; az[0] = yz.x; az[1] = yz.y;
  const f32x2 ul = unit_pre2(tl), u0 = unit_pre2(t0), u1 = unit_pre2(t1), u2 = unit_pre2(t2);
  lp[0] = clamp01(ul.x); lp[1] = clamp01(ul.y);
  w0[0] = clamp01(u0.x); w1[0] = clamp01(u0.y); w0[1] = clamp01(u1.x); w1[1] = clamp01(u1.y);
  w2[0] = clamp01(u2.x); w2[1] = clamp01(u2.y);
}
// The heads' chain rule for one lobe (the backward kernels' epilogue): cotangents w.r.t. the unit axis and the [0, 1] sharpness /
// intensity in, cotangents w.r.t. the seven raw decoder outputs out.  The raw values are read again here (xa / xl / xw: the lobe's
// planes, wave-uniform; o3 / o1: the lane's byte offsets; rc4 = 4 R C) -- 28 bytes per lobe against seven live registers per lobe
// across the whole kernel.  Gates and formulas as heads_bwd_kernel (sgr_heads.hip): the min-clamp of the norm blocks the radial
// term below 1e-6, the [0, 1] clamp passes the cotangent on 0 <= pre <= 1 inclusive (torch).
__device__ __forceinline__ void heads_bwd_lobe(const char* xa, const char* xl, const char* xw, unsigned o3, unsigned o1, size_t rc4, float& gx,
                                               float& gy, float& gz, float& gl, float& q0, float& q1, float& q2) {
  const float x0 = *reinterpret_cast<const float*>(xa + o3), x1 = *reinterpret_cast<const float*>(xa + rc4 + o3);
  const float x2 = *reinterpret_cast<const float*>(xa + 2 * rc4 + o3), x3 = *reinterpret_cast<const float*>(xl + o1);
  const float x4 = *reinterpret_cast<const float*>(xw + o3), x5 = *reinterpret_cast<const float*>(xw + rc4 + o3);
  const float x6 = *reinterpret_cast<const float*>(xw + 2 * rc4 + o3);
  const f32x2 t01 = tanh2_abs(f32x2{x0, x1}), t23 = tanh2_abs(f32x2{x2, x3}), t45 = tanh2_abs(f32x2{x4, x5}), t66 = tanh2_abs(f32x2{x6, x6});
  const f32x2 one = splat2(1.0f);
  const f32x2 s01 = pfma(-t01, t01, one), s23 = pfma(-t23, t23, one), s45 = pfma(-t45, t45, one), s66 = pfma(-t66, t66, one);   // 1 - tanh^2
  const float a0 = 1.01f * t01.x, a1 = 1.01f * t01.y, a2 = 1.01f * t23.x;
  const float n2 = fmaf(a2, a2, fmaf(a1, a1, a0 * a0));
  const bool live = n2 >= 1e-12f;                                         // |a| >= 1e-6
  const float inv = live ? frsq_nr(n2) : 1e6f;
  const float dot = live ? fmaf(a2, gz, fmaf(a1, gy, a0 * gx)) * (inv * inv) : 0.0f;
  gx = (gx - a0 * dot) * inv * (1.01f * s01.x);
  gy = (gy - a1 * dot) * inv * (1.01f * s01.y);
  gz = (gz - a2 * dot) * inv * (1.01f * s23.x);
  const f32x2 p3 = unit_pre2(f32x2{t23.y, t66.x}), p45 = unit_pre2(t45);
  gl = (p3.x >= 0.0f && p3.x <= 1.0f) ? gl * (0.505f * s23.y) : 0.0f;
  q0 = (p45.x >= 0.0f && p45.x <= 1.0f) ? q0 * (0.505f * s45.x) : 0.0f;
  q1 = (p45.y >= 0.0f && p45.y <= 1.0f) ? q1 * (0.505f * s45.y) : 0.0f;
  q2 = (p3.y >= 0.0f && p3.y <= 1.0f) ? q2 * (0.505f * s66.x) : 0.0f;
}

// |lp| below this floor stands for lam == 0 (see LobesPk): 2^-40 -- exp2(2^-40 t) is exactly 1 for the |t| <= 2 of the
// layer, like exp2(0 t), and sums of T (lp t) stay twenty orders of magnitude above the denormal range even for the
// 1e-9-sized cotangents of a normalised training loss (1e-30, the first choice, did not: ADVICE round 2)
constexpr float kLpFloor = 9.094947017729282e-13f;

// dL/dlam of a lobe WITHOUT a per-direction accumulator (round 6).  With T_j = (g_j . w) E_j and t_j = a . l_j - 1:
//     dL/dlam = sum_j T_j t_j = a . (sum_j T_j l_j) - sum_j T_j = a . S - w . q,
// where S = (sum T s_e ca, sum T s_e sa, sum T c_e) are the axis accumulators the kernels carry anyway (dL/da = lam S) and
// q_c = sum_j g_cj E_j the intensity gradients (sum_j T_j = sum_c w_c q_c).  The backward loops therefore need T only through
// T+ + T- and T+ - T- -- three packed instructions per lobe and azimuth pair (T-, then two FMAs) instead of four plus two for sum T t:
// 231 -> 213 VALU instructions per azimuth pair in the layer's backward (254 -> 228 VGPRs), 308 -> 290 in the objective's, whose
// exponents lp t no longer stay live through its gradient half.  The difference cancels by a factor ~1 / mean|t| (up to ~100 for the sharpest
// lobes on the 8x16 grid), so the four-term sum is formed in double; measured on 20 000 random lobes (lam = tan(pi/2 0.999 U[0,1]),
// signed and unsigned cotangents, weighted by the pre-map's 1 + lam^2): 3e-6 .. 9e-6 rel-L2 against 0.5e-6 .. 1.6e-6 for the
// direct sum -- an order below the kernels' other fp32 error (3e-5 .. 8e-5).  `afx, afy, afz, lp`: the FOLDED axis (lp a) and lp.
__device__ __forceinline__ float sharpness_grad(float afx, float afy, float afz, float lp, float sx, float sy, float sz, float w0, float w1, float w2,
                                                float q0, float q1, float q2) {
  const double aS = (double)afx * (double)sx + (double)afy * (double)sy + (double)afz * (double)sz;      // lp (a . S)
  const double wq = (double)w0 * (double)q0 + (double)w1 * (double)q1 + (double)w2 * (double)q2;
  return (float)(aS - (double)lp * wq) * frcp(lp);
}

// Loads (and pre-maps) the lobes straight into pairs; same two-pass structure as load_lobes (sgr_fast.inl): every load
// of every lobe is in flight before the pre-map consumes any.  `kg` = first lobe (may differ between the two halves of
// a wave, so the lobe planes are addressed by 32-bit per-lane offsets into the image's SG block); lobes past K re-read
// lobe K-1 and get zero weights.
// a.premap: 1 = lamb / weight are the decoders' outputs in [0, 1] (pre-mapped here); 0 and 2 = they are post-tan already (2: the
// backward still applies the pre-map's chain rule, see sg_bwd_pk_kernel).  The post-tan values go to a.lamb_tan /
// a.weight_tan when those are given (an output of output2env.output2env, models.py:396-404, and what the backward
// kernels read instead of re-evaluating 24 tangents per lane) -- in a loop of their own after the last pre-map, so that
// the pre-map chains of all lobes interleave freely.
// HEADS (premap == 3 at the C ABI): axis / lamb / weight are the decoders' last-convolution outputs -- heads_two_lobes, then the
// pre-map.  A template parameter, not a branch on a.premap: as a run-time branch the extra code cost the default kernels 1 % (layer)
// to 5 % (objective backward) through register allocation alone (measured, round 3).
template <int KP, bool FOLD, bool HEADS = false>
__device__ __forceinline__ void load_lobes_pk(const Args& a, int b, unsigned up, bool active, int kg, LobesPk<KP>& P, bool write_tan) {
  static_assert(KP % 2 == 0, "lobes come in packed pairs: every instantiation carries 6 or 12 per lane group (round 5's odd-count pad slot had none and is gone)");
  constexpr int KH = KP / 2, KE = KP;
  const int RC = a.R * a.C, K = a.K;
  const float* axis_b = a.axis + (size_t)b * K * 3 * RC;
  const float* lamb_b = a.lamb + (size_t)b * K * RC;
  const float* weight_b = a.weight + (size_t)b * K * 3 * RC;
  float ax[KE], ay[KE], az[KE], lp[KE], w0[KE], w1[KE], w2[KE];
  // addresses: wave-uniform plane base (SGPR pair, scalar arithmetic) + one 32-bit per-lane BYTE offset -> the
  // `global_load_dword v, v_off, s[base:base+1]` form with no per-load vector arithmetic (indexing a float* with a 32-bit
  // element index costs a 64-bit shift + add per load: the scaled index may not fit 32 bits as far as the compiler knows).
  // Lane offsets: first lobe kg of this lane's half; a lobe slot past K (second half of a wave, K < 2 KP) re-reads the
  // slot's lobe of the first half instead (its weights are zeroed below).
  const unsigned o3_own = ((unsigned)(kg * 3 * RC) + up) * 4u, o1_own = ((unsigned)(kg * RC) + up) * 4u, o_low = up * 4u;
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    const bool live = kg + k < K;
    const int ku = min(k, K - 1);                                         // wave-uniform slot (K < KP: clamp)
    const unsigned v3 = live ? o3_own : o_low, v1 = live ? o1_own : o_low;
    const char* pa = reinterpret_cast<const char*>(axis_b + (size_t)ku * 3 * RC);
    const char* pw = reinterpret_cast<const char*>(weight_b + (size_t)ku * 3 * RC);
    const char* pl = reinterpret_cast<const char*>(lamb_b + (size_t)ku * RC);
#if SGR_ABLATE & 8
    { const float t_ = (float)((v3 >> 2) & 63) * (1.0f / 64.0f) + (float)k * 0.01f;
      ax[k] = 0.6f - 0.3f * t_; ay[k] = 0.5f * t_; az[k] = 0.7f; lp[k] = 0.2f + 0.5f * t_; w0[k] = 0.3f + 0.4f * t_; w1[k] = 0.5f; w2[k] = 0.9f - 0.5f * t_;
      (void)pa; (void)pw; (void)pl; (void)v1; }
#else
    ax[k] = *reinterpret_cast<const float*>(pa + v3);
    ay[k] = *reinterpret_cast<const float*>(pa + (size_t)RC * 4 + v3);
    az[k] = *reinterpret_cast<const float*>(pa + (size_t)RC * 8 + v3);
    lp[k] = *reinterpret_cast<const float*>(pl + v1);
    w0[k] = *reinterpret_cast<const float*>(pw + v3);
    w1[k] = *reinterpret_cast<const float*>(pw + (size_t)RC * 4 + v3);
    w2[k] = *reinterpret_cast<const float*>(pw + (size_t)RC * 8 + v3);
#endif
  }
  if (HEADS) {
#pragma unroll
    for (int m = 0; m < KH; ++m) {
      float hx[2] = {ax[2 * m], ax[2 * m + 1]}, hy[2] = {ay[2 * m], ay[2 * m + 1]}, hz[2] = {az[2 * m], az[2 * m + 1]};
      float hl[2] = {lp[2 * m], lp[2 * m + 1]}, h0[2] = {w0[2 * m], w0[2 * m + 1]}, h1[2] = {w1[2 * m], w1[2 * m + 1]};
      float h2[2] = {w2[2 * m], w2[2 * m + 1]};
      heads_two_lobes(hx, hy, hz, hl, h0, h1, h2);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ax[2 * m + i] = hx[i]; ay[2 * m + i] = hy[i]; az[2 * m + i] = hz[i]; lp[2 * m + i] = hl[i];
        w0[2 * m + i] = h0[i]; w1[2 * m + i] = h1[i]; w2[2 * m + i] = h2[i];
      }
    }
  }
  if (HEADS || a.premap == 1) {
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      f32x2 r0 = {lp[k], w0[k]}, r1 = {w1[k], w2[k]};
      premap2x2(r0, r1);
      lp[k] = r0.x; w0[k] = r0.y; w1[k] = r1.x; w2[k] = r1.y;
    }
    if (write_tan && (a.lamb_tan || a.weight_tan)) {
      char* lt_b = a.lamb_tan ? reinterpret_cast<char*>(a.lamb_tan + (size_t)b * K * RC) : nullptr;
      char* wt_b = a.weight_tan ? reinterpret_cast<char*>(a.weight_tan + (size_t)b * K * 3 * RC) : nullptr;
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        if (kg + k < K && active) {
          if (lt_b) *reinterpret_cast<float*>(lt_b + (size_t)k * RC * 4 + o1_own) = lp[k];
          if (wt_b) {
            char* pw = wt_b + (size_t)k * 3 * RC * 4;
            *reinterpret_cast<float*>(pw + o3_own) = w0[k];
            *reinterpret_cast<float*>(pw + (size_t)RC * 4 + o3_own) = w1[k];
            *reinterpret_cast<float*>(pw + (size_t)RC * 8 + o3_own) = w2[k];
          }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    const bool live = kg + k < K;
    float lpk = lp[k] * kLog2e;
    if (FOLD) lpk = fabsf(lpk) < kLpFloor ? kLpFloor : lpk;
    lp[k] = lpk;
    if (FOLD) { ax[k] *= lpk; ay[k] *= lpk; az[k] *= lpk; }
    if (!live) w0[k] = w1[k] = w2[k] = 0.0f;
  }
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    P.axy[k] = f32x2{ax[k], ay[k]};
    P.w01[k] = f32x2{w0[k], w1[k]};
  }
#pragma unroll
  for (int m = 0; m < KH; ++m) {
    P.w2p[m] = f32x2{w2[2 * m], w2[2 * m + 1]};
    P.azp[m] = f32x2{az[2 * m], az[2 * m + 1]};
    P.lpp[m] = f32x2{lp[2 * m], lp[2 * m + 1]};
  }
}
template <int KP>
__device__ __forceinline__ void fence_lobes(LobesPk<KP>& P) {
#pragma unroll
  for (int k = 0; k < KP; ++k) { SGR_FENCE2(P.axy[k]); SGR_FENCE2(P.w01[k]); }
#pragma unroll
  for (int m = 0; m < KP / 2; ++m) SGR_FENCE2(P.w2p[m]);
}

// Per-pixel and per-(pixel, table row) constants of the orthonormal microfacet path, in pairs (see brdf_ortho_dir)
struct OrthoPix { f32x2 vB, ff; };            // (vBx, vBy), (fb, fa)
struct OrthoRow { f32x2 n2c, cvc; float rowe; };      // (nw^2, rowc), (vv + Cv, c1n2), 4e-6 - nw^2
__device__ __forceinline__ OrthoPix make_ortho_pix(const PixLocal& q) {
  OrthoPix o;
  o.vB = f32x2{q.vBx, q.vBy}; o.ff = f32x2{q.fb, q.fa};
  return o;
}
__device__ __forceinline__ OrthoRow make_ortho_row(const RowOrtho& r, float vv) {
  OrthoRow o;
  const float n2 = r.nw * r.nw;
  o.n2c = f32x2{n2, r.rowc}; o.cvc = f32x2{vv + r.Cv, r.c1n2}; o.rowe = 4e-6f - n2;
  return o;
}
// spec of the directions (ss ca_i, ss sa_i, c_e), i = 0, 1: brdf_ortho_dir (sgr_math.h), two azimuths per instruction.
// Pv = vBx ca + vBy sa.  `ss` = +-s_e (wave-uniform).
// Round 6: the clamp |v + l|^2 >= 4e-6 moved onto the azimuth part alone -- with T2c = max(T2, 4e-6 - nw^2) both  max(T2 + nw^2, 4e-6) =
// T2c + nw^2  and the numerator  T2 + (Hm - hh4) + rowc = T2c + rowc  are one packed add each (the same sums of non-negative terms as
// brdf_ortho_dir's, three packed instructions fewer), and |v|^2 rides in the row constant of v.l: 26 instructions per pair instead of 29.
__device__ __forceinline__ f32x2 brdf_ortho_pair(const OrthoPix& q, const OrthoRow& r, float ss, f32x2 ca, f32x2 sa, f32x2 Pv) {
  const f32x2 sv = splat2(ss);
  const f32x2 tx = pfma(sv, ca, SGR_LO(q.vB)), ty = pfma(sv, sa, SGR_HI(q.vB));
  const f32x2 T2 = pfma(tx, tx, ty * ty);
  const f32x2 T2c = {fmaxf(T2.x, r.rowe), fmaxf(T2.y, r.rowe)};
  const f32x2 Hm = T2c + SGR_LO(r.n2c);
  const f32x2 r4 = {frsq(Hm.x), frsq(Hm.y)};
  const f32x2 vdh = pfma(sv, Pv, SGR_LO(r.cvc)) * r4;
  const f32x2 pa = pfma(splat2(-5.55472f), vdh, splat2(-6.98316f)) * vdh;
  const f32x2 pw = {fexp2(pa.x), fexp2(pa.y)};
  const f32x2 nom0 = (T2c + SGR_HI(r.n2c)) * (r4 * r4);
  const f32x2 nr = (nom0 * nom0) * SGR_HI(r.cvc);
  const f32x2 rn = {frcp(clampf(nr.x, 1e-6f, kFourPi)), frcp(clampf(nr.y, 1e-6f, kFourPi))};
  return pfma(SGR_LO(q.ff), pw, SGR_HI(q.ff)) * rn;
}

// (weight, spec) of two directions: packed orthonormal path, or two scalar evaluations of the Gram-matrix path
template <bool ORTHO>
__device__ __forceinline__ void shade_pair(const PixLocal& q, const OrthoPix& oq, const RowCtx& rc, const OrthoRow& orow, int sg, f32x2 ca,
                                           f32x2 sa, f32x2 Pv, XTable xt, int a0, f32x2& wt, f32x2& sp) {
  if (ORTHO) {
    sp = brdf_ortho_pair(oq, orow, sg ? -rc.sr : rc.sr, ca, sa, Pv);
    wt = splat2(rc.ro.wt);
  } else {
    float w0, w1, s0, s1;
    shade_dir<false>(q, rc, sg, ca.x, sa.x, xt, a0, w0, s0);
    shade_dir<false>(q, rc, sg, ca.y, sa.y, xt, a0 + 1, w1, s1);
    wt = f32x2{w0, w1};
    sp = f32x2{s0, s1};
  }
}

// ============================== forward, one pixel per lane, packed ===============================
// fwd_fast_kernel's work decomposition (64 pixels per wave, all K <= KP lobes in registers, env rows leave through
// the 64 x 16 LDS tile) with the arithmetic in azimuth pairs.  Per lobe and azimuth quad: 8 packed U/exponent
// instructions, 8 v_exp_f32, 12 packed accumulations -- against 40 scalar instructions + 8 v_exp_f32.
#ifndef SGR_PK_BWD_AUX
#define SGR_PK_BWD_AUX 2   // cache policy of the packed backward's cotangent rows: 2 = non-temporal (read once)
#endif
#ifndef SGR_PK_BWD_FENCES
#define SGR_PK_BWD_FENCES 2      // loop-body register fences of sg_bwd_pk_kernel: 2 = lobes, row constants, BRDF / cotangent constants; 1 = the first two
#endif
#ifndef SGR_PK_TJ
#define SGR_PK_TJ 16      // directions per flushed env tile row: 16 = one table row (64-byte segments), 32 = two rows (128-byte)
#endif
// HAS_GT (fused objective, no env image; see fwd_fast_kernel): the ground-truth env rows stream in by LDS-DMA, one whole
// table row per 12 KB tile requested the moment the previous row's last pairs are in registers, and every lane accumulates
// <pred, gt>, <pred, pred> and sum(gt) of its pixel as azimuth pairs -- the statistics behind the env mask and the
// LSregress scale (wrapperBRDFLight.py:172-176, models.py:7-21).  Per-wave partials land in a.ws[blockIdx.x * 3 + {0,1,2}].
// `unit` = index of the 64-pixel group in the launch's (image, tile) order; `tile` / `gtile` = the workgroup's LDS (env tile
// of Tile<TJ>::kFloats floats when WRITE_ENV, ground-truth row tile of DmaTile<16>::kFloats floats when HAS_GT)
template <int KP, int POOL, bool WRITE_ENV, bool DO_RENDER, bool HAS_GT, bool HEADS = false>
__device__ __forceinline__ void fwd_pk_body(const Args& a, int unit, float* tile, float* gtile) {
  static_assert(!(HAS_GT && WRITE_ENV), "the statistics variant does not write the env image");
  constexpr int EW = 16, TJ = SGR_PK_TJ, HALF = 8, NQ = 2, RPT = TJ / EW;
  SGR_TRACE_BEGIN_UNLESS(HAS_GT)

  const Pix x = locate_unit(a, unit);
  const int lane = x.lane, b = x.b, p = x.p;
  const int RC = a.R * a.C;

  LobesPk<KP> P;
  load_lobes_pk<KP, true, HEADS>(a, b, (unsigned)p, x.active, 0, P, true);      // post-tan copies leave when a.lamb_tan / a.weight_tan are given

  PixLocal q;
  OrthoPix oq;
  float alb[3] = {0.f, 0.f, 0.f};
  bool ortho = true;
  if (DO_RENDER) {
    const Frame f = load_frame<POOL>(a, x, alb);
    q = make_local(f, a.F0);
    oq = make_ortho_pix(q);
    ortho = __all(frame_is_orthonormal(q));
  }
  const SepTable rows = as_sep_table(a.rows);
  const PairTable cpt = as_pair_table(a.cols, EW);
  const XTable xt = (XTable)(a.cols + EW);
  const size_t img = (size_t)b * 3 * RC * a.J;
  const int eh = a.eh;
  f32x2 dacc[3] = {splat2(0.f), splat2(0.f), splat2(0.f)}, sacc[3] = {splat2(0.f), splat2(0.f), splat2(0.f)};
  f32x2 s_pg = splat2(0.f), s_pp = splat2(0.f), s_g = splat2(0.f);
  __amdgpu_buffer_rsrc_t gimg = env_rsrc(HAS_GT ? a.env_gt + img : a.view, RC, a.J);
  if (HAS_GT) tile_dma_issue<16>(gtile, gimg, x.p0, RC, a.J, 0, lane);
  SGR_TRACE_MARK_UNLESS(HAS_GT)

  auto row_loop = [&](auto ortho_c) {
    constexpr bool ORTHO = decltype(ortho_c)::value;
    for (int e = 0; e < eh; ++e) {
      if (DO_RENDER && !ORTHO) fence_row_invariants(q);
      const f32x8 row = rows[e];
      const float sr = row[0];
      f32x2 Ck[KP / 2];
#pragma unroll
      for (int m = 0; m < KP / 2; ++m) Ck[m] = pfma(P.azp[m], splat2(row[1]), -P.lpp[m]);
      const RowCtx rc = make_row_ctx(q, row, DO_RENDER);
      OrthoRow orow = make_ortho_row(rc.ro, q.vv);
#pragma unroll 1
      for (int aq = 0; aq < NQ; ++aq) {
        fence_lobes<KP>(P);
#pragma unroll
        for (int m = 0; m < KP / 2; ++m) SGR_FENCE2(Ck[m]);
        if (DO_RENDER && ORTHO) { SGR_FENCE2(oq.vB); SGR_FENCE2(oq.ff); SGR_FENCE2(orow.n2c); SGR_FENCE2(orow.cvc); }
        const f32x4 t0 = cpt[2 * aq], t1 = cpt[2 * aq + 1];
        const f32x2 ca[2] = {f32x2{t0[0], t0[1]}, f32x2{t1[0], t1[1]}}, sa[2] = {f32x2{t0[2], t0[3]}, f32x2{t1[2], t1[3]}};
        f32x2 acc[2][3][2];   // [sign][colour][azimuth pair of the quad]
#pragma unroll
        for (int sg = 0; sg < 2; ++sg)
#pragma unroll
          for (int c = 0; c < 3; ++c) acc[sg][c][0] = acc[sg][c][1] = splat2(0.f);
#pragma unroll
        for (int k = 0; k < KP; ++k) {
          const f32x2 ck = half_of(Ck[k / 2], k & 1), w2 = half_of(P.w2p[k / 2], k & 1);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x2 U = pfma(SGR_HI(P.axy[k]), sa[h], SGR_LO(P.axy[k]) * ca[h]);
            const f32x2 tp = pfma(splat2(sr), U, ck);
            const f32x2 tm = pfma(splat2(-sr), U, ck);
            const f32x2 ep = {fexp2(tp.x), fexp2(tp.y)};
            const f32x2 em = {fexp2(tm.x), fexp2(tm.y)};
            acc[0][0][h] = pfma(SGR_LO(P.w01[k]), ep, acc[0][0][h]);
            acc[0][1][h] = pfma(SGR_HI(P.w01[k]), ep, acc[0][1][h]);
            acc[0][2][h] = pfma(w2, ep, acc[0][2][h]);
            acc[1][0][h] = pfma(SGR_LO(P.w01[k]), em, acc[1][0][h]);
            acc[1][1][h] = pfma(SGR_HI(P.w01[k]), em, acc[1][1][h]);
            acc[1][2][h] = pfma(w2, em, acc[1][2][h]);
          }
        }
        if (DO_RENDER) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x2 Pv = pfma(SGR_HI(oq.vB), sa[h], SGR_LO(oq.vB) * ca[h]);
#pragma unroll
            for (int sg = 0; sg < 2; ++sg) {
              f32x2 wt, sp;
              shade_pair<ORTHO>(q, oq, rc, orow, sg, ca[h], sa[h], Pv, xt, aq * 4 + 2 * h, wt, sp);
              const f32x2 sw = sp * wt;
#pragma unroll
              for (int c = 0; c < 3; ++c) {
                dacc[c] = pfma(wt, acc[sg][c][h], dacc[c]);
                sacc[c] = pfma(sw, acc[sg][c][h], sacc[c]);
              }
            }
          }
        }
        if (HAS_GT) {
          if (aq == 0) wait_vmcnt<0>();      // this row's tile (requested a quad of arithmetic ago) has landed
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            float g[2][3][2];
            tile_dma_read_pairs<16>(gtile, lane, aq * 4 + 2 * h, HALF + aq * 4 + 2 * h, g);
            if (h == 1 && aq == NQ - 1 && e + 1 < eh) tile_dma_issue<16>(gtile, gimg, x.p0, RC, a.J, (e + 1) * EW, lane);
#pragma unroll
            for (int sg = 0; sg < 2; ++sg)
#pragma unroll
              for (int c = 0; c < 3; ++c) {
                const f32x2 gv = {g[sg][c][0], g[sg][c][1]};
                s_pg = pfma(acc[sg][c][h], gv, s_pg);
                s_pp = pfma(acc[sg][c][h], acc[sg][c][h], s_pp);
                s_g += gv;
              }
          }
        }
        if (WRITE_ENV) {
#pragma unroll
          for (int sg = 0; sg < 2; ++sg) {
            float e0[4], e1[4], e2[4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              e0[2 * h] = acc[sg][0][h].x; e0[2 * h + 1] = acc[sg][0][h].y;
              e1[2 * h] = acc[sg][1][h].x; e1[2 * h + 1] = acc[sg][1][h].y;
              e2[2 * h] = acc[sg][2][h].x; e2[2 * h + 1] = acc[sg][2][h].y;
            }
            tile_row_write<TJ>(tile, lane, (e % RPT) * EW + sg * HALF + aq * 4, e0, e1, e2);
          }
        }
      }
      if (WRITE_ENV && ((e + 1) % RPT == 0 || e + 1 == eh)) {
        __syncthreads();
        tile_store_global<TJ, true>(tile, a.env_out + img, x.p0, RC, a.J, (e / RPT) * TJ, lane);
        __syncthreads();
      }
    }
  };
  if (ortho) row_loop(std::true_type{}); else row_loop(std::false_type{});

  if (HAS_GT) {
    // env mask of the pixel (wrapperBRDFLight.py:172-174) and the wave's share of the per-image sums
    const float not_dark = ((s_g.x + s_g.y) / (3.0f * (float)a.J)) > 0.001f ? 1.0f : 0.0f;
    const float m = x.active ? seg_small_at(a, b, p) * a.env_ind[b] * not_dark : 0.0f;
    if (x.active) (a.mask + (size_t)b * RC)[(unsigned)p] = m;
    float r0 = m * m * (s_pg.x + s_pg.y), r1 = m * m * (s_pp.x + s_pp.y), r2 = m;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      r0 += __shfl_xor(r0, off, 64); r1 += __shfl_xor(r1, off, 64); r2 += __shfl_xor(r2, off, 64);
    }
    if (lane == 0) {
      float* w = a.ws + (size_t)unit * 3;
      w[0] = r0; w[1] = r1; w[2] = r2;
    }
  }
  if (DO_RENDER && x.active) {
    const size_t o = (size_t)b * 3 * RC;
    const unsigned up = (unsigned)p;
    (a.diffuse + o)[up] = (alb[0] * kInvPi) * (dacc[0].x + dacc[0].y);
    (a.diffuse + o + RC)[up] = (alb[1] * kInvPi) * (dacc[1].x + dacc[1].y);
    (a.diffuse + o + 2 * (size_t)RC)[up] = (alb[2] * kInvPi) * (dacc[2].x + dacc[2].y);
    (a.spec + o)[up] = sacc[0].x + sacc[0].y;
    (a.spec + o + RC)[up] = sacc[1].x + sacc[1].y;
    (a.spec + o + 2 * (size_t)RC)[up] = sacc[2].x + sacc[2].y;
  }
  SGR_TRACE_END
}
template <int KP, int POOL, bool WRITE_ENV, bool DO_RENDER, bool HAS_GT = false, bool HEADS = false>
__global__ __launch_bounds__(kWave, 2) void fwd_pk_kernel(const Args a) {
  __shared__ __attribute__((aligned(16))) float tile[WRITE_ENV ? Tile<SGR_PK_TJ>::kFloats : 4];
  __shared__ __attribute__((aligned(16))) float gtile[HAS_GT ? DmaTile<16>::kFloats : 4];
  fwd_pk_body<KP, POOL, WRITE_ENV, DO_RENDER, HAS_GT, HEADS>(a, (int)blockIdx.x, tile, gtile);
}


// EW = 32 (BASELINE config 5's 16 x 32 grid): a table row is walked as Q = 2 "virtual rows" of 8 + 8 directions -- the
// azimuths [8 q, 8 q + 8) of both half rows -- whose cotangents are gathered into the SAME 16-float-per-pixel LDS tile
// layout by giving the DMA lanes the matching source columns (two 32-byte pieces per pixel and colour), so the loop body is
// the EW = 16 one with the azimuth tables indexed at 4 q + ap.  More than 12 lobes: one workgroup per (32-pixel group,
// group of 12 lobes), the workgroups of a pixel group 8 ids apart (same XCD, shared L2; see sg_bwd_fast_kernel).
template <int AUX, int EW>
__device__ __forceinline__ void tile32_dma_issue_vrow(float* tile, __amdgpu_buffer_rsrc_t rsrc, int p0, int RC, int J, int vr, int lane) {
  if (EW == 16) {
    tile32_dma_issue<AUX>(tile, rsrc, p0, RC, J, vr * 16, lane);
  } else {
    const int e = vr >> 1, q = vr & 1;
    const int lrow = lane >> 2, slot = lane & 3;
    const int ls = slot ^ ((lrow >> 2) & 3);                               // logical 16-byte slot this lane fills (rows lrow and 16 + lrow alike)
    const int col = 4 * ls + 8 * q + (ls >= 2 ? 8 : 0);                    // slots 0,1: half row 0; slots 2,3: half row 1
    const int voff = (lrow * J + col) * 4;                                 // one lane offset; the second request's 16 rows are wave-uniform (tile32_dma_issue)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int soff = (int)((((size_t)c * RC + p0 + it * 16) * J + e * 32) * 4);
        float* dst = tile + (c * kPx + it * 16) * 16;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LdsPtr)dst, 16, voff, soff, 0, AUX);
      }
    }
  }
}


// ============================== forward, half-wave lobe split, packed ==============================================
// fwd_half_kernel's decomposition (sgr_fast.inl: one wave = 32 pixels x 2 groups of 6 lobes; swap(D = share of half row 1,
// S = share of half row 0); D + S leaves lanes 0..31 with the radiance of half row 1 and lanes 32..63 with that of half
// row 0, which each half then shades and stores) with the arithmetic in azimuth pairs.  Against fwd_pk_kernel: half as long
// work units (9600 instead of 4800 at config 2: a shorter last round), half the lobes per lane (fewer registers: OCC = 3
// resident waves per SIMD), no duplicated prologue work (each half pre-maps its own six lobes, the frame is evaluated
// per lane as before) -- for 12 swaps + 6 packed adds per azimuth quad.
// KPW = 12, EW = 32 (OCC 2): SGNum up to 24 on the 16x32 grid of BASELINE config 5 -- the one-pixel-per-lane kernels would
// need 24 lobes per lane there and spill.
// `x` = the 32-pixel group of this wave (locate_group32 / locate_half_of_unit); `tile` = T32Out<EW>::kFloats floats of LDS when WRITE_ENV
// RPF = table rows per flush of the env tile: 1 = one row (EW floats per pixel and colour: 64-byte segments at EW 16), 2 = two
// rows (128-byte segments = whole cache lines: the env stores are what bounds the forward once the working set cycles
// through HBM, and 64-byte segments write at ~3.4 TB/s where 128-byte ones reach ~5, profiles/r02b_storebench*)
// HAS_GT (fused objective without the env image, any SGNum <= 24 on 16- and 32-wide grids; fwd_pk_kernel<.., HAS_GT> covers
// SGNum <= 12 on the 8x16 grid): the ground-truth env streams in by LDS-DMA one VIRTUAL row at a time (8 + 8 directions: the
// azimuths [8 q, 8 q + 8) of both half rows, gathered into the 16-float-per-pixel tile layout of the backward kernels --
// tile32_dma_issue_vrow; on the 8x16 grid a virtual row is a table row), double-buffered, and each half accumulates
// <pred, gt>, <pred, pred>, sum gt over the half rows whose totals it holds; `gtile` = 2 x kT32Floats floats of LDS, `unit` =
// the 32-pixel group's slot in a.ws.
template <int POOL, bool WRITE_ENV, bool DO_RENDER, int KPW, int EW, int RPF = 1, bool HAS_GT = false, bool HEADS = false>
__device__ __forceinline__ void fwd_pk_half_body(const Args& a, const Pix x, float* tile, float* gtile = nullptr, int unit = 0) {
  static_assert(!(HAS_GT && WRITE_ENV), "the statistics variant does not write the env image");
  constexpr int HALF = EW / 2, NQ = HALF / 4, TD = EW * RPF, Q = EW / 16;
  SGR_TRACE_BEGIN_UNLESS(HAS_GT)

  const int lane = threadIdx.x, half = lane >> 5, pl = lane & 31;
  const int own = 1 - half;                         // the half row (sign) whose totals this half-wave ends up holding
  const int RC = a.R * a.C;
  const int b = x.b, p = x.p;

  LobesPk<KPW> P;      // this half's lobes, folded (axis pre-multiplied by lam * log2e)
  load_lobes_pk<KPW, true, HEADS>(a, b, (unsigned)p, x.active, half * KPW, P, true);

  PixLocal q;
  OrthoPix oq;
  float alb[3] = {0.f, 0.f, 0.f};
  bool ortho = true;
  if (DO_RENDER) {
    const Frame f = load_frame<POOL>(a, x, alb);
    q = make_local(f, a.F0);
    oq = make_ortho_pix(q);
    ortho = __all(frame_is_orthonormal(q));
  }
  const SepTable rows = as_sep_table(a.rows);
  const PairTable cpt = as_pair_table(a.cols, EW);
  const XTable xt = (XTable)(a.cols + EW);
  const size_t img = (size_t)b * 3 * RC * a.J;
  const int eh = a.eh;
  f32x2 dacc[3] = {splat2(0.f), splat2(0.f), splat2(0.f)}, sacc[3] = {splat2(0.f), splat2(0.f), splat2(0.f)};
  f32x2 s_pg = splat2(0.f), s_pp = splat2(0.f), s_g = splat2(0.f);
  __amdgpu_buffer_rsrc_t gimg = env_rsrc(HAS_GT ? a.env_gt + img : a.view, RC, a.J);
  const int nvr = eh * Q;
  if (HAS_GT) tile32_dma_issue_vrow<SGR_DMA_AUX, EW>(gtile, gimg, x.p0, RC, a.J, 0, lane);
  SGR_TRACE_MARK_UNLESS(HAS_GT)

  auto row_loop = [&](auto ortho_c) {
    constexpr bool ORTHO = decltype(ortho_c)::value;
    for (int e = 0; e < eh; ++e) {
      if (DO_RENDER && !ORTHO) fence_row_invariants(q);
      const f32x8 row = rows[e];
      const float sr = row[0];
      f32x2 Ck[KPW / 2];
#pragma unroll
      for (int m = 0; m < KPW / 2; ++m) Ck[m] = pfma(P.azp[m], splat2(row[1]), -P.lpp[m]);
      const RowCtx rc = make_row_ctx(q, row, DO_RENDER);
      OrthoRow orow = make_ortho_row(rc.ro, q.vv);
#pragma unroll 1
      for (int aq = 0; aq < NQ; ++aq) {
        const int vr = e * Q + (aq >> 1);               // virtual row of this quad (two quads each)
        const float* gcur = gtile + (HAS_GT ? (vr & 1) * kT32Floats : 0);
        if (HAS_GT && (aq & 1) == 0) {
          // the next virtual row goes into the other buffer (its last reader finished a virtual row ago); this one has landed
          if (vr + 1 < nvr) {
            tile32_dma_issue_vrow<SGR_DMA_AUX, EW>(gtile + ((vr + 1) & 1) * kT32Floats, gimg, x.p0, RC, a.J, vr + 1, lane);
            wait_vmcnt<6>();
          } else {
            wait_vmcnt<0>();
          }
        }
        fence_lobes<KPW>(P);
#pragma unroll
        for (int m = 0; m < KPW / 2; ++m) SGR_FENCE2(Ck[m]);
        if (DO_RENDER && ORTHO) { SGR_FENCE2(oq.vB); SGR_FENCE2(oq.ff); SGR_FENCE2(orow.n2c); SGR_FENCE2(orow.cvc); }
        const f32x4 t0 = cpt[2 * aq], t1 = cpt[2 * aq + 1];
        const f32x2 ca[2] = {f32x2{t0[0], t0[1]}, f32x2{t1[0], t1[1]}}, sa[2] = {f32x2{t0[2], t0[3]}, f32x2{t1[2], t1[3]}};
        f32x2 acc[2][3][2];   // [sign][colour][azimuth pair of the quad]: this half's six lobes' share
#pragma unroll
        for (int sg = 0; sg < 2; ++sg)
#pragma unroll
          for (int c = 0; c < 3; ++c) acc[sg][c][0] = acc[sg][c][1] = splat2(0.f);
#pragma unroll
        for (int k = 0; k < KPW; ++k) {
          const f32x2 ck = half_of(Ck[k / 2], k & 1), w2 = half_of(P.w2p[k / 2], k & 1);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x2 U = pfma(SGR_HI(P.axy[k]), sa[h], SGR_LO(P.axy[k]) * ca[h]);
            const f32x2 tp = pfma(splat2(sr), U, ck);
            const f32x2 tm = pfma(splat2(-sr), U, ck);
            const f32x2 ep = {fexp2(tp.x), fexp2(tp.y)};
            const f32x2 em = {fexp2(tm.x), fexp2(tm.y)};
            acc[0][0][h] = pfma(SGR_LO(P.w01[k]), ep, acc[0][0][h]);
            acc[0][1][h] = pfma(SGR_HI(P.w01[k]), ep, acc[0][1][h]);
            acc[0][2][h] = pfma(w2, ep, acc[0][2][h]);
            acc[1][0][h] = pfma(SGR_LO(P.w01[k]), em, acc[1][0][h]);
            acc[1][1][h] = pfma(SGR_HI(P.w01[k]), em, acc[1][1][h]);
            acc[1][2][h] = pfma(w2, em, acc[1][2][h]);
          }
        }
        // radiance of the half row this half-wave owns
        f32x2 tot[3][2];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            float dx = acc[1][c][h].x, sx = acc[0][c][h].x, dy = acc[1][c][h].y, sy = acc[0][c][h].y;
            swap32(dx, sx);
            swap32(dy, sy);
            tot[c][h] = f32x2{dx, dy} + f32x2{sx, sy};      // one v_pk_add_f32: the swapped halves stay in their register pairs
          }
        if (DO_RENDER) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x2 Pv = pfma(SGR_HI(oq.vB), sa[h], SGR_LO(oq.vB) * ca[h]);
            f32x2 wt, sp;
            shade_pair<ORTHO>(q, oq, rc, orow, own, ca[h], sa[h], Pv, xt, aq * 4 + 2 * h, wt, sp);
            const f32x2 sw = sp * wt;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              dacc[c] = pfma(wt, tot[c][h], dacc[c]);
              sacc[c] = pfma(sw, tot[c][h], sacc[c]);
            }
          }
        }
#if SGR_ABLATE & 16
        if (WRITE_ENV && !DO_RENDER) { dacc[0] += tot[0][0] + tot[0][1]; dacc[1] += tot[1][0] + tot[1][1]; dacc[2] += tot[2][0] + tot[2][1]; }
#endif
        if (HAS_GT) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            float gt[3][2];
            tile32_read_pair(gcur, pl, own * 8 + (aq & 1) * 4 + 2 * h, gt);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const f32x2 gv = {gt[c][0], gt[c][1]};
              s_pg = pfma(tot[c][h], gv, s_pg);
              s_pp = pfma(tot[c][h], tot[c][h], s_pp);
              s_g += gv;
            }
          }
        }
        if (WRITE_ENV && !(SGR_ABLATE & 16)) {
          const float e0[4] = {tot[0][0].x, tot[0][0].y, tot[0][1].x, tot[0][1].y};
          const float e1[4] = {tot[1][0].x, tot[1][0].y, tot[1][1].x, tot[1][1].y};
          const float e2[4] = {tot[2][0].x, tot[2][0].y, tot[2][1].x, tot[2][1].y};
          tile32_write4<TD>(tile, pl, (e % RPF) * EW + own * HALF + aq * 4, e0, e1, e2);
        }
      }
      if (WRITE_ENV && !(SGR_ABLATE & 16) && ((e + 1) % RPF == 0 || e + 1 == eh)) {
        __syncthreads();
        tile32_store_global<TD>(tile, a.env_out + img, x.p0, RC, a.J, (e / RPF) * TD, (e % RPF + 1) * EW, lane);
        __syncthreads();
      }
    }
  };
  if (ortho) row_loop(std::true_type{}); else row_loop(std::false_type{});

  if (HAS_GT) {
    // each half holds the statistics of its half rows: add the two, then env mask of the pixel (wrapperBRDFLight.py:172-174) and
    // the wave's share of the per-image sums (lower half only: both halves now hold the same totals)
    float st[3] = {s_pg.x + s_pg.y, s_pp.x + s_pp.y, s_g.x + s_g.y};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float d_ = st[i], s_ = st[i];
      swap32(d_, s_);
      st[i] = d_ + s_;
    }
    const float not_dark = (st[2] / (3.0f * (float)a.J)) > 0.001f ? 1.0f : 0.0f;
    const float m = x.active ? seg_small_at(a, b, p) * a.env_ind[b] * not_dark : 0.0f;
    if (x.active && half == 0) (a.mask + (size_t)b * RC)[(unsigned)p] = m;
    const float hm = half == 0 ? m : 0.0f;
    float r0 = hm * m * st[0], r1 = hm * m * st[1], r2 = hm;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      r0 += __shfl_xor(r0, off, 64); r1 += __shfl_xor(r1, off, 64); r2 += __shfl_xor(r2, off, 64);
    }
    if (lane == 0) {
      float* w = a.ws + (size_t)unit * 3;
      w[0] = r0; w[1] = r1; w[2] = r2;
    }
  }
  if (DO_RENDER) {
    // each half integrated one half row: add the two
    float v[6] = {dacc[0].x + dacc[0].y, dacc[1].x + dacc[1].y, dacc[2].x + dacc[2].y, sacc[0].x + sacc[0].y, sacc[1].x + sacc[1].y, sacc[2].x + sacc[2].y};
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
  SGR_TRACE_END
}
template <int POOL, bool WRITE_ENV, bool DO_RENDER, int OCC, int KPW = 6, int EW = 16, int RPF = 1, bool HEADS = false>
__global__ __launch_bounds__(kWave, OCC) void fwd_pk_half_kernel(const Args a) {
  __shared__ __attribute__((aligned(16))) float tile[WRITE_ENV ? T32Out<EW * RPF>::kFloats : 4];
  fwd_pk_half_body<POOL, WRITE_ENV, DO_RENDER, KPW, EW, RPF, false, HEADS>(a, locate_group32(a, (int)blockIdx.x), tile);
}
// the statistics variant (fused light objective): render + <pred, gt>, <pred, pred>, sum gt against the streamed ground truth
template <int POOL, int KPW, int EW, int OCC = 2, bool HEADS = false>
__global__ __launch_bounds__(kWave, OCC) void fwd_pk_half_gt_kernel(const Args a) {
  __shared__ __attribute__((aligned(16))) float gtile[2 * kT32Floats];
  fwd_pk_half_body<POOL, false, true, KPW, EW, 1, true, HEADS>(a, locate_group32(a, (int)blockIdx.x), nullptr, gtile, (int)blockIdx.x);
}


// ============================== backward w.r.t. the SG parameters, half-wave, packed ===============
// sg_bwd_half_kernel's decomposition (one wave = 32 pixels x 2 groups of 6 lobes; the env cotangent arrives one table
// row at a time by double-buffered LDS-DMA; each half evaluates the microfacet terms of one half row and the halves
// trade them with v_permlane32_swap) with the arithmetic in azimuth pairs: the cotangent pairs are the ds_read_b64
// results as they come, every gradient accumulator is a pair over the azimuth's parity, folded at the end.
// Per lobe and azimuth pair (4 directions): 27 packed instructions + 4 v_exp_f32, against 58 scalar + 4 v_exp_f32.
//   g[c,j] = gEnv[c,j] + omega_j ndl_j (gD_c A_c/pi + gS_c spec_j);  T = (g . w) E:
//   dL/dw_c = sum g_c E,   dL/dlam = sum T t,   dL/da = lam (sum_a ca_a A_a, sum_a sa_a A_a, sum T c_e),  A_a = sum_e s_e (T+ - T-)
__device__ __forceinline__ void tile32_read_two_pairs2(const float* tile, int pl, int jjA, int jjB, f32x2 (&gA)[3], f32x2 (&gB)[3]) {
  const unsigned base = lds_addr(tile) + (unsigned)(pl * 64);
  const unsigned aA = base + (unsigned)((((jjA >> 2) ^ ((pl >> 2) & 3)) * 4 + (jjA & 3)) * 4);
  const unsigned aB = base + (unsigned)((((jjB >> 2) ^ ((pl >> 2) & 3)) * 4 + (jjB & 3)) * 4);
  asm volatile("ds_read_b64 %0, %1" : "=v"(gA[0]) : "v"(aA) : "memory");
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(gA[1]) : "v"(aA), "n"(1 * kPx * 64) : "memory");
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(gA[2]) : "v"(aA), "n"(2 * kPx * 64) : "memory");
  asm volatile("ds_read_b64 %0, %1" : "=v"(gB[0]) : "v"(aB) : "memory");
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(gB[1]) : "v"(aB), "n"(1 * kPx * 64) : "memory");
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(gB[2]) : "v"(aB), "n"(2 * kPx * 64) : "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// ============================== forwardEnv alone (env image read), packed half-wave (round 4) ======
// The un-fused drop-in call renderingLayer.forwardEnv (models.py:461-522): HBM-bound (1672 B per shaded pixel in).  Round 1's
// kernel (one pixel per lane, scalar arithmetic, 24 KB of LDS per wave: six waves per CU) reached 58 % of the HBM peak, this one 61 %
// (110 -> 105 us warm, 160 -> 158 us with cold buffers, where both sit on the 64-byte-per-128-byte-line access pattern).  This is
// the half-wave form of the other kernels: one wave = 32 pixels, lanes l and l + 32 own the same pixel and integrate one half row
// (sign) each, in azimuth pairs (shade_pair); the env rows arrive one virtual row (8 + 8 directions) at a time by double-buffered
// LDS-DMA into 6 KB tiles -- 12 KB per wave, so twelve waves per CU keep twice the bytes in flight -- and the two halves' six
// sums meet once at the end (6 swaps).
template <int POOL, int EW>
__global__ __launch_bounds__(kWave, 3) void render_pk_half_kernel(const Args a) {
  constexpr int NP = 4, Q = EW / 16;
  __shared__ __attribute__((aligned(16))) float tile[2 * kT32Floats];
  const Pix x = locate_group32(a, (int)blockIdx.x);
  const int lane = threadIdx.x, half = lane >> 5, pl = lane & 31;
  const int own = 1 - half;                         // the half row (sign) this half-wave integrates
  const int RC = a.R * a.C, b = x.b, p = x.p;
  __amdgpu_buffer_rsrc_t eimg = env_rsrc(a.env_in + (size_t)b * 3 * RC * a.J, RC, a.J);
  const int nvr = a.eh * Q;
  tile32_dma_issue_vrow<SGR_DMA_AUX, EW>(tile, eimg, x.p0, RC, a.J, 0, lane);

  float alb[3];
  const Frame f = load_frame<POOL>(a, x, alb);
  PixLocal q = make_local(f, a.F0);
  OrthoPix oq = make_ortho_pix(q);
  const bool ortho = __all(frame_is_orthonormal(q));
  const SepTable rows = as_sep_table(a.rows);
  const PairTable cpt = as_pair_table(a.cols, EW);
  const XTable xt = (XTable)(a.cols + EW);
  f32x2 dacc[3] = {splat2(0.f), splat2(0.f), splat2(0.f)}, sacc[3] = {splat2(0.f), splat2(0.f), splat2(0.f)};

  auto row_loop = [&](auto ortho_c) {
    constexpr bool ORTHO = decltype(ortho_c)::value;
    for (int vr = 0; vr < nvr; ++vr) {
      const int e = Q == 1 ? vr : (vr >> 1), aoff = Q == 1 ? 0 : (vr & 1) * NP;      // table row; first azimuth pair of this virtual row
      const float* cur = tile + (vr & 1) * kT32Floats;
      if (vr + 1 < nvr) {
        tile32_dma_issue_vrow<SGR_DMA_AUX, EW>(tile + ((vr + 1) & 1) * kT32Floats, eimg, x.p0, RC, a.J, vr + 1, lane);
        wait_vmcnt<6>();        // this virtual row has landed; the next stays in flight
      } else {
        wait_vmcnt<0>();
      }
      if (!ORTHO) fence_row_invariants(q);
      const RowCtx rc = make_row_ctx(q, rows[e], true);
      OrthoRow orow = make_ortho_row(rc.ro, q.vv);
#pragma unroll 1
      for (int ap = 0; ap < NP; ++ap) {
        if (ORTHO) { SGR_FENCE2(oq.vB); SGR_FENCE2(oq.ff); SGR_FENCE2(orow.n2c); SGR_FENCE2(orow.cvc); }
        const f32x4 cs = cpt[aoff + ap];
        const f32x2 ca = {cs[0], cs[1]}, sa = {cs[2], cs[3]};
        float g[3][2];
        tile32_read_pair(cur, pl, own * 8 + ap * 2, g);
        const f32x2 Pv = pfma(SGR_HI(oq.vB), sa, SGR_LO(oq.vB) * ca);
        f32x2 wt, sp;
        shade_pair<ORTHO>(q, oq, rc, orow, own, ca, sa, Pv, xt, (aoff + ap) * 2, wt, sp);
        const f32x2 sw = sp * wt;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const f32x2 gv = {g[c][0], g[c][1]};
          dacc[c] = pfma(wt, gv, dacc[c]);
          sacc[c] = pfma(sw, gv, sacc[c]);
        }
      }
    }
  };
  if (ortho) row_loop(std::true_type{}); else row_loop(std::false_type{});

  // each half integrated one half row: add the two
  float v[6] = {dacc[0].x + dacc[0].y, dacc[1].x + dacc[1].y, dacc[2].x + dacc[2].y, sacc[0].x + sacc[0].y, sacc[1].x + sacc[1].y, sacc[2].x + sacc[2].y};
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

// ============================== dL/dEnv of forwardEnv alone, packed half-wave (round 4) ============
// dL/dEnv[c,j] = omega_j ndl_j (gD_c A_c/pi + gS_c spec_j) (adjoint of models.py:511-520): what the two-call drop-in sequence's backward
// writes before sg_to_env's backward reads it.  Write-bound (1536 of 1672 B per shaded pixel go out); the generic kernel (one pixel per
// lane, world-space terms per direction, 64-byte segments) reached 54 % of the HBM peak.  Same shape as the half-wave forward: each half
// evaluates the BRDF terms of the half row it owns in azimuth pairs and writes them into the 32-pixel tile, which is flushed in whole
// 128-byte lines (two table rows on the 8x16 grid, one on 16x32).
template <int POOL, int EW, int RPF>
__global__ __launch_bounds__(kWave, 3) void render_genv_pk_half_kernel(const Args a) {
  constexpr int HALF = EW / 2, NQ = HALF / 4, TD = EW * RPF;
  __shared__ __attribute__((aligned(16))) float tile[T32Out<TD>::kFloats];
  const Pix x = locate_group32(a, (int)blockIdx.x);
  const int lane = threadIdx.x, half = lane >> 5, pl = lane & 31;
  const int own = 1 - half;
  const int RC = a.R * a.C, b = x.b, p = x.p;
  float alb[3];
  const Frame f = load_frame<POOL>(a, x, alb);
  PixLocal q = make_local(f, a.F0);
  OrthoPix oq = make_ortho_pix(q);
  const bool ortho = __all(frame_is_orthonormal(q));
  f32x2 gds[3];                                     // (gD_c A_c / pi, gS_c)
  {
    const size_t o = (size_t)b * 3 * RC;
    const unsigned up = (unsigned)p;
#pragma unroll
    for (int c = 0; c < 3; ++c)
      gds[c] = f32x2{(a.g_diffuse + o + (size_t)c * RC)[up] * (alb[c] * kInvPi), (a.g_spec + o + (size_t)c * RC)[up]};
  }
  const SepTable rows = as_sep_table(a.rows);
  const PairTable cpt = as_pair_table(a.cols, EW);
  const XTable xt = (XTable)(a.cols + EW);
  const size_t img = (size_t)b * 3 * RC * a.J;
  const int eh = a.eh;

  auto row_loop = [&](auto ortho_c) {
    constexpr bool ORTHO = decltype(ortho_c)::value;
    for (int e = 0; e < eh; ++e) {
      if (!ORTHO) fence_row_invariants(q);
      const RowCtx rc = make_row_ctx(q, rows[e], true);
      OrthoRow orow = make_ortho_row(rc.ro, q.vv);
#pragma unroll 1
      for (int aq = 0; aq < NQ; ++aq) {
        if (ORTHO) { SGR_FENCE2(oq.vB); SGR_FENCE2(oq.ff); SGR_FENCE2(orow.n2c); SGR_FENCE2(orow.cvc); }
#pragma unroll
        for (int c = 0; c < 3; ++c) SGR_FENCE2(gds[c]);
        const f32x4 t0 = cpt[2 * aq], t1 = cpt[2 * aq + 1];
        const f32x2 ca[2] = {f32x2{t0[0], t0[1]}, f32x2{t1[0], t1[1]}}, sa[2] = {f32x2{t0[2], t0[3]}, f32x2{t1[2], t1[3]}};
        f32x2 val[3][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x2 Pv = pfma(SGR_HI(oq.vB), sa[h], SGR_LO(oq.vB) * ca[h]);
          f32x2 wt, sp;
          shade_pair<ORTHO>(q, oq, rc, orow, own, ca[h], sa[h], Pv, xt, aq * 4 + 2 * h, wt, sp);
#pragma unroll
          for (int c = 0; c < 3; ++c) val[c][h] = wt * pfma(SGR_HI(gds[c]), sp, SGR_LO(gds[c]));
        }
        const float e0[4] = {val[0][0].x, val[0][0].y, val[0][1].x, val[0][1].y};
        const float e1[4] = {val[1][0].x, val[1][0].y, val[1][1].x, val[1][1].y};
        const float e2[4] = {val[2][0].x, val[2][0].y, val[2][1].x, val[2][1].y};
        tile32_write4<TD>(tile, pl, (e % RPF) * EW + own * HALF + aq * 4, e0, e1, e2);
      }
      if ((e + 1) % RPF == 0 || e + 1 == eh) {
        __syncthreads();
        tile32_store_global<TD>(tile, a.g_env_out + img, x.p0, RC, a.J, (e / RPF) * TD, (e % RPF + 1) * EW, lane);
        __syncthreads();
      }
    }
  };
  if (ortho) row_loop(std::true_type{}); else row_loop(std::false_type{});
}

template <int POOL, bool HAS_GENV, bool HAS_RENDER, int EW = 16, bool HEADS = false>
__global__ __launch_bounds__(kWave, 2) void sg_bwd_pk_kernel(const Args a) {
  constexpr int HALF = 8, NP = 4, KPW = 6, Q = EW / 16;
  // env cotangent rows: a ring of three one-row LDS-DMA buffers (18 KB).  Rows are requested two at a time, back to back
  // (the two 64-byte halves of every 128-byte line of the image), one row ahead of their use; non-temporal, since the
  // cotangent is read exactly once and must not push the SG parameters out of the Infinity Cache
  __shared__ __attribute__((aligned(16))) float tile[HAS_GENV ? 3 * kT32Floats : 4];
  SGR_TRACE_BEGIN

  const int lane = threadIdx.x, half = lane >> 5, pl = lane & 31;
  const int own = 1 - half;                         // the half row (sign) whose BRDF terms this half-wave evaluates
  const int RC = a.R * a.C, K = a.K;
  // workgroup id -> (32-pixel group t, group of 12 lobes): ids [8 ng m, 8 ng m + 8) are lobe group 0 of pixel groups
  // 8 m .. 8 m + 7, the next 8 ids lobe group 1 of the same pixel groups, ...  (ng = 1: id == t)
  const int ng = (K + 2 * KPW - 1) / (2 * KPW);
  const int chunk = (int)blockIdx.x / (8 * ng), within = (int)blockIdx.x - chunk * (8 * ng);
  const int grp = within >> 3, t = chunk * 8 + (within & 7);
  if (t >= a.bn * ((RC + kPx - 1) / kPx)) return;
  const Pix x = locate_group32(a, t);
  const int b = x.b, p = x.p;
  const int nvr = a.eh * Q;                         // virtual rows

  __amdgpu_buffer_rsrc_t gimg = env_rsrc(HAS_GENV ? a.g_env + (size_t)b * 3 * RC * a.J : a.view, RC, a.J);
  if (HAS_GENV && !(SGR_ABLATE & 1)) {
    tile32_dma_issue_vrow<SGR_PK_BWD_AUX, EW>(tile, gimg, x.p0, RC, a.J, 0, lane);
    if (nvr > 1) tile32_dma_issue_vrow<SGR_PK_BWD_AUX, EW>(tile + kT32Floats, gimg, x.p0, RC, a.J, 1, lane);
  }

  PixLocal q;
  OrthoPix oq;
  bool ortho = true;
  f32x2 gds[3] = {splat2(0.f), splat2(0.f), splat2(0.f)};     // (gD_c A_c / pi, gS_c)
  if (HAS_RENDER) {
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

  // axes pre-multiplied by lp = lam * log2e (floored, see load_lobes_pk): the exponents t' = lp t come straight out of the
  // packed FMAs, and the sharpness gradient is accumulated as sum T t' = lp sum T t and divided by lp at the end -- two
  // packed multiplies fewer per lobe and azimuth pair
  LobesPk<KPW> P;
  load_lobes_pk<KPW, true, HEADS>(a, b, (unsigned)p, x.active, grp * 2 * KPW + half * KPW, P, false);

  f32x2 gw0[KPW], gw1[KPW], gw2[KPW], gz[KPW], gx[KPW], gy[KPW];
#pragma unroll
  for (int k = 0; k < KPW; ++k) gw0[k] = gw1[k] = gw2[k] = gz[k] = gx[k] = gy[k] = splat2(0.f);

  const SepTable rows = as_sep_table(a.rows);
  const PairTable cpt = as_pair_table(a.cols, EW);
  const XTable xt = (XTable)(a.cols + EW);
  SGR_TRACE_MARK

  auto row_loop = [&](auto ortho_c) {
    constexpr bool ORTHO = decltype(ortho_c)::value;
    for (int vr = 0; vr < nvr; ++vr) {
      const int e = Q == 1 ? vr : (vr >> 1), aoff = Q == 1 ? 0 : (vr & 1) * NP;      // table row; first azimuth pair of this virtual row
      const float* cur = tile + (HAS_GENV ? (vr % 3) * kT32Floats : 0);
      if (HAS_GENV && !(SGR_ABLATE & 1)) {
        // rows vr+1, vr+2 (vr odd) were requested when row vr-1 was done; up to two rows (12 instructions) may stay in flight
        if ((vr & 1) == 0) {
          if (vr + 1 < nvr) wait_vmcnt<6>(); else wait_vmcnt<0>();          // in flight at most: row vr+1
        } else {
          if (vr + 2 < nvr) wait_vmcnt<12>(); else if (vr + 1 < nvr) wait_vmcnt<6>(); else wait_vmcnt<0>();   // rows vr+1, vr+2
        }
      }
      if (HAS_RENDER && !ORTHO) fence_row_invariants(q);
      const f32x8 row = rows[e];
      const float sr = row[0], cr = row[1];
      f32x2 czr[KPW / 2];
#pragma unroll
      for (int m = 0; m < KPW / 2; ++m) czr[m] = pfma(P.azp[m], splat2(cr), -P.lpp[m]);      // lp (az c_e - 1)
      const RowCtx rc = make_row_ctx(q, row, HAS_RENDER);
      OrthoRow orow = make_ortho_row(rc.ro, q.vv);

#pragma unroll 1
      for (int ap = 0; ap < NP; ++ap) {
        fence_lobes<KPW>(P);
#pragma unroll
        for (int m = 0; m < KPW / 2; ++m) { SGR_FENCE2(czr[m]); }
#if SGR_PK_BWD_FENCES >= 2
#pragma unroll
        for (int m = 0; m < KPW / 2; ++m) { SGR_FENCE2(P.lpp[m]); }
        if (HAS_RENDER) {
#pragma unroll
          for (int c = 0; c < 3; ++c) SGR_FENCE2(gds[c]);
          if (ORTHO) { SGR_FENCE2(oq.vB); SGR_FENCE2(oq.ff); SGR_FENCE2(orow.n2c); SGR_FENCE2(orow.cvc); }
        }
#endif
        // (round 4, measured and not adopted: this pair's table entries requested one iteration ahead, and the cotangent pairs read behind
        // the BRDF terms instead of in front of them -- the loop head then waits for nothing -- 232.8 vs 228.8-230.1 us in the bench loop
        // on one box, profiles/r04d_variants.txt: at two waves per SIMD the other wave already covers the ~150 cycles)
        const f32x4 cs = cpt[aoff + ap];
        const f32x2 ca = {cs[0], cs[1]}, sa = {cs[2], cs[3]};
        f32x2 g[2][3];            // [sign][colour], the azimuth pair (2 ap, 2 ap + 1) of this virtual row
        if (HAS_GENV && !(SGR_ABLATE & 2)) {
          tile32_read_two_pairs2(cur, pl, ap * 2, HALF + ap * 2, g[0], g[1]);
        } else if (HAS_GENV) {      // ablation: cotangents that are neither loaded nor constant-foldable
#pragma unroll
          for (int c = 0; c < 3; ++c) g[0][c] = g[1][c] = splat2(1e-3f * (float)(ap + c + lane));
          (void)cur;
        } else {
#pragma unroll
          for (int c = 0; c < 3; ++c) g[0][c] = g[1][c] = splat2(0.f);
        }
        if (HAS_RENDER) {
          // the render term of the half row this half-wave owns, then both halves' terms to all lanes
          const f32x2 Pv = pfma(SGR_HI(oq.vB), sa, SGR_LO(oq.vB) * ca);
          f32x2 wt, sp;
          shade_pair<ORTHO>(q, oq, rc, orow, own, ca, sa, Pv, xt, (aoff + ap) * 2, wt, sp);
          // what travels between the halves is the specular term alone (2 swaps; the weight as well for degenerate frames,
          // where it depends on the direction): the cotangent  wt (gD A/pi + gS spec)  of both half rows is then formed by
          // every lane -- 12 packed instructions either way, four (two) v_permlane32_swap fewer than trading the six products
          float dx = sp.x, sx = sp.x, dy = sp.y, sy = sp.y;
          swap32(dx, sx);
          swap32(dy, sy);
          const f32x2 sp1 = {dx, dy}, sp0 = {sx, sy};      // half row 1: evaluated by lanes 0..31; half row 0: by lanes 32..63
          f32x2 wt1 = wt, wt0 = wt;
          if (!ORTHO) {
            float ux = wt.x, vx = wt.x, uy = wt.y, vy = wt.y;
            swap32(ux, vx);
            swap32(uy, vy);
            wt1 = f32x2{ux, uy}; wt0 = f32x2{vx, vy};
          }
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            g[1][c] = pfma(wt1, pfma(SGR_HI(gds[c]), sp1, SGR_LO(gds[c])), g[1][c]);
            g[0][c] = pfma(wt0, pfma(SGR_HI(gds[c]), sp0, SGR_LO(gds[c])), g[0][c]);
          }
        }
        const f32x2 srv = splat2(sr);
        const f32x2 sca = srv * ca, ssa = srv * sa;
#pragma unroll
        for (int k = 0; k < KPW; ++k) {
          const f32x2 cz = half_of(czr[k / 2], k & 1), w2 = half_of(P.w2p[k / 2], k & 1);
          const f32x2 u = pfma(SGR_HI(P.axy[k]), sa, SGR_LO(P.axy[k]) * ca);
          const f32x2 tp = pfma(srv, u, cz), tm = pfma(-srv, u, cz);      // lp t
          const f32x2 ep = {fexp2(tp.x), fexp2(tp.y)}, em = {fexp2(tm.x), fexp2(tm.y)};
          gw0[k] = pfma(g[0][0], ep, gw0[k]); gw1[k] = pfma(g[0][1], ep, gw1[k]); gw2[k] = pfma(g[0][2], ep, gw2[k]);
          gw0[k] = pfma(g[1][0], em, gw0[k]); gw1[k] = pfma(g[1][1], em, gw1[k]); gw2[k] = pfma(g[1][2], em, gw2[k]);
          // T+- = (g . w) E+-  enter through their sum and difference alone (no sum T t: see the epilogue) -- 3 packed instructions
          const f32x2 Sp = pfma(g[0][2], w2, pfma(g[0][1], SGR_HI(P.w01[k]), g[0][0] * SGR_LO(P.w01[k])));
          const f32x2 Sm = pfma(g[1][2], w2, pfma(g[1][1], SGR_HI(P.w01[k]), g[1][0] * SGR_LO(P.w01[k])));
          const f32x2 Tm = Sm * em;
          const f32x2 Ts = pfma(Sp, ep, Tm), Td = pfma(Sp, ep, -Tm);
          gz[k] = pfma(splat2(cr), Ts, gz[k]);
          gx[k] = pfma(sca, Td, gx[k]);
          gy[k] = pfma(ssa, Td, gy[k]);
        }
      }
      if (HAS_GENV && !(SGR_ABLATE & 1) && (vr & 1) == 0) {
        // row vr is consumed: its buffer and the one of row vr-1 are free -> request rows vr+2 and vr+3 back to back
        if (vr + 2 < nvr) tile32_dma_issue_vrow<SGR_PK_BWD_AUX, EW>(tile + ((vr + 2) % 3) * kT32Floats, gimg, x.p0, RC, a.J, vr + 2, lane);
        if (vr + 3 < nvr) tile32_dma_issue_vrow<SGR_PK_BWD_AUX, EW>(tile + ((vr + 3) % 3) * kT32Floats, gimg, x.p0, RC, a.J, vr + 3, lane);
      }
    }
  };
  if (ortho) row_loop(std::true_type{}); else row_loop(std::false_type{});

  if (x.active) {
    // same addressing as the loads: wave-uniform plane base of lobe slot k of lobe group `grp` + the lane's byte offset
    const unsigned o3_own = ((unsigned)(half * KPW * 3 * RC) + (unsigned)p) * 4u, o1_own = ((unsigned)(half * KPW * RC) + (unsigned)p) * 4u;
    char* g_axis_b = reinterpret_cast<char*>(a.g_axis + (size_t)b * K * 3 * RC);
    char* g_lamb_b = reinterpret_cast<char*>(a.g_lamb + (size_t)b * K * RC);
    char* g_weight_b = reinterpret_cast<char*>(a.g_weight + (size_t)b * K * 3 * RC);
#pragma unroll
    for (int k = 0; k < KPW; ++k) {
      const int ks = grp * 2 * KPW + k, kk = ks + half * KPW;      // ks: wave-uniform slot; kk: this lane's lobe
      if (kk < K) {
        const float lpk = (k & 1) ? P.lpp[k / 2].y : P.lpp[k / 2].x;
        const float w0 = P.w01[k].x, w1 = P.w01[k].y, w2 = (k & 1) ? P.w2p[k / 2].y : P.w2p[k / 2].x;
        const float lam = fabsf(lpk) <= kLpFloor ? 0.0f : lpk * kLn2;      // the floor stands for lam == 0
        float q0 = gw0[k].x + gw0[k].y, q1 = gw1[k].x + gw1[k].y, q2 = gw2[k].x + gw2[k].y;
        const float sx = gx[k].x + gx[k].y, sy = gy[k].x + gy[k].y, sz = gz[k].x + gz[k].y;
        float glk = sharpness_grad(P.axy[k].x, P.axy[k].y, (k & 1) ? P.azp[k / 2].y : P.azp[k / 2].x, lpk, sx, sy, sz, w0, w1, w2, q0, q1, q2);
        if (HEADS || a.premap) {
          glk *= premap_grad(lam);
          q0 *= premap_grad(w0); q1 *= premap_grad(w1); q2 *= premap_grad(w2);
        }
        float gax = lam * sx, gay = lam * sy, gaz = lam * sz;
        if (HEADS)
          heads_bwd_lobe(reinterpret_cast<const char*>(a.axis + (size_t)b * K * 3 * RC) + (size_t)ks * 3 * RC * 4,
                         reinterpret_cast<const char*>(a.lamb + (size_t)b * K * RC) + (size_t)ks * RC * 4,
                         reinterpret_cast<const char*>(a.weight + (size_t)b * K * 3 * RC) + (size_t)ks * 3 * RC * 4, o3_own, o1_own,
                         (size_t)RC * 4, gax, gay, gaz, glk, q0, q1, q2);
        char* pa = g_axis_b + (size_t)ks * 3 * RC * 4;
        char* pw = g_weight_b + (size_t)ks * 3 * RC * 4;
        *reinterpret_cast<float*>(pa + o3_own) = gax;
        *reinterpret_cast<float*>(pa + (size_t)RC * 4 + o3_own) = gay;
        *reinterpret_cast<float*>(pa + (size_t)RC * 8 + o3_own) = gaz;
        *reinterpret_cast<float*>(g_lamb_b + (size_t)ks * RC * 4 + o1_own) = glk;
        *reinterpret_cast<float*>(pw + o3_own) = q0;
        *reinterpret_cast<float*>(pw + (size_t)RC * 4 + o3_own) = q1;
        *reinterpret_cast<float*>(pw + (size_t)RC * 8 + o3_own) = q2;
      }
    }
  }
  SGR_TRACE_END
}

}  // namespace sgr
