// Per-pixel / per-direction arithmetic of the SG x microfacet render path (fp32).
//
// Everything here is `__host__ __device__` so that the exact expressions the gfx950
// kernels evaluate can also be compiled with g++ and checked on a GPU-less machine
// against the oracle (tests/host_emul/, test infrastructure only -- the product has
// no CPU path).  On the device the transcendental wrappers map to single CDNA4
// instructions (v_exp_f32, v_rcp_f32, v_rsq_f32).
//
// Reference semantics restated here (file:line relative to /root/reference):
//   pre-map tan(pi/2 * 0.999 * x) ............ models.py:396-400
//   SG lobe  w * exp(lam * (a.l - 1)) ......... models.py:378-387
//   normal renormalisation, local frame ...... models.py:465-484
//   half vector, Fresnel, GGX, Smith, clamp .. models.py:486-509
//   quadrature terms .......................... models.py:511-520
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SGR_HD __host__ __device__ __forceinline__
#else
#define SGR_HD inline
#endif

namespace sgr {

constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kLn2 = 0.69314718055994530942f;
constexpr float kPi = 3.14159265358979323846f;
constexpr float kInvPi = 0.31830988618379067154f;
constexpr float kFourPi = 12.566370614359172f;          // fp32(4*np.pi), models.py:508
constexpr float kHalfPiHi = 1.57079637050628662109375f;  // fp32(np.pi/2), models.py:397,400
constexpr float kPremapScale = 0.999f;                   // models.py:396,399

// ---- single-instruction transcendentals -----------------------------------------
SGR_HD float fexp2(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_exp2f(x);   // v_exp_f32 (results below 2^-126 flush to 0)
#else
  return exp2f(x);
#endif
}
SGR_HD float frcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcpf(x);    // v_rcp_f32, 1 ulp
#else
  return 1.0f / x;
#endif
}
SGR_HD float frsq(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rsqf(x);    // v_rsq_f32, 1 ulp
#else
  return 1.0f / sqrtf(x);
#endif
}
// One Newton step on top of the 1-ulp hardware estimates (~0.5 ulp afterwards): used where a result feeds a difference of
// nearly equal terms -- the BRDF-map adjoints, whose per-direction contributions cancel across the hemisphere to 1e-3 of
// their size, so that 1-ulp errors in 1/|h| and 1/nom alone cost a factor 2-3 against the exact-division host evaluation
// of the same formulas (measured, round 3: roughness gradient 8.7e-4 / 1.1e-3 vs 4.9e-4 / 3.9e-4 on fixtures g8 / g7).
SGR_HD float frcp_nr(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const float y = __builtin_amdgcn_rcpf(x);
  return fmaf(fmaf(-x, y, 1.0f), y, y);
#else
  return 1.0f / x;
#endif
}
SGR_HD float frsq_nr(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const float y = __builtin_amdgcn_rsqf(x);
  return fmaf(fmaf(-0.5f * x * y, y, 0.5f), y, y);
#else
  return 1.0f / sqrtf(x);
#endif
}
// Individually rounded multiply / add.  On the device these are inline asm: HIP's __fmul_rn/__fadd_rn
// are plain a*b / a+b, which -ffp-contract=fast is free to fuse into an FMA.
SGR_HD float fmul_rn(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
  float r;
  asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
#else
  volatile float r = a * b;           // keep the compiler from contracting / reassociating
  return r;
#endif
}
SGR_HD float fadd_rn(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
  float r;
  asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
#else
  volatile float r = a + b;
  return r;
#endif
}
SGR_HD float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
SGR_HD float clamp01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

// ---- SG pre-map -------------------------------------------------------------------
// y = fl(fl(0.999f * x) * fl(pi/2)) exactly as torch evaluates models.py:396-400 in fp32
// (the argument rounding is amplified ~1000x by tan near pi/2, so it has to be the same
// rounding as the reference's); then tan(y) to <= ~2 ulp, branch-free:
//   n = rint(y * 2/pi),  z = y - n*pi/2  (three-constant Cody-Waite, exact for small n),
//   tan(y) = tan(z) for even n, -rcp(tan(z)) for odd n,  tan(z) on |z| <= pi/4 by the Cephes
//   single-precision odd polynomial.
// Decoder outputs are clamped to [0,1] (models.py:338-340), i.e. y in [0, 1.5692] and n in
// {0,1}; for |y| beyond ~1e4 the reduction loses accuracy (far outside the layer's domain).
constexpr float kTwoOverPi = 0.63661977236758134308f;
constexpr float kPio2_1 = 1.5703125f;                   // pi/2 split: 8 + 11 + 24 significant bits
constexpr float kPio2_2 = 4.837512969970703125e-4f;
constexpr float kPio2_3 = 7.54978995489188e-8f;

SGR_HD float tan_kernel(float z) {   // |z| <= pi/4
  const float zz = z * z;
  float p = 9.38540185543e-3f;
  p = fmaf(p, zz, 3.11992232697e-3f);
  p = fmaf(p, zz, 2.44301354525e-2f);
  p = fmaf(p, zz, 5.34112807005e-2f);
  p = fmaf(p, zz, 1.33387994085e-1f);
  p = fmaf(p, zz, 3.33331568548e-1f);
  return fmaf(p * zz, z, z);
}
SGR_HD float premap_arg(float x) { return fmul_rn(fmul_rn(kPremapScale, x), kHalfPiHi); }
SGR_HD float tan_f32(float y) {
  const float n = rintf(y * kTwoOverPi);
  float z = fmaf(-n, kPio2_1, y);
  z = fmaf(-n, kPio2_2, z);
  z = fmaf(-n, kPio2_3, z);
  const float t = tan_kernel(z);
  const bool odd = ((int)n) & 1;
  return odd ? -frcp(t) : t;      // v_rcp_f32 (1 ulp) instead of the ten-instruction IEEE division: -2..5 % kernel time (48 pre-maps per pixel)
}
SGR_HD float premap(float x) { return tan_f32(premap_arg(x)); }
// d premap / dx given y_tan = premap(x):  0.999 * pi/2 * (1 + tan^2)
SGR_HD float premap_grad(float y_tan) { return (kPremapScale * kHalfPiHi) * fmaf(y_tan, y_tan, 1.0f); }

// ---- per-pixel shading frame ------------------------------------------------------
struct Frame {
  float nx, ny, nz;      // renormalised normal
  float cxx, cxy, cxz;   // camx
  float cyx, cyy, cyz;   // camy
  float vx, vy, vz;      // view vector
  float ndv;             // clamp(N.v, 0, 1)
  float alpha2;          // ((rho+1)/2)^4
  float k;               // ((rho+1)/2 + 1)^2 / 8
  float nom1;            // ndv * (1-k) + k
};

// Inputs are the *pooled* (average over the q image pixels) normal and roughness.
SGR_HD Frame make_frame(float pnx, float pny, float pnz, float prho, float vx, float vy, float vz) {
  Frame f;
  // N / sqrt(clamp(|N|^2, 1e-6, 1))                                   models.py:467-468
  // |N|^2 is evaluated exactly as torch.sum(N*N, dim=1) does in fp32 -- three rounded products,
  // added left to right, no FMA -- because the two-sided clamp makes the normal's gradient
  // discontinuous at |N|^2 == 1 and unit input normals (ratio-1 maps) sit right on that kink: the
  // side of the kink has to be decided with the reference's own rounding.
  const float nn = fadd_rn(fadd_rn(fmul_rn(pnx, pnx), fmul_rn(pny, pny)), fmul_rn(pnz, pnz));
  const float inv = frsq_nr(clampf(nn, 1e-6f, 1.0f));
  f.nx = pnx * inv; f.ny = pny * inv; f.nz = pnz * inv;
  // camy = normalize(up - (up.N) N), up = (0,1,0)                     models.py:477-478
  const float proj = f.ny;
  float ax = -proj * f.nx, ay = 1.0f - proj * f.ny, az = -proj * f.nz;
  float nrm = sqrtf(ax * ax + ay * ay + az * az);
  float s = 1.0f / fmaxf(nrm, 1e-12f);
  f.cyx = ax * s; f.cyy = ay * s; f.cyz = az * s;
  // camx = -normalize(cross(camy, N))                                 models.py:479
  float bx = f.cyy * f.nz - f.cyz * f.ny;
  float by = f.cyz * f.nx - f.cyx * f.nz;
  float bz = f.cyx * f.ny - f.cyy * f.nx;
  nrm = sqrtf(bx * bx + by * by + bz * bz);
  s = -1.0f / fmaxf(nrm, 1e-12f);
  f.cxx = bx * s; f.cxy = by * s; f.cxz = bz * s;
  f.vx = vx; f.vy = vy; f.vz = vz;
  // roughness terms                                                   models.py:494-498
  const float r = (prho + 1.0f) * 0.5f;
  f.k = (r + 1.0f) * (r + 1.0f) * 0.125f;
  const float alpha = r * r;
  f.alpha2 = alpha * alpha;
  f.ndv = clamp01(f.nx * vx + f.ny * vy + f.nz * vz);               // models.py:500
  f.nom1 = f.ndv * (1.0f - f.k) + f.k;                                // models.py:506
  return f;
}

// One quadrature direction: ndl = clamp(N.l,0,1), spec = alpha^2 F / nom.   models.py:481-509
SGR_HD void brdf_dir(const Frame& f, float lx, float ly, float lz, float F0, float& ndl, float& spec) {
  const float wx = lx * f.cxx + ly * f.cyx + lz * f.nx;
  const float wy = lx * f.cxy + ly * f.cyy + lz * f.ny;
  const float wz = lx * f.cxz + ly * f.cyz + lz * f.nz;
  float hx = (f.vx + wx) * 0.5f, hy = (f.vy + wy) * 0.5f, hz = (f.vz + wz) * 0.5f;
  const float hinv = frsq(fmaxf(hx * hx + hy * hy + hz * hz, 1e-6f));
  hx *= hinv; hy *= hinv; hz *= hinv;
  const float vdh = f.vx * hx + f.vy * hy + f.vz * hz;
  const float fres = F0 + (1.0f - F0) * fexp2((-5.55472f * vdh - 6.98316f) * vdh);
  const float ndh = clamp01(f.nx * hx + f.ny * hy + f.nz * hz);
  ndl = clamp01(f.nx * wx + f.ny * wy + f.nz * wz);
  const float nom0 = ndh * ndh * (f.alpha2 - 1.0f) + 1.0f;
  const float nom2 = ndl * (1.0f - f.k) + f.k;
  const float nom = clampf(kFourPi * nom0 * nom0 * f.nom1 * nom2, 1e-6f, kFourPi);
  spec = f.alpha2 * fres * frcp(nom);
}

// ---- local-frame ("factorised") form of the quadrature direction -----------------------------
// The hemisphere table is a tensor product: l_j = (s_e ca_a, s_e sa_a, c_e), j = e*ew + a
// (models.py:353-363), and every per-direction quantity of models.py:481-509 depends on the
// world-space l only through dot products with per-pixel vectors:
//   v.l = s_e (vB.x ca_a + vB.y sa_a) + vB.z c_e          with vB = (v.camx, v.camy, v.N)
//   N.l = s_e (nB.x ca_a + nB.y sa_a) + nB.z c_e          with nB = (N.camx, N.camy, N.N)
//   |l|^2 = s_e^2 Q_a + 2 s_e c_e R_a + c_e^2 G_zz        (Gram matrix of (camx, camy, N); == 1 for
//                                                          an orthonormal frame, kept general so the
//                                                          degenerate frames of models.py:467-479 --
//                                                          N || up, |N| != 1 -- follow the reference)
//   |v + l|^2 = |v|^2 + 2 v.l + |l|^2,  v.h = (|v|^2 + v.l) r,  N.h = (N.v + N.l) r,
//   r = rsqrt(max(|v+l|^2, 4e-6))      (== the reference's h / sqrt(max(|h|^2, 1e-6)), h = (v+l)/2)
// which turns ~35 world-space FMAs per direction into ~4, with the (a)-only and (e)-only parts
// hoisted out of the direction loop.
struct PixLocal {
  float vBx, vBy, vBz;     // v in the local frame
  float nBx, nBy, nBz;     // N in the local frame (nBz = |N|^2)
  float vv, nv;            // |v|^2, N.v (unclamped)
  float Gxx, Gxy, Gyy, Gxz, Gyz, Gzz;
  float alpha2m1, k, omk;  // alpha^2 - 1, k, 1 - k
  float c1;                // 4 pi * nom1
  float fa, fb;            // alpha^2 F0, alpha^2 (1 - F0)
};
SGR_HD PixLocal make_local(const Frame& f, float F0) {
  PixLocal q;
  q.vBx = f.vx * f.cxx + f.vy * f.cxy + f.vz * f.cxz;
  q.vBy = f.vx * f.cyx + f.vy * f.cyy + f.vz * f.cyz;
  q.vBz = f.vx * f.nx + f.vy * f.ny + f.vz * f.nz;
  q.nBx = f.nx * f.cxx + f.ny * f.cxy + f.nz * f.cxz;
  q.nBy = f.nx * f.cyx + f.ny * f.cyy + f.nz * f.cyz;
  q.nBz = f.nx * f.nx + f.ny * f.ny + f.nz * f.nz;
  q.vv = f.vx * f.vx + f.vy * f.vy + f.vz * f.vz;
  q.nv = q.vBz;
  q.Gxx = f.cxx * f.cxx + f.cxy * f.cxy + f.cxz * f.cxz;
  q.Gxy = f.cxx * f.cyx + f.cxy * f.cyy + f.cxz * f.cyz;
  q.Gyy = f.cyx * f.cyx + f.cyy * f.cyy + f.cyz * f.cyz;
  q.Gxz = q.nBx;
  q.Gyz = q.nBy;
  q.Gzz = q.nBz;
  q.alpha2m1 = f.alpha2 - 1.0f;
  q.k = f.k;
  q.omk = 1.0f - f.k;
  q.c1 = kFourPi * f.nom1;
  q.fa = f.alpha2 * F0;
  q.fb = f.alpha2 * (1.0f - F0);
  return q;
}
// vdl = v.l, ndl_raw = N.l (unclamped), ll = |l|^2 (all in the factorised form above)
SGR_HD void brdf_local_dir(const PixLocal& q, float vdl, float ndl_raw, float ll, float& ndl, float& sp) {
  const float hh4 = fmaf(2.0f, vdl, q.vv + ll);
  const float r4 = frsq(fmaxf(hh4, 4e-6f));
  const float vdh = (q.vv + vdl) * r4;
  const float ndh = clamp01((q.nv + ndl_raw) * r4);
  const float pw = fexp2((-5.55472f * vdh - 6.98316f) * vdh);
  ndl = clamp01(ndl_raw);
  const float nom0 = fmaf(ndh * ndh, q.alpha2m1, 1.0f);
  const float nom2 = fmaf(ndl, q.omk, q.k);
  const float nom = clampf((nom0 * nom0) * (q.c1 * nom2), 1e-6f, kFourPi);
  sp = fmaf(q.fb, pw, q.fa) * frcp(nom);
}

// Orthonormal frame (every non-degenerate pixel: |N| = 1, N not parallel to up): N.l = c_e and the
// numerically delicate GGX term is evaluated without cancellation.  With w = v + l in local
// coordinates (tx, ty, nw):  |w|^2 = tx^2 + ty^2 + nw^2,  N.h = nw/|w|,  and
//     nom0 = 1 + ndh^2 (alpha^2 - 1) = ( tx^2 + ty^2 + [nw<0] nw^2 + alpha^2 max(nw,0)^2 ) / |w|^2
// is a sum of non-negative terms, where the reference's fp32 form (models.py:504) subtracts two
// numbers that agree to 7 digits when the half vector is near the normal and alpha is small.  Same
// function, smaller error than the reference's own against fp64.
SGR_HD bool frame_is_orthonormal(const PixLocal& q) {
  const float tol = 2e-6f;
  return fabsf(q.Gxx - 1.0f) < tol && fabsf(q.Gyy - 1.0f) < tol && fabsf(q.Gzz - 1.0f) < tol &&
         fabsf(q.Gxy) < tol && fabsf(q.Gxz) < tol && fabsf(q.Gyz) < tol;
}
struct RowOrtho {           // per (pixel, table row) constants of the orthonormal path
  float nw;                 // N.(v + l) = vBz + c_e
  float rowc;               // [nw<0] nw^2 + alpha^2 max(nw,0)^2
  float c1n2;               // 4 pi nom1 nom2,  nom2 = c_e (1-k) + k   (ndl = c_e)
  float wt;                 // ndl * omega = c_e * omega_e
  float Cv;                 // vBz * c_e
};
SGR_HD RowOrtho make_row_ortho(const PixLocal& q, float c_e, float omega_e) {
  RowOrtho r;
  r.nw = q.vBz + c_e;
  const float neg = fminf(r.nw, 0.0f), pos = fmaxf(r.nw, 0.0f);
  r.rowc = fmaf(neg, neg, (q.alpha2m1 + 1.0f) * pos * pos);
  r.c1n2 = q.c1 * fmaf(c_e, q.omk, q.k);
  r.wt = c_e * omega_e;
  r.Cv = q.vBz * c_e;
  return r;
}
// direction (+-s_e ca_a, +-s_e sa_a, c_e): ss = +-s_e, Pv = vBx ca + vBy sa.   Returns spec.
SGR_HD float brdf_ortho_dir(const PixLocal& q, const RowOrtho& r, float ss, float ca, float sa, float Pv) {
  const float tx = fmaf(ss, ca, q.vBx), ty = fmaf(ss, sa, q.vBy);
  const float T2 = fmaf(tx, tx, ty * ty);
  const float hh4 = fmaf(r.nw, r.nw, T2);
  const float Hm = fmaxf(hh4, 4e-6f);
  const float r4 = frsq(Hm);
  const float vdh = (q.vv + fmaf(ss, Pv, r.Cv)) * r4;
  const float pw = fexp2((-5.55472f * vdh - 6.98316f) * vdh);
  const float nom0 = ((T2 + (Hm - hh4)) + r.rowc) * (r4 * r4);
  const float nom = clampf((nom0 * nom0) * r.c1n2, 1e-6f, kFourPi);
  return fmaf(q.fb, pw, q.fa) * frcp(nom);
}

// ---- adjoints of the shading frame and of one quadrature direction --------------------------
// Hand-derived reverse mode of make_frame()/brdf_dir(), following torch.autograd's conventions
// for the kinks of models.py:465-509: clamp(min,max) passes the gradient when min <= x <= max
// (inclusive), clamp(min=) when x >= min, F.normalize = x / clamp_min(||x||, eps) with a zero
// sub-gradient for the norm at x == 0.
struct FrameGrad {          // accumulated over the J directions
  float gN[3];              // d/dN   (renormalised normal)
  float gcx[3], gcy[3];     // d/dcamx, d/dcamy
  float galpha2, gk, gndv;
};
SGR_HD void frame_grad_zero(FrameGrad& g) {
  for (int i = 0; i < 3; ++i) g.gN[i] = g.gcx[i] = g.gcy[i] = 0.0f;
  g.galpha2 = g.gk = g.gndv = 0.0f;
}

// Contribution of direction (lx,ly,lz): loss term  ndl * (Ed + spec * Es)  with
//   Ed = omega_j sum_c gD_c (A_c/pi) e_cj,   Es = omega_j sum_c gS_c e_cj.
// Returns ndl (for the albedo gradient) and accumulates into g.
SGR_HD float brdf_dir_bwd(const Frame& f, float lx, float ly, float lz, float F0, float Ed, float Es, FrameGrad& g) {
  const float wx = lx * f.cxx + ly * f.cyx + lz * f.nx;
  const float wy = lx * f.cxy + ly * f.cyy + lz * f.ny;
  const float wz = lx * f.cxz + ly * f.cyz + lz * f.nz;
  const float hsx = (f.vx + wx) * 0.5f, hsy = (f.vy + wy) * 0.5f, hsz = (f.vz + wz) * 0.5f;
  const float hh = hsx * hsx + hsy * hsy + hsz * hsz;
  const float hinv = frsq_nr(fmaxf(hh, 1e-6f));
  const float hx = hsx * hinv, hy = hsy * hinv, hz = hsz * hinv;
  const float vdh = f.vx * hx + f.vy * hy + f.vz * hz;
  const float pw = fexp2((-5.55472f * vdh - 6.98316f) * vdh);
  const float fres = F0 + (1.0f - F0) * pw;
  const float ndh_raw = f.nx * hx + f.ny * hy + f.nz * hz;
  const float ndl_raw = f.nx * wx + f.ny * wy + f.nz * wz;
  const float ndh = clamp01(ndh_raw), ndl = clamp01(ndl_raw);
  const float omk = 1.0f - f.k;
  // nom0 = 1 + ndh^2 (alpha^2 - 1) without the cancellation of that form (the reference's, models.py:504) where it matters --
  // half vector near the normal, small alpha, which is where the roughness / normal gradients live: for a unit normal and
  // 0 <= N.h <= 1,   nom0 = ( |hs - (N.hs) N|^2 + alpha^2 (N.hs)^2 ) / |hs|^2 ,   the tangential part of the half vector squared
  // (a sum of squares) plus a non-negative term.  Same function; other cases (renormalisation clamps live, |hs| floored)
  // keep the literal form.
  float nom0 = ndh * ndh * (f.alpha2 - 1.0f) + 1.0f;
  {
    const float nn = f.nx * f.nx + f.ny * f.ny + f.nz * f.nz;
    const float nh = f.nx * hsx + f.ny * hsy + f.nz * hsz;
    const float tx = fmaf(-nh, f.nx, hsx), ty = fmaf(-nh, f.ny, hsy), tz = fmaf(-nh, f.nz, hsz);
    const float tt = tx * tx + ty * ty + tz * tz;
    const bool regular = fabsf(nn - 1.0f) < 4e-7f && hh >= 1e-6f && ndh_raw >= 0.0f && ndh_raw <= 1.0f;
    const float alt = fmaf(f.alpha2, nh * nh, tt) * (hinv * hinv);
    nom0 = regular ? alt : nom0;
  }
  const float nom2 = ndl * omk + f.k;
  const float nomr = kFourPi * nom0 * nom0 * f.nom1 * nom2;
  const float nom = clampf(nomr, 1e-6f, kFourPi);
  const float rn = frcp_nr(nom);
  const float spec = f.alpha2 * fres * rn;

  float gndl = Ed + spec * Es;                 // direct
  const float gsp = ndl * Es;
  g.galpha2 += gsp * fres * rn;
  const float gfres = gsp * f.alpha2 * rn;
  const float gnom = (nomr >= 1e-6f && nomr <= kFourPi) ? (-gsp * spec * rn) : 0.0f;
  const float gnom0 = gnom * (2.0f * kFourPi) * nom0 * f.nom1 * nom2;
  const float gnom1 = gnom * kFourPi * nom0 * nom0 * nom2;
  const float gnom2 = gnom * kFourPi * nom0 * nom0 * f.nom1;
  float gndh = gnom0 * 2.0f * ndh * (f.alpha2 - 1.0f);
  g.galpha2 += gnom0 * ndh * ndh;
  g.gndv += gnom1 * omk;
  g.gk += gnom1 * (1.0f - f.ndv) + gnom2 * (1.0f - ndl);
  gndl += gnom2 * omk;
  const float gvdh = gfres * (1.0f - F0) * pw * kLn2 * (-2.0f * 5.55472f * vdh - 6.98316f);
  if (!(ndh_raw >= 0.0f && ndh_raw <= 1.0f)) gndh = 0.0f;
  if (!(ndl_raw >= 0.0f && ndl_raw <= 1.0f)) gndl = 0.0f;
  // h: from N.h and v.h
  const float ghx = gndh * f.nx + gvdh * f.vx, ghy = gndh * f.ny + gvdh * f.vy, ghz = gndh * f.nz + gvdh * f.vz;
  // hs -> h = hs * rsqrt(max(hh, 1e-6))
  const float proj = (hh >= 1e-6f) ? (ghx * hx + ghy * hy + ghz * hz) : 0.0f;
  const float ghsx = hinv * (ghx - proj * hx), ghsy = hinv * (ghy - proj * hy), ghsz = hinv * (ghz - proj * hz);
  // l: from N.l and hs = (v+l)/2
  const float glx = gndl * f.nx + 0.5f * ghsx, gly = gndl * f.ny + 0.5f * ghsy, glz = gndl * f.nz + 0.5f * ghsz;
  // N: from N.h, N.l and l = lx camx + ly camy + lz N
  g.gN[0] += gndh * hx + gndl * wx + lz * glx;
  g.gN[1] += gndh * hy + gndl * wy + lz * gly;
  g.gN[2] += gndh * hz + gndl * wz + lz * glz;
  g.gcx[0] += lx * glx; g.gcx[1] += lx * gly; g.gcx[2] += lx * glz;
  g.gcy[0] += ly * glx; g.gcy[1] += ly * gly; g.gcy[2] += ly * glz;
  return ndl;
}

// Back through make_frame(): gradients w.r.t. the pooled normal and pooled roughness.
SGR_HD void frame_bwd(float pnx, float pny, float pnz, float prho, const Frame& f, const FrameGrad& g,
                      float gpn[3], float& gprho) {
  float gN[3] = {g.gN[0], g.gN[1], g.gN[2]};
  // ndv = clamp(N.v, 0, 1)
  const float ndv_raw = f.nx * f.vx + f.ny * f.vy + f.nz * f.vz;
  if (ndv_raw >= 0.0f && ndv_raw <= 1.0f) {
    gN[0] += g.gndv * f.vx; gN[1] += g.gndv * f.vy; gN[2] += g.gndv * f.vz;
  }
  const float N[3] = {f.nx, f.ny, f.nz};
  // recompute camy_raw, cr = camy x N and their norms (as in make_frame)
  const float proj = f.ny;
  const float raw[3] = {-proj * f.nx, 1.0f - proj * f.ny, -proj * f.nz};
  const float m2 = sqrtf(raw[0] * raw[0] + raw[1] * raw[1] + raw[2] * raw[2]);
  const float mc2 = fmaxf(m2, 1e-12f);
  const float cy[3] = {f.cyx, f.cyy, f.cyz};
  const float cr[3] = {cy[1] * N[2] - cy[2] * N[1], cy[2] * N[0] - cy[0] * N[2], cy[0] * N[1] - cy[1] * N[0]};
  const float m1 = sqrtf(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
  const float mc1 = fmaxf(m1, 1e-12f);
  // camx = -cr / max(|cr|, eps)
  float gcr[3];
  {
    const float dotg = g.gcx[0] * cr[0] + g.gcx[1] * cr[1] + g.gcx[2] * cr[2];
    const float gm = (m1 >= 1e-12f && m1 > 0.0f) ? (dotg / (mc1 * mc1)) / m1 : 0.0f;
    for (int i = 0; i < 3; ++i) gcr[i] = -g.gcx[i] / mc1 + gm * cr[i];
  }
  // cr = camy x N :  g_camy += N x gcr ,  gN += gcr x camy
  float gcy[3] = {g.gcy[0] + (N[1] * gcr[2] - N[2] * gcr[1]), g.gcy[1] + (N[2] * gcr[0] - N[0] * gcr[2]),
                  g.gcy[2] + (N[0] * gcr[1] - N[1] * gcr[0])};
  gN[0] += gcr[1] * cy[2] - gcr[2] * cy[1];
  gN[1] += gcr[2] * cy[0] - gcr[0] * cy[2];
  gN[2] += gcr[0] * cy[1] - gcr[1] * cy[0];
  // camy = raw / max(|raw|, eps)
  float graw[3];
  {
    const float dotg = gcy[0] * raw[0] + gcy[1] * raw[1] + gcy[2] * raw[2];
    const float gm = (m2 >= 1e-12f && m2 > 0.0f) ? (-dotg / (mc2 * mc2)) / m2 : 0.0f;
    for (int i = 0; i < 3; ++i) graw[i] = gcy[i] / mc2 + gm * raw[i];
  }
  // raw = up - N_y N
  {
    const float gdotN = graw[0] * N[0] + graw[1] * N[1] + graw[2] * N[2];
    for (int i = 0; i < 3; ++i) gN[i] -= proj * graw[i];
    gN[1] -= gdotN;
  }
  // N = pooled / sqrt(clamp(|pooled|^2, 1e-6, 1))
  {
    const float nn = fadd_rn(fadd_rn(fmul_rn(pnx, pnx), fmul_rn(pny, pny)), fmul_rn(pnz, pnz));   // as in make_frame
    const float sc = clampf(nn, 1e-6f, 1.0f);
    const float inv = frsq_nr(sc);
    const float pn[3] = {pnx, pny, pnz};
    const float dotg = gN[0] * pnx + gN[1] * pny + gN[2] * pnz;
    const float gs = (nn >= 1e-6f && nn <= 1.0f) ? (-0.5f * dotg * inv * inv * inv) : 0.0f;
    for (int i = 0; i < 3; ++i) gpn[i] = gN[i] * inv + gs * 2.0f * pn[i];
  }
  // r = (rho+1)/2, k = (r+1)^2/8, alpha2 = r^4
  const float r = (prho + 1.0f) * 0.5f;
  const float gr = g.gk * (r + 1.0f) * 0.25f + g.galpha2 * 4.0f * r * r * r;
  gprho = 0.5f * gr;
}

}  // namespace sgr
