// Scale-invariant regressions and the masked-L2 render loss on gfx950.
//
//   LSregressDiffSpec ........ models.py:23-84      LSregress ........ models.py:7-21
//   render loss .............. wrapperBRDFLight.py:170-171,192,197-207
//
// These are HBM-bound streaming reductions over a few MB (launch-latency territory at the
// reference's sizes), so the structure is: grid = (SPLIT, bn) blocks, fp32 per-thread partials,
// wave reduction by DPP shuffles, cross-wave through LDS, one partial per block written to a
// workspace; the NEXT kernel's prologue folds the SPLIT partials of its image (in a fixed order,
// in double) -- no atomics, bit-reproducible, no host synchronisation (the reference's
// `.item()` on pixelNum, wrapperBRDFLight.py:192, stays on the device).
#include "sgr_launch.h"
#include "sgr_recon_fold.h"

namespace sgr {

constexpr int kLossThreads = 256;
static_assert(kLossThreads == kRThreads, "the fold side job of stage A runs on a stage-A workgroup");
constexpr int kSplit = 16;           // blocks per image (passes over data the previous pass left in cache; 32: no change in the loop)
#ifndef SGR_LOSS_SPLIT_BC
#define SGR_LOSS_SPLIT_BC 16         // development knob: workgroups per image of the second and third pass; 64 = one partial per lane, butterfly folds.
                                     // Round 6 (the review asked for >= 2 resident waves per SIMD here), profiles/r06d_loss_stages_64_workgroups.txt: the three
                                     // launches 39.3 us against 25.1 (kbench), the with-loss step 0.405-0.407 ms against 0.393-0.399 -- every one of the 4x more
                                     // workgroups repeats the per-image folds in its prologue.  16 stays (round 3 found the same for 64 and no change for 32).
#endif
constexpr int kSplitBC = SGR_LOSS_SPLIT_BC;
static_assert(kSplitBC == kSplit || kSplitBC == 64, "the second / third pass fold their partials either sequentially (kSplit) or one per lane (64)");
constexpr int kStreamUnroll = 4;      // elements per thread and round of the streaming passes, all loads issued before the first use
#ifndef SGR_LOSS_UNROLL_BC
#define SGR_LOSS_UNROLL_BC 16
#endif
constexpr int kStreamUnrollBC = SGR_LOSS_UNROLL_BC;      // the second / third pass: 16 workgroups per image leave 14 elements per thread at config 2 -- ONE round of loads
                                                        // instead of four dependent ones (same elements per thread in the same order: same sums).  Round 6, kbench, the three
                                                        // launches: 25.1-25.5 -> 23.2 us (profiles/r06k_*); requesting that round BEFORE the per-image folds as well: 24.1-24.2, not kept
constexpr int kSplitA = 64;          // blocks per image of the FIRST pass (stage A / diffspec_partial_a): it reads the full-resolution
                                     // image and mask cold from HBM, and a quarter of the blocks left it latency-bound (16.7 us in the
                                     // training loop for 32 MB); = lanes of a wave, see fold_a

template <int N>
__device__ __forceinline__ void block_reduce(float (&v)[N], float* lds /* [4*N] */) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[i] += __shfl_down(v[i], off, 64);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) lds[wave * N + i] = v[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = (lds[i] + lds[N + i]) + (lds[2 * N + i] + lds[3 * N + i]);
  }
  __syncthreads();
}

// fold the kSplit partials of image b (N values each) in double, fixed order.  (Measured in round 3 on WARM buffers: 64 blocks per
// image for all three passes are 2 us slower than 16 -- the second and third pass read what the first left in cache and are not
// short of parallelism; the first pass is another matter in a training loop, see kSplitA.)
template <int N>
__device__ __forceinline__ void fold(const float* __restrict__ ws, int b, double (&out)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) out[i] = 0.0;
  for (int s = 0; s < kSplit; ++s) {
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] += (double)ws[((size_t)b * kSplit + s) * N + i];
  }
}

// the second pass's partials of image b (2 values each): sequentially (kSplit of them), or -- 64 of them -- one per lane and an xor butterfly
// in double like fold_a (same bits in every lane of every wave)
__device__ __forceinline__ void fold_bc(const float* __restrict__ wsB, int b, double (&out)[2]) {
  if constexpr (kSplitBC == kSplit) {
    out[0] = out[1] = 0.0;
    for (int s = 0; s < kSplit; ++s) {
      out[0] += (double)wsB[((size_t)b * kSplit + s) * 2 + 0];
      out[1] += (double)wsB[((size_t)b * kSplit + s) * 2 + 1];
    }
  } else {
    const int lane = threadIdx.x & 63;
    out[0] = (double)wsB[((size_t)b * kSplitBC + lane) * 2 + 0];
    out[1] = (double)wsB[((size_t)b * kSplitBC + lane) * 2 + 1];
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      out[0] += __shfl_xor(out[0], off, 64);
      out[1] += __shfl_xor(out[1], off, 64);
    }
  }
}

// the kSplitA stage-A partials of image b: lane l of every wave takes partial l, then an xor butterfly in double -- a fixed tree
// whose additions commute pairwise, so all lanes of all waves end with the same bits
__device__ __forceinline__ void fold_a(const float* __restrict__ wsA, int b, double (&out)[6]) {
  static_assert(kSplitA == 64, "one partial per lane");
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < 6; ++i) out[i] = (double)wsA[((size_t)b * kSplitA + lane) * 6 + i];
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
    for (int i = 0; i < 6; ++i) out[i] += __shfl_xor(out[i], off, 64);
  }
}

// (c_d, c_s) of models.py:44-63 from the five masked sums
__device__ __forceinline__ void diffspec_coefs(const double (&s)[5], float n_elems, float& cd, float& cs) {
  const float a11 = (float)s[0], a22 = (float)s[1], a12 = (float)s[2], b1 = (float)s[3], b2 = (float)s[4];
  const float frac = a11 * a22 - a12 * a12;
  const float c1 = (b1 * a22 - b2 * a12) / fmaxf(frac, 1e-2f);
  const float c2 = (-b1 * a12 + a11 * b2) / fmaxf(frac, 1e-2f);
  const float c3 = fminf(fmaxf(b1 / fmaxf(a11, 1e-5f), 0.001f), 1000.0f);
  const bool two = (frac / n_elems) > 1e-2f;
  cd = fminf(fmaxf(two ? c1 : c3, 0.0f), 1000.0f);
  cs = fminf(fmaxf(two ? c2 : 0.0f, 0.0f), 1000.0f);
}
__device__ __forceinline__ float unit_coef(double num, double den) {   // models.py:13-14, 72-77
  return fminf(fmaxf((float)num / fmaxf((float)den, 1e-5f), 0.001f), 1000.0f);
}

template <int POOL>
__device__ __forceinline__ float pool_at(const float* __restrict__ plane, int r, int c, int imW) {
  if (POOL == 1) return plane[(size_t)r * imW + c];
  const float2 t = *reinterpret_cast<const float2*>(plane + (size_t)(2 * r) * imW + 2 * c);
  const float2 u = *reinterpret_cast<const float2*>(plane + (size_t)(2 * r + 1) * imW + 2 * c);
  return (((t.x + t.y) + u.x) + u.y) * 0.25f;
}

// ---- stage A: pool im/seg to the env grid, five masked sums + sum(seg) ------------------------
template <int POOL>
__global__ __launch_bounds__(kLossThreads) void loss_stage_a(const float* __restrict__ diffuse, const float* __restrict__ spec,
                                                              const float* __restrict__ im, const float* __restrict__ seg,
                                                              float* __restrict__ im_s, float* __restrict__ seg_s,
                                                              float* __restrict__ wsA /* [bn,kSplit,6] */, unsigned* __restrict__ ticket,
                                                              int R, int C, int imH, int imW, FoldJob job) {
  __shared__ float lds[4 * 6];
  if (blockIdx.x == kSplitA) {      // the side job's workgroup of this image (launched only when there is one)
    __shared__ double fold_lds[kRThreads * 3];
    recon_fold0_image(job.ws, job.coef, job.den_img, job.nblk, (int)blockIdx.y, fold_lds);
    return;
  }
  const int b = blockIdx.y, RC = R * C, n = 3 * RC;
  if (blockIdx.x == 0 && b == 0 && threadIdx.x == 0) ticket[0] = 0u;      // stage C's arrival counter (two kernel boundaries ahead of its use)
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const size_t plane = (size_t)imH * imW;
  // kStreamUnroll elements per round with every load issued before the first use: the plain grid-stride loop compiles to one memory round
  // trip per element (rocprof: 14 dependent rounds in the third pass = 15 of its 17 us).  Same elements per thread in the same order: same sums.
  constexpr int stride = kSplitA * kLossThreads;
  for (int i0 = blockIdx.x * kLossThreads + threadIdx.x; i0 < n; i0 += kStreamUnroll * stride) {
    float v[kStreamUnroll], dv[kStreamUnroll], sv[kStreamUnroll], sg[kStreamUnroll];
    int pp[kStreamUnroll];
#pragma unroll
    for (int u = 0; u < kStreamUnroll; ++u) {
      const int i = i0 + u * stride < n ? i0 + u * stride : i0;      // lanes past the end re-read their first element (unused)
      const int ch = i / RC, p = i - ch * RC, r = p / C, c = p - r * C;
      pp[u] = ch == 0 ? p : -1;
      v[u] = pool_at<POOL>(im + ((size_t)b * 3 + ch) * plane, r, c, imW);
      dv[u] = diffuse[(size_t)b * n + i];
      sv[u] = spec[(size_t)b * n + i];
      sg[u] = ch == 0 ? pool_at<POOL>(seg + (size_t)b * plane, r, c, imW) : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < kStreamUnroll; ++u) {
      const int i = i0 + u * stride;
      if (i < n) {
        im_s[(size_t)b * n + i] = v[u];
        const float m = v[u] < 0.9f ? 1.0f : 0.0f;
        const float d = dv[u] * m, s = sv[u] * m, vm = v[u] * m;
        acc[0] = fmaf(d, d, acc[0]); acc[1] = fmaf(s, s, acc[1]); acc[2] = fmaf(d, s, acc[2]);
        acc[3] = fmaf(d, vm, acc[3]); acc[4] = fmaf(s, vm, acc[4]);
        if (pp[u] >= 0) {
          seg_s[(size_t)b * RC + pp[u]] = sg[u];
          acc[5] += sg[u];
        }
      }
    }
  }
  block_reduce<6>(acc, lds);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) wsA[((size_t)b * kSplitA + blockIdx.x) * 6 + k] = acc[k];
  }
}

// ---- stage B: sums for the second (one-unknown) regression -------------------------------------
__global__ __launch_bounds__(kLossThreads) void loss_stage_b(const float* __restrict__ diffuse, const float* __restrict__ spec,
                                                              const float* __restrict__ im_s, const float* __restrict__ wsA,
                                                              float* __restrict__ wsB /* [bn,kSplit,2] */, int n) {
  __shared__ float lds[4 * 2];
  const int b = blockIdx.y;
  double sA[6];
  fold_a(wsA, b, sA);
  const double s5[5] = {sA[0], sA[1], sA[2], sA[3], sA[4]};
  float cd, cs;
  diffspec_coefs(s5, (float)n, cd, cs);
  float acc[2] = {0.f, 0.f};
  constexpr int stride = kSplitBC * kLossThreads;
  for (int i0 = blockIdx.x * kLossThreads + threadIdx.x; i0 < n; i0 += kStreamUnrollBC * stride) {      // loads of a round in flight together (see stage A)
    float dv[kStreamUnrollBC], sv[kStreamUnrollBC], iv[kStreamUnrollBC];
#pragma unroll
    for (int u = 0; u < kStreamUnrollBC; ++u) {
      const size_t o = (size_t)b * n + (i0 + u * stride < n ? i0 + u * stride : i0);
      dv[u] = diffuse[o]; sv[u] = spec[o]; iv[u] = im_s[o];
    }
#pragma unroll
    for (int u = 0; u < kStreamUnrollBC; ++u) {
      if (i0 + u * stride < n) {
        const float r = fminf(fmaxf(cd * dv[u] + cs * sv[u], 0.0f), 1.0f);
        acc[0] = fmaf(r, iv[u], acc[0]);
        acc[1] = fmaf(r, r, acc[1]);
      }
    }
  }
  block_reduce<2>(acc, lds);
  if (threadIdx.x == 0) {
    wsB[((size_t)b * kSplitBC + blockIdx.x) * 2 + 0] = acc[0];
    wsB[((size_t)b * kSplitBC + blockIdx.x) * 2 + 1] = acc[1];
  }
}

// ---- loss value and gradient scale from the (rank-summed) totals: wrapperBRDFLight.py:192,205-207 --------
//   loss = num / max(den, 1e-5) / divisor,   scale = d loss / d num      (two separate outputs: the host layer returns the
//   first and saves the second for the backward pass, and they must not share a buffer / version counter)
__device__ __forceinline__ void finalize_pair(const float num, const float den_raw, float divisor, float* loss, float* scale) {
  const float den = fmaxf(den_raw, 1e-5f);
  loss[0] = num / den / divisor;
  scale[0] = 1.0f / den / divisor;
}
// sum of the pooled object mask over the shard (stage A's sixth partial of every workgroup), in double, thread t adding partials
// t, t + 256, ... and a fixed LDS tree: the same value in whichever workgroup evaluates it
__device__ __forceinline__ double shard_mask_total(const float* __restrict__ wsA, int nparts_a, double* lds /* [kLossThreads] */) {
  double den = 0.0;
  for (int i0 = threadIdx.x; i0 < nparts_a; i0 += kStreamUnroll * kLossThreads) {
    float t[kStreamUnroll];
#pragma unroll
    for (int u = 0; u < kStreamUnroll; ++u) t[u] = i0 + u * kLossThreads < nparts_a ? wsA[(size_t)(i0 + u * kLossThreads) * 6 + 5] : 0.0f;
#pragma unroll
    for (int u = 0; u < kStreamUnroll; ++u) den += (double)t[u];
  }
  lds[threadIdx.x] = den;
  __syncthreads();
  for (int s = kLossThreads / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) lds[threadIdx.x] += lds[threadIdx.x + s];
    __syncthreads();
  }
  const double r = lds[0];
  __syncthreads();
  return r;
}
// ---- stage C: final coefficients, rendered image, masked squared error -------------------------
// g_diffuse / g_spec (nullable; one rank, loss != NULL): weight * d loss / d{diffuse, spec} in the same pass -- what loss_bwd would
// produce from the saved images in a fifth launch (the fused light objective asks for it: its gradients are formed ahead of the
// backward call).  The scale 1 / max(den, 1e-5) / divisor needs the shard's mask total, which stage A left in wsA: every workgroup
// folds it (1 024 floats at config 2) the way the last arrival does.
__global__ __launch_bounds__(kLossThreads) void loss_stage_c(const float* __restrict__ diffuse, const float* __restrict__ spec,
                                                              const float* __restrict__ im_s, const float* __restrict__ seg_s,
                                                              const float* __restrict__ wsA, const float* __restrict__ wsB,
                                                              float* __restrict__ coef /* [bn,2] */, float* __restrict__ rendered,
                                                              float* __restrict__ wsC /* [bn,kSplit] */, int RC, unsigned* __restrict__ ticket,
                                                              float* __restrict__ parts, float* __restrict__ loss, float* __restrict__ scale,
                                                              float divisor, float weight, float* __restrict__ g_diffuse,
                                                              float* __restrict__ g_spec) {
  __shared__ float lds[4];
  __shared__ double fold_lds[kLossThreads * 2];
  __shared__ unsigned last;
  const int b = blockIdx.y, n = 3 * RC;
  float gn = 0.0f;
  if (g_diffuse) {
    const float den = fmaxf((float)shard_mask_total(wsA, (int)gridDim.y * kSplitA, fold_lds), 1e-5f);
    gn = weight * (1.0f / den / divisor);      // finalize_pair's scale, loss_bwd's product
  }
  double sA[6], sB[2];
  fold_a(wsA, b, sA);
  fold_bc(wsB, b, sB);
  const double s5[5] = {sA[0], sA[1], sA[2], sA[3], sA[4]};
  float cd, cs;
  diffspec_coefs(s5, (float)n, cd, cs);
  const float cim = unit_coef(sB[0], sB[1]);
  const float kd = cim * cd, ks = cim * cs;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    coef[2 * b] = kd;
    coef[2 * b + 1] = ks;
  }
  float acc[1] = {0.f};
  constexpr int stride = kSplitBC * kLossThreads;
  for (int i0 = blockIdx.x * kLossThreads + threadIdx.x; i0 < n; i0 += kStreamUnrollBC * stride) {      // loads of a round in flight together (see stage A)
    float dv[kStreamUnrollBC], sv[kStreamUnrollBC], iv[kStreamUnrollBC], gv[kStreamUnrollBC];
#pragma unroll
    for (int u = 0; u < kStreamUnrollBC; ++u) {
      const int i = i0 + u * stride < n ? i0 + u * stride : i0;
      const size_t o = (size_t)b * n + i;
      dv[u] = diffuse[o]; sv[u] = spec[o]; iv[u] = im_s[o]; gv[u] = seg_s[(size_t)b * RC + i % RC];
    }
#pragma unroll
    for (int u = 0; u < kStreamUnrollBC; ++u) {
      if (i0 + u * stride < n) {
        const size_t o = (size_t)b * n + i0 + u * stride;
        const float raw = kd * dv[u] + ks * sv[u];
        const float r = fminf(fmaxf(raw, 0.0f), 1.0f);
        rendered[o] = r;
        const float e = r - iv[u], sg = gv[u];
        acc[0] = fmaf(e * e, sg, acc[0]);
        if (g_diffuse) {
          const float g = (raw >= 0.0f && raw <= 1.0f) ? 2.0f * e * sg * gn : 0.0f;
          g_diffuse[o] = g * kd;
          g_spec[o] = g * ks;
        }
      }
    }
  }
  block_reduce<1>(acc, lds);
  // The batch totals [num, den_raw] of this rank's shard, by whichever workgroup arrives last (no fourth launch): every
  // workgroup publishes its partial at agent scope, then draws a ticket; the holder of the last ticket reads all partials
  // back -- in index order, in double, through a fixed LDS tree: the result does not depend on which workgroup that is.
  const int nparts = (int)(gridDim.x * gridDim.y);
  if (threadIdx.x == 0) {
    __hip_atomic_store(&wsC[(size_t)b * kSplitBC + blockIdx.x], acc[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // RELEASE on the ticket: this workgroup's partial is visible at agent scope before its ticket is; the RMWs of the other
    // workgroups continue the release sequence, so the ACQUIRE fence of the last arrival synchronises with every one of them
    // (round 3 relied on the write-through store being counted in vmcnt -- true on gfx950, a data race in the memory model)
    const bool mine = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(nparts - 1);
    if (mine) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    last = mine ? 1u : 0u;
  }
  __syncthreads();      // workgroup-scope release / acquire: the other waves of the last workgroup inherit thread 0's view
  if (!last) return;
  const double den = shard_mask_total(wsA, (int)gridDim.y * kSplitA, fold_lds);      // stage A's partials: a kernel boundary away
  double num = 0.0;
  for (int i = threadIdx.x; i < nparts; i += kLossThreads) num += (double)__hip_atomic_load(&wsC[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  fold_lds[threadIdx.x] = num;
  __syncthreads();
  for (int s = kLossThreads / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) fold_lds[threadIdx.x] += fold_lds[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float fnum = (float)fold_lds[0], fden = (float)den;
    parts[0] = fnum;
    parts[1] = fden;
    if (loss) finalize_pair(fnum, fden, divisor, loss, scale);
  }
}

// ---- loss value and gradient scale after an all-reduce of the totals (sharded batches) ----
__global__ void loss_finalize(const float* __restrict__ parts, float* __restrict__ loss, float* __restrict__ scale, float divisor) {
  finalize_pair(parts[0], parts[1], divisor, loss, scale);
}

// objective = ren_w * renderErr + rec_w * reconstErr with reconstErr = num_e / max(den_e, 1e-5) / divisor_e (trainLight.py:237,
// wrapperBRDFLight.py:179-188): the tail of sgr.light_objective in one launch instead of a dozen one-element torch kernels
__global__ void objective_finalize(const float* __restrict__ render_err, const float* __restrict__ parts_e, float ren_w, float rec_w,
                                   float divisor_e, float* __restrict__ objective, float* __restrict__ recon_err) {
  const float rec = parts_e[0] / fmaxf(parts_e[1], 1e-5f) / divisor_e;
  recon_err[0] = rec;
  objective[0] = ren_w * render_err[0] + rec_w * rec;
}

// (Round 3 measured the four stages in ONE cooperative launch -- per-image device barriers between the passes, the batch fold by
// the last workgroup: 68-85 us against 29 us for the four launches including their gaps.  The workgroups of an image sit on
// different XCDs, so every barrier is a device-scope release + acquire, i.e. an L2 write-back and invalidate per workgroup and
// barrier, which costs more than the launch gaps it removes.  Removed)

// ---- backward: d(num)/d{diffuse, spec} * g_num ------------------------------------------------
// num = sum (clamp(kd D + ks S, 0, 1) - imS)^2 seg ; kd, ks are constants HERE because the images that define them arrive
// detached (wrapperBRDFLight.py:197-201; coefIm and the det indicator are detached by the reference itself, models.py:54,76).
// The reference does not detach coefDiffuse / coefSpecular: call sites that pass live images differentiate through them;
// that mode runs in torch on the host layer (losses.py: _lsregress_diffspec_live), not through this kernel.
__global__ __launch_bounds__(kLossThreads) void loss_bwd(const float* __restrict__ g_num /* device scalar or NULL */, float weight,
                                                          const float* __restrict__ g_scale /* device scalar or NULL */,
                                                          const float* __restrict__ diffuse, const float* __restrict__ spec,
                                                          const float* __restrict__ im_s, const float* __restrict__ seg_s,
                                                          const float* __restrict__ coef, float* __restrict__ g_diffuse,
                                                          float* __restrict__ g_spec, int RC, size_t total) {
  const float gn = (g_num ? g_num[0] : 1.0f) * weight * (g_scale ? g_scale[0] : 1.0f);
  const int n = 3 * RC;
  for (size_t o = (size_t)blockIdx.x * kLossThreads + threadIdx.x; o < total; o += (size_t)gridDim.x * kLossThreads) {
    const int b = (int)(o / n);
    const int p = (int)(o % RC);
    const float kd = coef[2 * b], ks = coef[2 * b + 1];
    const float raw = kd * diffuse[o] + ks * spec[o];
    const float r = fminf(fmaxf(raw, 0.0f), 1.0f);
    float g = (raw >= 0.0f && raw <= 1.0f) ? 2.0f * (r - im_s[o]) * seg_s[(size_t)b * RC + p] * gn : 0.0f;
    g_diffuse[o] = g * kd;
    g_spec[o] = g * ks;
  }
}

// ---- generic per-image <a,b>, <a,a> (LSregress, models.py:7-21) --------------------------------
__global__ __launch_bounds__(kLossThreads) void dot2_partial(const float* __restrict__ a, const float* __restrict__ bb,
                                                              float* __restrict__ ws /* [bn,kSplit,2] */, size_t n) {
  __shared__ float lds[4 * 2];
  const int b = blockIdx.y;
  float acc[2] = {0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * kLossThreads + threadIdx.x; i < n; i += (size_t)kSplit * kLossThreads) {
    const float x = a[(size_t)b * n + i], y = bb[(size_t)b * n + i];
    acc[0] = fmaf(x, y, acc[0]);
    acc[1] = fmaf(x, x, acc[1]);
  }
  block_reduce<2>(acc, lds);
  if (threadIdx.x == 0) {
    ws[((size_t)b * kSplit + blockIdx.x) * 2 + 0] = acc[0];
    ws[((size_t)b * kSplit + blockIdx.x) * 2 + 1] = acc[1];
  }
}
__global__ void unit_coef_finish(const float* __restrict__ ws, float* __restrict__ coef, int bn) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= bn) return;
  double s[2];
  fold<2>(ws, b, s);
  coef[b] = unit_coef(s[0], s[1]);
}

// (c_im c_d, c_im c_s) of LSregressDiffSpec for images already on a common grid
__global__ __launch_bounds__(kLossThreads) void diffspec_partial_a(const float* __restrict__ diffuse, const float* __restrict__ spec,
                                                                    const float* __restrict__ im, float* __restrict__ wsA, int n) {
  __shared__ float lds[4 * 6];
  const int b = blockIdx.y;
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int i = blockIdx.x * kLossThreads + threadIdx.x; i < n; i += kSplitA * kLossThreads) {
    const size_t o = (size_t)b * n + i;
    const float v = im[o];
    const float m = v < 0.9f ? 1.0f : 0.0f;
    const float d = diffuse[o] * m, s = spec[o] * m, vm = v * m;
    acc[0] = fmaf(d, d, acc[0]); acc[1] = fmaf(s, s, acc[1]); acc[2] = fmaf(d, s, acc[2]);
    acc[3] = fmaf(d, vm, acc[3]); acc[4] = fmaf(s, vm, acc[4]);
  }
  block_reduce<6>(acc, lds);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) wsA[((size_t)b * kSplitA + blockIdx.x) * 6 + k] = acc[k];
  }
}
__global__ __launch_bounds__(64) void diffspec_finish(const float* __restrict__ wsA, const float* __restrict__ wsB, float* __restrict__ coef, int n) {
  const int b = blockIdx.x;      // one wave per image (fold_a is lane-cooperative)
  double sA[6], sB[2];
  fold_a(wsA, b, sA);
  fold_bc(wsB, b, sB);
  const double s5[5] = {sA[0], sA[1], sA[2], sA[3], sA[4]};
  float cd, cs;
  diffspec_coefs(s5, (float)n, cd, cs);
  const float cim = unit_coef(sB[0], sB[1]);
  if (threadIdx.x == 0) {
    coef[2 * b] = cim * cd;
    coef[2 * b + 1] = cim * cs;
  }
}

}  // namespace sgr

using namespace sgr;

extern "C" int sgr_loss_workspace_floats(int bn) { return bn * (kSplitA * 6 + (kSplitBC > kSplit ? kSplitBC : kSplit) * (2 + 1)) + 1; }      // + the arrival counter

// the three passes; shared with sgr_light_objective_fwd (sgr_fused_recon.hip), which hands the env-statistics fold to the first one
int sgr::render_loss_fwd_launch(const float* diffuse, const float* spec, const float* im, const float* seg, float* im_small, float* seg_small,
                                float* rendered, float* coef, float* parts, float* loss, float* scale, float divisor, float weight,
                                float* g_diffuse, float* g_spec, float* workspace, int bn, int R, int C, int imH, int imW, FoldJob job,
                                void* stream) {
  SGR_REQUIRE(diffuse && spec && im && seg && im_small && seg_small && rendered && coef && parts && workspace,
              "sgr_render_loss_fwd: NULL tensor");
  SGR_REQUIRE((loss == nullptr) == (scale == nullptr) && (!loss || divisor > 0.0f), "sgr_render_loss_fwd: loss / scale / divisor");
  SGR_REQUIRE(bn > 0 && R > 0 && C > 0, "sgr_render_loss_fwd: non-positive size");
  const bool ok = (imH == R && imW == C) || (imH == 2 * R && imW == 2 * C);
  SGR_SUPPORTED(ok, "sgr_render_loss_fwd: image / env-grid ratio must be 1 or 2 (pool first)");
  const hipStream_t st = (hipStream_t)stream;
  float* wsA = workspace;
  float* wsB = wsA + (size_t)bn * kSplitA * 6;
  float* wsC = wsB + (size_t)bn * kSplitBC * 2;
  unsigned* ticket = reinterpret_cast<unsigned*>(wsC + (size_t)bn * kSplitBC);
  const dim3 grid(kSplitBC, bn), block(kLossThreads);
  const int RC = R * C;
  const dim3 grid_a(kSplitA + (job.ws ? 1 : 0), bn);
  if (imH == R)
    hipLaunchKernelGGL((loss_stage_a<1>), grid_a, block, 0, st, diffuse, spec, im, seg, im_small, seg_small, wsA, ticket, R, C, imH, imW, job);
  else
    hipLaunchKernelGGL((loss_stage_a<2>), grid_a, block, 0, st, diffuse, spec, im, seg, im_small, seg_small, wsA, ticket, R, C, imH, imW, job);
  hipLaunchKernelGGL(loss_stage_b, grid, block, 0, st, diffuse, spec, im_small, wsA, wsB, 3 * RC);
  hipLaunchKernelGGL(loss_stage_c, grid, block, 0, st, diffuse, spec, im_small, seg_small, wsA, wsB, coef, rendered, wsC, RC, ticket, parts,
                     loss, scale, divisor, weight, g_diffuse, g_spec);
  return sgr_check((int)hipGetLastError(), "sgr_render_loss_fwd");
}

extern "C" int sgr_render_loss_fwd_total(const float* diffuse, const float* spec, const float* im, const float* seg, float* im_small,
                                         float* seg_small, float* rendered, float* coef, float* parts, float* loss, float* scale,
                                         float divisor, float* workspace, int bn, int R, int C, int imH, int imW, void* stream) {
  return render_loss_fwd_launch(diffuse, spec, im, seg, im_small, seg_small, rendered, coef, parts, loss, scale, divisor, 0.0f, nullptr, nullptr,
                                workspace, bn, R, C, imH, imW, FoldJob{}, stream);
}

// one rank, loss value AND weight * d loss / d{diffuse, spec} in the three launches (ABI 5; the fused light objective)
extern "C" int sgr_render_loss_fwd_total_grads(const float* diffuse, const float* spec, const float* im, const float* seg, float* im_small,
                                               float* seg_small, float* rendered, float* coef, float* parts, float* loss, float* scale,
                                               float divisor, float weight, float* g_diffuse, float* g_spec, float* workspace, int bn,
                                               int R, int C, int imH, int imW, void* stream) {
  SGR_REQUIRE(loss && scale && g_diffuse && g_spec, "sgr_render_loss_fwd_total_grads: NULL output (the gradient needs the one-rank loss / scale pair)");
  return render_loss_fwd_launch(diffuse, spec, im, seg, im_small, seg_small, rendered, coef, parts, loss, scale, divisor, weight, g_diffuse, g_spec,
                                workspace, bn, R, C, imH, imW, FoldJob{}, stream);
}

extern "C" int sgr_render_loss_fwd(const float* diffuse, const float* spec, const float* im, const float* seg,
                                   float* im_small, float* seg_small, float* rendered, float* coef, float* parts,
                                   float* workspace, int bn, int R, int C, int imH, int imW, void* stream) {
  return sgr_render_loss_fwd_total(diffuse, spec, im, seg, im_small, seg_small, rendered, coef, parts, nullptr, nullptr, 0.0f, workspace,
                                   bn, R, C, imH, imW, stream);
}

extern "C" int sgr_render_loss_bwd_scaled(const float* g_loss, float weight, const float* g_scale, const float* diffuse, const float* spec,
                                          const float* im_small, const float* seg_small, const float* coef, float* g_diffuse,
                                          float* g_spec, int bn, int R, int C, void* stream) {
  SGR_REQUIRE(diffuse && spec && im_small && seg_small && coef && g_diffuse && g_spec, "sgr_render_loss_bwd: NULL tensor");
  SGR_REQUIRE(bn > 0 && R > 0 && C > 0, "sgr_render_loss_bwd: non-positive size");
  const size_t total = (size_t)bn * 3 * R * C;
  const int blocks = (int)((total + kLossThreads * 4 - 1) / (kLossThreads * 4));
  hipLaunchKernelGGL(loss_bwd, dim3(blocks > 2048 ? 2048 : blocks), dim3(kLossThreads), 0, (hipStream_t)stream, g_loss, weight, g_scale,
                     diffuse, spec, im_small, seg_small, coef, g_diffuse, g_spec, R * C, total);
  return sgr_check((int)hipGetLastError(), "sgr_render_loss_bwd");
}

extern "C" int sgr_render_loss_bwd(const float* g_num, const float* diffuse, const float* spec, const float* im_small,
                                   const float* seg_small, const float* coef, float* g_diffuse, float* g_spec, int bn,
                                   int R, int C, void* stream) {
  SGR_REQUIRE(g_num, "sgr_render_loss_bwd: NULL cotangent");
  return sgr_render_loss_bwd_scaled(g_num, 1.0f, nullptr, diffuse, spec, im_small, seg_small, coef, g_diffuse, g_spec, bn, R, C, stream);
}

extern "C" int sgr_loss_finalize(const float* parts, float* loss, float* scale, float divisor, void* stream) {
  SGR_REQUIRE(parts && loss && scale && divisor > 0.0f, "sgr_loss_finalize: bad argument");
  hipLaunchKernelGGL(loss_finalize, dim3(1), dim3(1), 0, (hipStream_t)stream, parts, loss, scale, divisor);
  return sgr_check((int)hipGetLastError(), "sgr_loss_finalize");
}

extern "C" int sgr_objective_finalize(const float* render_err, const float* parts_e, float ren_w, float rec_w, float divisor_e,
                                      float* objective, float* recon_err, void* stream) {
  SGR_REQUIRE(render_err && parts_e && objective && recon_err && divisor_e > 0.0f, "sgr_objective_finalize: bad argument");
  hipLaunchKernelGGL(objective_finalize, dim3(1), dim3(1), 0, (hipStream_t)stream, render_err, parts_e, ren_w, rec_w, divisor_e, objective, recon_err);
  return sgr_check((int)hipGetLastError(), "sgr_objective_finalize");
}

extern "C" int sgr_lsregress_coef(const float* pred, const float* gt, float* coef, float* workspace, int bn, long long n,
                                  void* stream) {
  SGR_REQUIRE(pred && gt && coef && workspace, "sgr_lsregress_coef: NULL tensor");
  SGR_REQUIRE(bn > 0 && n > 0, "sgr_lsregress_coef: non-positive size");
  const hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(dot2_partial, dim3(kSplit, bn), dim3(kLossThreads), 0, st, pred, gt, workspace, (size_t)n);
  hipLaunchKernelGGL(unit_coef_finish, dim3((bn + 63) / 64), dim3(64), 0, st, workspace, coef, bn);
  return sgr_check((int)hipGetLastError(), "sgr_lsregress_coef");
}

extern "C" int sgr_lsregress_diffspec_coef(const float* diffuse, const float* spec, const float* im, float* coef,
                                           float* workspace, int bn, int n, void* stream) {
  SGR_REQUIRE(diffuse && spec && im && coef && workspace, "sgr_lsregress_diffspec_coef: NULL tensor");
  SGR_REQUIRE(bn > 0 && n > 0, "sgr_lsregress_diffspec_coef: non-positive size");
  const hipStream_t st = (hipStream_t)stream;
  float* wsA = workspace;
  float* wsB = wsA + (size_t)bn * kSplitA * 6;
  const dim3 grid(kSplitBC, bn), block(kLossThreads);
  hipLaunchKernelGGL(diffspec_partial_a, dim3(kSplitA, bn), block, 0, st, diffuse, spec, im, wsA, n);
  hipLaunchKernelGGL(loss_stage_b, grid, block, 0, st, diffuse, spec, im, wsA, wsB, n);
  hipLaunchKernelGGL(diffspec_finish, dim3(bn), dim3(64), 0, st, wsA, wsB, coef, n);
  return sgr_check((int)hipGetLastError(), "sgr_lsregress_diffspec_coef");
}
