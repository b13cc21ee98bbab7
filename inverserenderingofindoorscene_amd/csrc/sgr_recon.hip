// Log-L2 reconstruction loss between the predicted and the ground-truth env images
// (wrapperBRDFLight.py:172-188 with models.LSregress, models.py:7-21) on gfx950.
//
// Pure HBM streaming over two [bn,3,R,C,J] tensors (1536 B per shaded pixel each at J=128), so the
// structure is the opposite of the SG kernels: lanes <-> directions.  One wavefront takes one pixel
// at a time; for each colour its 64 lanes read the pixel's J contiguous floats as coalesced float2 /
// float4 rows (512 B per wave-instruction), reduce the per-pixel sums by DPP shuffles, and the
// per-block partials go to a workspace that the next stage folds in double, in a fixed order
// (no atomics, bit-reproducible, no host synchronisation -- the reference's `.item()` on pixelNum,
// wrapperBRDFLight.py:179, stays on the device).
//
//   stage 0: mask_p = segSmall_p * envInd_b * [mean_{c,j} gt > 0.001]                 (:172-174)
//            per image  <pred m, gt m>, <pred m, pred m>  -> coef_b (clamped, models.py:13-14),  sum_p mask_p
//   stage 1: num = sum mask (log(coef pred + off) - log(gt + off))^2                   (:183-187)
//   backward: dnum/dpred = 2 mask (log(coef pred + off) - log(gt + off)) coef / (coef pred + off)
//            (coef is a constant: models.py:13 detaches it)
#include "sgr_recon_fold.h"

namespace sgr {

constexpr int kRWaves = kRThreads / 64;
#ifndef SGR_RECON_NT
#define SGR_RECON_NT 1      // 1: the two env images are streamed with the non-temporal policy (each byte is read once per pass)
#endif
__device__ __forceinline__ float2 ld2(const float* p) {
#if SGR_RECON_NT
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  const f32x2_ v = __builtin_nontemporal_load(reinterpret_cast<const f32x2_*>(p));
  return make_float2(v.x, v.y);
#else
  return *reinterpret_cast<const float2*>(p);
#endif
}
constexpr int kPixPerBlock = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// one pixel's J values of colour c, strided over the wave: element index = lane + 64*i
template <typename F>
__device__ __forceinline__ void for_each_dir(const float* __restrict__ a, const float* __restrict__ b, int J, int lane, F&& f) {
  for (int j = lane; j < J; j += 64) f(a[j], b[j]);
}

__global__ __launch_bounds__(kRThreads) void recon_stage0(const float* __restrict__ env, const float* __restrict__ gt,
                                                           const float* __restrict__ seg_small, const float* __restrict__ env_ind,
                                                           float* __restrict__ mask, float* __restrict__ ws /* [bn,nblk,3] */,
                                                           int RC, int J, int nblk) {
  __shared__ float red[kRWaves][3];
  const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t img = (size_t)b * 3 * RC * J;
  const float ind = env_ind[b];
  float a_eg = 0.f, a_ee = 0.f, a_m = 0.f;
  for (int pi = wave; pi < kPixPerBlock; pi += kRWaves) {
    const int p = blockIdx.x * kPixPerBlock + pi;
    if (p >= RC) break;
    float sg = 0.f, seg_ = 0.f, see = 0.f;
    if ((J & 127) == 0) {          // 8-byte loads: 512 B per wave-instruction, all six rows of the pixel in flight
      for (int j0 = 0; j0 < J; j0 += 128) {
        float2 ev[3], gv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          ev[c] = ld2(env + img + ((size_t)c * RC + p) * J + j0 + 2 * lane);
          gv[c] = ld2(gt + img + ((size_t)c * RC + p) * J + j0 + 2 * lane);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          sg += gv[c].x + gv[c].y;
          seg_ = fmaf(ev[c].x, gv[c].x, fmaf(ev[c].y, gv[c].y, seg_));
          see = fmaf(ev[c].x, ev[c].x, fmaf(ev[c].y, ev[c].y, see));
        }
      }
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float* e = env + img + ((size_t)c * RC + p) * J;
        const float* g = gt + img + ((size_t)c * RC + p) * J;
        for (int j = lane; j < J; j += 64) {
          const float ev = e[j], gv = g[j];
          sg += gv;
          seg_ = fmaf(ev, gv, seg_);
          see = fmaf(ev, ev, see);
        }
      }
    }
    sg = wave_sum(sg); seg_ = wave_sum(seg_); see = wave_sum(see);
    const float not_dark = (sg / (3.0f * (float)J)) > 0.001f ? 1.0f : 0.0f;
    const float m = seg_small[(size_t)b * RC + p] * ind * not_dark;
    if (lane == 0) mask[(size_t)b * RC + p] = m;
    a_eg = fmaf(m * m, seg_, a_eg);
    a_ee = fmaf(m * m, see, a_ee);
    a_m += m;
  }
  if (lane == 0) { red[wave][0] = a_eg; red[wave][1] = a_ee; red[wave][2] = a_m; }
  __syncthreads();
  if (threadIdx.x < 3) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < kRWaves; ++w) s += red[w][threadIdx.x];
    ws[((size_t)b * nblk + blockIdx.x) * 3 + threadIdx.x] = s;
  }
}

__global__ __launch_bounds__(kRThreads) void recon_stage1(const float* __restrict__ env, const float* __restrict__ gt,
                                                           const float* __restrict__ mask, const float* __restrict__ coef,
                                                           float* __restrict__ ws /* [bn,nblk] */, int RC, int J, int nblk, float offset) {
  __shared__ float red[kRWaves];
  const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t img = (size_t)b * 3 * RC * J;
  const float cf = coef[b];
  float acc = 0.f;
  for (int pi = wave; pi < kPixPerBlock; pi += kRWaves) {
    const int p = blockIdx.x * kPixPerBlock + pi;
    if (p >= RC) break;
    const float m = mask[(size_t)b * RC + p];
    float s = 0.f;
    if (m != 0.0f) {      // wave-uniform
      if ((J & 127) == 0) {
        for (int j0 = 0; j0 < J; j0 += 128) {
          float2 ev[3], gv[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            ev[c] = ld2(env + img + ((size_t)c * RC + p) * J + j0 + 2 * lane);
            gv[c] = ld2(gt + img + ((size_t)c * RC + p) * J + j0 + 2 * lane);
          }
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float d0 = __logf(fmaf(cf, ev[c].x, offset)) - __logf(gv[c].x + offset);
            const float d1 = __logf(fmaf(cf, ev[c].y, offset)) - __logf(gv[c].y + offset);
            s = fmaf(d0, d0, fmaf(d1, d1, s));
          }
        }
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float* e = env + img + ((size_t)c * RC + p) * J;
          const float* g = gt + img + ((size_t)c * RC + p) * J;
          for (int j = lane; j < J; j += 64) {
            const float d = __logf(fmaf(cf, e[j], offset)) - __logf(g[j] + offset);
            s = fmaf(d, d, s);
          }
        }
      }
    }
    acc = fmaf(m, s, acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) ws[(size_t)b * nblk + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(kRThreads) void recon_bwd(const float* __restrict__ g_num, const float* __restrict__ env,
                                                        const float* __restrict__ gt, const float* __restrict__ mask,
                                                        const float* __restrict__ coef, float* __restrict__ g_env, int RC, int J,
                                                        float offset) {
  const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t img = (size_t)b * 3 * RC * J;
  const float cf = coef[b];
  const float gn = g_num[0];
  for (int pi = wave; pi < kPixPerBlock; pi += kRWaves) {
    const int p = blockIdx.x * kPixPerBlock + pi;
    if (p >= RC) break;
    const float m2 = 2.0f * mask[(size_t)b * RC + p] * gn * cf;
    if ((J & 127) == 0) {
      for (int j0 = 0; j0 < J; j0 += 128) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const size_t o = img + ((size_t)c * RC + p) * J + j0 + 2 * lane;
          float2 r = make_float2(0.f, 0.f);
          if (m2 != 0.0f) {
            const float2 ev = ld2(env + o);
            const float2 gv = ld2(gt + o);
            const float x0 = fmaf(cf, ev.x, offset), x1 = fmaf(cf, ev.y, offset);
            r.x = m2 * (__logf(x0) - __logf(gv.x + offset)) / x0;
            r.y = m2 * (__logf(x1) - __logf(gv.y + offset)) / x1;
          }
          *reinterpret_cast<float2*>(g_env + o) = r;
        }
      }
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const size_t o = img + ((size_t)c * RC + p) * J;
        for (int j = lane; j < J; j += 64) {
          float r = 0.0f;
          if (m2 != 0.0f) {
            const float x = fmaf(cf, env[o + j], offset);
            r = m2 * (__logf(x) - __logf(gt[o + j] + offset)) / x;
          }
          g_env[o + j] = r;
        }
      }
    }
  }
}

}  // namespace sgr

using namespace sgr;

static int recon_blocks(int RC) { return (RC + kPixPerBlock - 1) / kPixPerBlock; }

extern "C" int sgr_recon_workspace_floats(int bn, int R, int C) { return bn * recon_blocks(R * C) * 4 + 2 * bn; }

extern "C" int sgr_recon_loss_fwd(const float* env, const float* env_gt, const float* seg_small, const float* env_ind,
                                  float* mask, float* coef, float* parts, float* workspace, int bn, int R, int C, int eh,
                                  int ew, float offset, void* stream) {
  SGR_REQUIRE(env && env_gt && seg_small && env_ind && mask && coef && parts && workspace, "sgr_recon_loss_fwd: NULL tensor");
  SGR_REQUIRE(bn > 0 && R > 0 && C > 0 && eh > 0 && ew > 0, "sgr_recon_loss_fwd: non-positive size");
  const int RC = R * C, J = eh * ew, nblk = recon_blocks(RC);
  const hipStream_t st = (hipStream_t)stream;
  float* ws0 = workspace;                          // [bn,nblk,3]
  float* ws1 = ws0 + (size_t)bn * nblk * 3;        // [bn,nblk]
  float* den_img = ws1 + (size_t)bn * nblk;        // [bn]
  const dim3 grid(nblk, bn), block(kRThreads);
  hipLaunchKernelGGL(recon_stage0, grid, block, 0, st, env, env_gt, seg_small, env_ind, mask, ws0, RC, J, nblk);
  hipLaunchKernelGGL(recon_fold0, dim3(bn), dim3(kRThreads), 0, st, ws0, coef, den_img, nblk);
  hipLaunchKernelGGL(recon_stage1, grid, block, 0, st, env, env_gt, mask, coef, ws1, RC, J, nblk, offset);
  hipLaunchKernelGGL(recon_fold1, dim3(1), dim3(kFold1Threads), 0, st, ws1, den_img, parts, bn, nblk, ObjectiveTail{});
  return sgr_check((int)hipGetLastError(), "sgr_recon_loss_fwd");
}

extern "C" int sgr_recon_loss_bwd(const float* g_num, const float* env, const float* env_gt, const float* mask,
                                  const float* coef, float* g_env, int bn, int R, int C, int eh, int ew, float offset,
                                  void* stream) {
  SGR_REQUIRE(g_num && env && env_gt && mask && coef && g_env, "sgr_recon_loss_bwd: NULL tensor");
  SGR_REQUIRE(bn > 0 && R > 0 && C > 0 && eh > 0 && ew > 0, "sgr_recon_loss_bwd: non-positive size");
  const int RC = R * C, J = eh * ew, nblk = recon_blocks(RC);
  hipLaunchKernelGGL(recon_bwd, dim3(nblk, bn), dim3(kRThreads), 0, (hipStream_t)stream, g_num, env, env_gt, mask, coef, g_env, RC,
                     J, offset);
  return sgr_check((int)hipGetLastError(), "sgr_recon_loss_bwd");
}
