// output2env.output2env / fromSGtoIm (models.py:371-404) on gfx950.
#include "sgr_forward.inl"
using namespace sgr;

extern "C" int sgr_sg_to_env_fwd(const float* axis, const float* lamb, const float* weight, const float* dirs,
                                 float* env, float* lamb_tan, float* weight_tan, int bn, int K, int R, int C,
                                 int eh, int ew, int premap, void* stream) {
  SGR_REQUIRE(axis && lamb && weight && dirs && env, "sgr_sg_to_env_fwd: NULL tensor");
  SGR_REQUIRE(bn > 0 && K > 0 && R > 0 && C > 0 && eh > 0 && ew > 0, "sgr_sg_to_env_fwd: non-positive size");
  SGR_SUPPORTED(K <= SGR_MAX_LOBES, "sgr_sg_to_env_fwd: SGNum > 32 is not supported");
  SGR_REQUIRE(premap >= 0 && premap <= 2, "sgr_sg_to_env_fwd: premap must be 0, 1 or 2");
  Args a{};
  a.axis = axis; a.lamb = lamb; a.weight = weight; a.dirs = reinterpret_cast<const float4*>(dirs);
  a.env_out = env; a.lamb_tan = lamb_tan; a.weight_tan = weight_tan;
  set_dims(a, bn, K, R, C, eh, ew, R, C);
  a.premap = premap == 1 ? 1 : 0;
  return sgr_check(fwd_launch<true, true, false>(a, (hipStream_t)stream), "sgr_sg_to_env_fwd");
}

// utils.predToShading (utils.py:156-195): cosine-weighted irradiance of the SG mixture per cell.
extern "C" int sgr_sg_shading(const float* axis, const float* lamb, const float* weight, const float* dirs, float* shading,
                              int bn, int K, int R, int C, int eh, int ew, int premap, void* stream) {
  SGR_REQUIRE(axis && lamb && weight && dirs && shading, "sgr_sg_shading: NULL tensor");
  SGR_REQUIRE(bn > 0 && K > 0 && R > 0 && C > 0 && eh > 0 && ew > 0, "sgr_sg_shading: non-positive size");
  SGR_SUPPORTED(K <= 24 && (ew == 16 || ew == 32), "sgr_sg_shading: needs envWidth 16 or 32 and SGNum <= 24");
  SGR_REQUIRE(premap >= 0 && premap <= 2, "sgr_sg_shading: premap must be 0, 1 or 2");
  Args a{};
  a.axis = axis; a.lamb = lamb; a.weight = weight; a.dirs = reinterpret_cast<const float4*>(dirs); a.diffuse = shading;
  set_dims(a, bn, K, R, C, eh, ew, R, C);
  a.premap = premap == 1 ? 1 : 0;
  const dim3 grid = wave_grid(bn, R, C), block(kWave);
  const hipStream_t st = (hipStream_t)stream;
  if (ew == 16) {
    if (K <= 12) hipLaunchKernelGGL((shading_fast_kernel<12, 16>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((shading_fast_kernel<24, 16>), grid, block, 0, st, a);
  } else {
    if (K <= 12) hipLaunchKernelGGL((shading_fast_kernel<12, 32>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((shading_fast_kernel<24, 32>), grid, block, 0, st, a);
  }
  return sgr_check((int)hipGetLastError(), "sgr_sg_shading");
}
