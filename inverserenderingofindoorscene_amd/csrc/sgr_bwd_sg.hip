// Backward of output2env.output2env / fromSGtoIm (autograd of models.py:371-404) on gfx950.
#include "sgr_backward.inl"
using namespace sgr;

extern "C" int sgr_sg_to_env_bwd(const float* g_env, const float* axis, const float* lamb, const float* weight,
                                 const float* dirs, float* g_axis, float* g_lamb, float* g_weight, int bn, int K,
                                 int R, int C, int eh, int ew, int premap, void* stream) {
  SGR_REQUIRE(g_env && axis && lamb && weight && dirs && g_axis && g_lamb && g_weight, "sgr_sg_to_env_bwd: NULL tensor");
  SGR_REQUIRE(bn > 0 && K > 0 && R > 0 && C > 0 && eh > 0 && ew > 0, "sgr_sg_to_env_bwd: non-positive size");
  SGR_REQUIRE(premap >= 0 && premap <= 2, "sgr_sg_to_env_bwd: premap must be 0, 1 or 2");
  Args a{};
  a.g_env = g_env; a.axis = axis; a.lamb = lamb; a.weight = weight; a.dirs = reinterpret_cast<const float4*>(dirs);
  a.g_axis = g_axis; a.g_lamb = g_lamb; a.g_weight = g_weight;
  set_dims_b(a, bn, K, R, C, eh, ew, R, C);
  a.premap = premap;
  return sgr_check(sgbwd_launch<true, false>(a, (hipStream_t)stream), "sgr_sg_to_env_bwd");
}
