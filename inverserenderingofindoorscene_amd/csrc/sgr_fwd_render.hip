// renderingLayer.forwardEnv (models.py:461-522) on gfx950.
#include "sgr_forward.inl"
using namespace sgr;

extern "C" int sgr_render_env_fwd(const float* albedo, const float* normal, const float* rough, const float* env,
                                  const float* dirs, const float* view, float* diffuse, float* spec, int bn, int R,
                                  int C, int eh, int ew, int imH, int imW, float F0, void* stream) {
  SGR_REQUIRE(albedo && normal && rough && env && dirs && view && diffuse && spec, "sgr_render_env_fwd: NULL tensor");
  SGR_REQUIRE(bn > 0 && R > 0 && C > 0 && eh > 0 && ew > 0, "sgr_render_env_fwd: non-positive size");
  if (int rc = check_pool(R, C, imH, imW, "sgr_render_env_fwd: BRDF-map / env-grid ratio must be 1 or 2 (pool first)")) return rc;
  Args a{};
  a.albedo = albedo; a.normal = normal; a.rough = rough; a.env_in = env;
  a.dirs = reinterpret_cast<const float4*>(dirs); a.view = view; a.diffuse = diffuse; a.spec = spec;
  set_dims(a, bn, 0, R, C, eh, ew, imH, imW);
  a.F0 = F0;
  return sgr_check(fwd_launch<false, false, true>(a, (hipStream_t)stream), "sgr_render_env_fwd");
}
