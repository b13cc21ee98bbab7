// Fused output2env.output2env + renderingLayer.forwardEnv (wrapperBRDFLight.py:177,194) on gfx950.
#include "sgr_forward.inl"
using namespace sgr;

static int fused_fwd_impl(const char* who, const float* albedo, const float* normal, const float* rough, const float* axis,
                          const float* lamb, const float* weight, const float* dirs, const float* view, float* env,
                          float* lamb_tan, float* weight_tan, float* diffuse, float* spec, int bn, int K, int R, int C, int eh,
                          int ew, int imH, int imW, float F0, int premap, void* stream) {
  SGR_REQUIRE(albedo && normal && rough && axis && lamb && weight && dirs && view && diffuse && spec, "sgr_fused_fwd: NULL tensor");
  SGR_REQUIRE(bn > 0 && K > 0 && R > 0 && C > 0 && eh > 0 && ew > 0, "sgr_fused_fwd: non-positive size");
  SGR_SUPPORTED(K <= SGR_MAX_LOBES, "sgr_fused_fwd: SGNum > 32 is not supported");
  if (int rc = check_pool(R, C, imH, imW, "sgr_fused_fwd: BRDF-map / env-grid ratio must be 1 or 2 (pool first)")) return rc;
  Args a{};
  a.albedo = albedo; a.normal = normal; a.rough = rough; a.axis = axis; a.lamb = lamb; a.weight = weight;
  a.dirs = reinterpret_cast<const float4*>(dirs); a.view = view; a.env_out = env; a.diffuse = diffuse; a.spec = spec;
  a.lamb_tan = lamb_tan; a.weight_tan = weight_tan;
  set_dims(a, bn, K, R, C, eh, ew, imH, imW);
  a.F0 = F0; a.premap = premap == 1 ? 1 : (premap == 3 ? 3 : 0);      // 2 (post-tan inputs, a backward-only distinction) is 0 here
  SGR_REQUIRE(premap >= 0 && premap <= 3, "sgr_fused_fwd: premap must be 0..3");
  SGR_SUPPORTED(premap != 3 || fwd_heads_ok(a), "sgr_fused_fwd: premap 3 (decoder heads as a prologue) needs envWidth 16 or 32 and 6 < SGNum <= 24 (sgr_heads_prologue_supported)");
  const hipStream_t st = (hipStream_t)stream;
  return sgr_check(env ? fwd_launch<true, true, true>(a, st) : fwd_launch<true, false, true>(a, st), who);
}

extern "C" int sgr_fused_fwd(const float* albedo, const float* normal, const float* rough, const float* axis,
                             const float* lamb, const float* weight, const float* dirs, const float* view,
                             float* env, float* diffuse, float* spec, int bn, int K, int R, int C, int eh, int ew,
                             int imH, int imW, float F0, int premap, void* stream) {
  return fused_fwd_impl("sgr_fused_fwd", albedo, normal, rough, axis, lamb, weight, dirs, view, env, nullptr, nullptr, diffuse, spec,
                        bn, K, R, C, eh, ew, imH, imW, F0, premap, stream);
}

// the same pass also returning the post-tan sharpness / intensity (what output2env.output2env returns next to the env
// image, models.py:396-404): saved by the host layer and handed to the backward entry points with premap = 2, which then
// skip the 24 tangents per lane
extern "C" int sgr_fused_fwd_tan(const float* albedo, const float* normal, const float* rough, const float* axis,
                                 const float* lamb, const float* weight, const float* dirs, const float* view,
                                 float* env, float* lamb_tan, float* weight_tan, float* diffuse, float* spec, int bn, int K, int R,
                                 int C, int eh, int ew, int imH, int imW, float F0, int premap, void* stream) {
  return fused_fwd_impl("sgr_fused_fwd_tan", albedo, normal, rough, axis, lamb, weight, dirs, view, env, lamb_tan, weight_tan, diffuse,
                        spec, bn, K, R, C, eh, ew, imH, imW, F0, premap, stream);
}

#ifdef SGR_TRACE
// development builds only (tools/wavetrace): where the per-wave trace records of this translation unit's kernels go
extern "C" int sgr_debug_trace_fwd(void* device_buffer) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(sgr::g_trace), &device_buffer, sizeof(void*));
}
#endif

// Whether premap = 3 (axis / lamb / weight are the light decoders' last-convolution outputs; the heads of models.py:336-346 run as
// the prologue of the fused kernels and their chain rule as the epilogue of the backward kernels) is available for a configuration
extern "C" int sgr_heads_prologue_supported(int K, int R, int C, int eh, int ew) {
  Args a{};
  set_dims(a, 1, K, R, C, eh, ew, R, C);
  return (K > 0 && R > 0 && C > 0 && eh > 0 && ew > 0 && fwd_heads_ok(a)) ? 1 : 0;
}
