// Fused backward of output2env.output2env + renderingLayer.forwardEnv w.r.t. the SG parameters,
// and dL/dEnv of forwardEnv alone (autograd of models.py:371-404, 511-520) on gfx950.
#include "sgr_backward.inl"
using namespace sgr;

extern "C" int sgr_fused_bwd_sg(const float* g_env, const float* g_diffuse, const float* g_spec,
                                const float* albedo, const float* normal, const float* rough, const float* axis,
                                const float* lamb, const float* weight, const float* dirs, const float* view,
                                float* g_axis, float* g_lamb, float* g_weight, int bn, int K, int R, int C, int eh,
                                int ew, int imH, int imW, float F0, int premap, void* stream) {
  SGR_REQUIRE(g_diffuse && g_spec && albedo && normal && rough && axis && lamb && weight && dirs && view && g_axis &&
                  g_lamb && g_weight, "sgr_fused_bwd_sg: NULL tensor");
  SGR_REQUIRE(bn > 0 && K > 0 && R > 0 && C > 0 && eh > 0 && ew > 0, "sgr_fused_bwd_sg: non-positive size");
  SGR_REQUIRE(premap >= 0 && premap <= 3, "sgr_fused_bwd_sg: premap must be 0..3");
  if (int rc = check_pool_b(R, C, imH, imW, "sgr_fused_bwd_sg: BRDF-map / env-grid ratio must be 1 or 2 (pool first)")) return rc;
  Args a{};
  a.g_env = g_env; a.g_diffuse = g_diffuse; a.g_spec = g_spec;
  a.albedo = albedo; a.normal = normal; a.rough = rough; a.axis = axis; a.lamb = lamb; a.weight = weight;
  a.dirs = reinterpret_cast<const float4*>(dirs); a.view = view;
  a.g_axis = g_axis; a.g_lamb = g_lamb; a.g_weight = g_weight;
  set_dims_b(a, bn, K, R, C, eh, ew, imH, imW);
  a.F0 = F0; a.premap = premap;
  SGR_SUPPORTED(premap != 3 || bwd_heads_ok(a), "sgr_fused_bwd_sg: premap 3 (decoder heads as a prologue) needs envWidth 16 or 32 and 6 < SGNum <= 24 (sgr_heads_prologue_supported)");
  const hipStream_t st = (hipStream_t)stream;
  return sgr_check(g_env ? sgbwd_launch<true, true>(a, st) : sgbwd_launch<false, true>(a, st), "sgr_fused_bwd_sg");
}

extern "C" int sgr_render_env_bwd_env(const float* g_diffuse, const float* g_spec, const float* albedo,
                                      const float* normal, const float* rough, const float* dirs, const float* view,
                                      float* g_env, int bn, int R, int C, int eh, int ew, int imH, int imW, float F0,
                                      void* stream) {
  SGR_REQUIRE(g_diffuse && g_spec && albedo && normal && rough && dirs && view && g_env, "sgr_render_env_bwd_env: NULL tensor");
  SGR_REQUIRE(bn > 0 && R > 0 && C > 0 && eh > 0 && ew > 0, "sgr_render_env_bwd_env: non-positive size");
  if (int rc = check_pool_b(R, C, imH, imW, "sgr_render_env_bwd_env: BRDF-map / env-grid ratio must be 1 or 2 (pool first)")) return rc;
  Args a{};
  a.g_diffuse = g_diffuse; a.g_spec = g_spec; a.albedo = albedo; a.normal = normal; a.rough = rough;
  a.dirs = reinterpret_cast<const float4*>(dirs); a.view = view; a.g_env_out = g_env;
  set_dims_b(a, bn, 0, R, C, eh, ew, imH, imW);
  a.F0 = F0;
  const hipStream_t st = (hipStream_t)stream;
  // the reference's direction grids: packed half-wave kernel, whole 128-byte lines -- 88 us against the table-driven kernel's 122 (warm),
  // 105 against 134 with cold buffers (profiles/r04f_kbench.txt); other grids (and SGR_GENERIC=1): the generic kernel below
  if (fast_ok(a) && !sgr_generic_forced()) {
    const dim3 grid32((unsigned)(bn * ((R * C + kPx - 1) / kPx))), block32(kWave);
    if (ew == 16) {
      if (imH == R) hipLaunchKernelGGL((render_genv_pk_half_kernel<1, 16, 2>), grid32, block32, 0, st, a);
      else hipLaunchKernelGGL((render_genv_pk_half_kernel<2, 16, 2>), grid32, block32, 0, st, a);
    } else {
      if (imH == R) hipLaunchKernelGGL((render_genv_pk_half_kernel<1, 32, 1>), grid32, block32, 0, st, a);
      else hipLaunchKernelGGL((render_genv_pk_half_kernel<2, 32, 1>), grid32, block32, 0, st, a);
    }
    return sgr_check((int)hipGetLastError(), "sgr_render_env_bwd_env");
  }
  const dim3 grid = wave_grid(bn, R, C), block(kWave);
  const bool vec = (a.J % 4 == 0);
  if (imH == R) {
    if (vec) hipLaunchKernelGGL((render_genv_kernel<1, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((render_genv_kernel<1, false>), grid, block, 0, st, a);
  } else {
    if (vec) hipLaunchKernelGGL((render_genv_kernel<2, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((render_genv_kernel<2, false>), grid, block, 0, st, a);
  }
  return sgr_check((int)hipGetLastError(), "sgr_render_env_bwd_env");
}

#ifdef SGR_TRACE
// development builds only (tools/wavetrace): where the per-wave trace records of this translation unit's kernels go
extern "C" int sgr_debug_trace_bwd(void* device_buffer) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(sgr::g_trace), &device_buffer, sizeof(void*));
}
#endif
