// Per-wave schedule trace of the packed forward / backward kernels (development tool; needs a -DSGR_TRACE build of the library:
//   make -C inverserenderingofindoorscene_amd/csrc OBJDIR=build_trace OUT=../variants/libsgrender_trace.so EXTRA=-DSGR_TRACE).
//   hipcc -O2 tools/wavetrace.cpp -o tools/wavetrace -ldl ;  tools/wavetrace variants/libsgrender_trace.so [bn] > trace.txt
// Output: one line per wave "kernel wave_id t0 t1 xcc se sh cu simd prologue_ticks shader_cycles" (t in 10 ns ticks of s_memrealtime, relative to the
// first wave; shader_cycles = s_memtime over the same interval: cycles / ticks / 10 ns = the clock the wave ran at).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../include/sgrender.h"
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
struct TraceRec { unsigned long long t0, t1, tp, c0, c1; unsigned hw, xcc; };
static float* dev_rand(size_t n, float lo, float hi, unsigned seed) {
  std::vector<float> h(n); unsigned s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = lo + (hi - lo) * ((s >> 8) * (1.0f / 16777216.0f)); }
  float* d; CHECK(hipMalloc(&d, n * 4)); CHECK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice)); return d;
}
static float* dev_unit3(size_t groups, size_t plane, unsigned seed) {
  std::vector<float> h(groups * 3 * plane); unsigned s = seed * 747796405u + 1u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) * (1.0f / 16777216.0f)) * 2.0f - 1.0f; };
  for (size_t g = 0; g < groups; ++g) for (size_t i = 0; i < plane; ++i) {
    float x = rnd(), y = rnd(), z = fabsf(rnd()) + 0.5f, n = sqrtf(x * x + y * y + z * z);
    h[(g * 3 + 0) * plane + i] = x / n; h[(g * 3 + 1) * plane + i] = y / n; h[(g * 3 + 2) * plane + i] = z / n; }
  float* d; CHECK(hipMalloc(&d, h.size() * 4)); CHECK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice)); return d;
}
static float* dev_empty(size_t n) { float* d; CHECK(hipMalloc(&d, n * 4)); CHECK(hipMemset(d, 0, n * 4)); return d; }
int main(int argc, char** argv) {
  const char* libpath = argc > 1 ? argv[1] : "inverserenderingofindoorscene_amd/variants/libsgrender_trace.so";
  const int bn = argc > 2 ? atoi(argv[2]) : 16;
  void* lib = dlopen(libpath, RTLD_NOW);
  if (!lib) { fprintf(stderr, "dlopen failed: %s\n", dlerror()); return 1; }
#define SYM(name) auto name##_p = (decltype(&name))dlsym(lib, #name); if (!name##_p) { fprintf(stderr, "missing %s\n", #name); return 1; }
  SYM(sgr_fill_direction_table) SYM(sgr_fill_view_vectors) SYM(sgr_dirs_floats) SYM(sgr_fused_fwd) SYM(sgr_fused_bwd_sg)
  typedef int (*set_t)(void*);
  set_t set_fwd = (set_t)dlsym(lib, "sgr_debug_trace_fwd"), set_bwd = (set_t)dlsym(lib, "sgr_debug_trace_bwd");
  if (!set_fwd || !set_bwd) { fprintf(stderr, "library was not built with -DSGR_TRACE\n"); return 1; }
  const int K = 12, imH = 240, imW = 320, R = 120, C = 160, eh = 8, ew = 16, J = eh * ew;
  const size_t RC = (size_t)R * C, P = (size_t)bn * RC;
  const int nd = sgr_dirs_floats_p(eh, ew);
  std::vector<float> hd(nd), hv(3 * RC);
  sgr_fill_direction_table_p(hd.data(), eh, ew); sgr_fill_view_vectors_p(hv.data(), R, C, 57.0f, nullptr);
  float *dirs, *view; CHECK(hipMalloc(&dirs, nd * 4)); CHECK(hipMemcpy(dirs, hd.data(), nd * 4, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&view, 3 * RC * 4)); CHECK(hipMemcpy(view, hv.data(), 3 * RC * 4, hipMemcpyHostToDevice));
  float* albedo = dev_rand((size_t)bn * 3 * imH * imW, 0, 1, 1); float* normal = dev_unit3(bn, (size_t)imH * imW, 2);
  float* rough = dev_rand((size_t)bn * imH * imW, -1, 1, 3); float* axis = dev_unit3((size_t)bn * K, RC, 4);
  float* lamb = dev_rand(P * K, 0, 1, 5); float* weight = dev_rand(P * K * 3, 0, 1, 6);
  float* env = dev_empty(P * 3 * J); float* g_env = dev_rand(P * 3 * J, -1e-3f, 1e-3f, 9);
  float* diffuse = dev_empty(P * 3); float* spec = dev_empty(P * 3);
  float* g_d = dev_rand(P * 3, -1, 1, 10); float* g_s = dev_rand(P * 3, -1, 1, 11);
  float* g_axis = dev_empty(P * K * 3); float* g_lamb = dev_empty(P * K); float* g_weight = dev_empty(P * K * 3);
  const size_t max_waves = (size_t)bn * ((RC + 31) / 32);
  TraceRec* trace; CHECK(hipMalloc(&trace, sizeof(TraceRec) * max_waves));
  std::vector<TraceRec> h(max_waves);
  hipStream_t st; CHECK(hipStreamCreate(&st));
  set_fwd(trace); set_bwd(trace);
  auto run = [&](const char* name, size_t waves, auto fn) {
    for (int i = 0; i < 3; ++i) fn();
    CHECK(hipStreamSynchronize(st));
    CHECK(hipMemset(trace, 0, sizeof(TraceRec) * max_waves));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, st)); fn(); CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipMemcpy(h.data(), trace, sizeof(TraceRec) * waves, hipMemcpyDeviceToHost));
    unsigned long long tmin = ~0ull; for (size_t i = 0; i < waves; ++i) if (h[i].t0 && h[i].t0 < tmin) tmin = h[i].t0;
    printf("# %s waves=%zu event_us=%.1f\n", name, waves, ms * 1e3);
    for (size_t i = 0; i < waves; ++i)
      printf("%s %zu %llu %llu %u %u %u %u %u %llu %llu\n", name, i, h[i].t0 - tmin, h[i].t1 - tmin, h[i].xcc & 0xf, (h[i].hw >> 13) & 7, (h[i].hw >> 12) & 1,
             (h[i].hw >> 8) & 15, (h[i].hw >> 4) & 3, h[i].tp - h[i].t0, h[i].c1 - h[i].c0);
  };
  const size_t w64 = (size_t)bn * ((RC + 63) / 64), w32 = (size_t)bn * ((RC + 31) / 32);
  const size_t wfe = w32;      // the forward that writes the env image: half-wave kernel, one workgroup per 32 pixels (round 3)
  run("fwd_env", wfe, [&] { return sgr_fused_fwd_p(albedo, normal, rough, axis, lamb, weight, dirs, view, env, diffuse, spec, bn, K, R, C, eh, ew, imH, imW, 0.05f, 1, st); });
  run("fwd_noenv", w64, [&] { return sgr_fused_fwd_p(albedo, normal, rough, axis, lamb, weight, dirs, view, (float*)nullptr, diffuse, spec, bn, K, R, C, eh, ew, imH, imW, 0.05f, 1, st); });
  run("bwd_genv", w32, [&] { return sgr_fused_bwd_sg_p(g_env, g_d, g_s, albedo, normal, rough, axis, lamb, weight, dirs, view, g_axis, g_lamb, g_weight, bn, K, R, C, eh, ew, imH, imW, 0.05f, 1, st); });
  // round 4: the fused objective's kernels (a trace build whose sgr_fused_recon.hip carries the trace points exports sgr_debug_trace_recon)
  if (set_t set_rec = (set_t)dlsym(lib, "sgr_debug_trace_recon")) {
    SYM(sgr_fused_fwd_recon) SYM(sgr_fused_bwd_recon) SYM(sgr_fused_recon_workspace_floats)
    set_rec(trace);
    float* env_gt = dev_rand(P * 3 * J, 0, 2, 21); float* seg_small = dev_rand(P, 1, 1, 22); float* env_ind = dev_rand(bn, 1, 1, 23);
    float* mask = dev_empty(P); float* coef = dev_empty(bn); float* parts = dev_empty(8);
    float* wsr = dev_empty((size_t)sgr_fused_recon_workspace_floats_p(bn, R, C));
    for (int pm = 1; pm <= 3; pm += 2) {
      char nm[64];
      snprintf(nm, sizeof nm, "obj_fwd_premap%d", pm);
      run(nm, w32 /* round 5: the half-wave statistics kernel for every 7..24-lobe call */, [&] { return sgr_fused_fwd_recon_p(albedo, normal, rough, axis, lamb, weight, dirs, view, env_gt, seg_small, env_ind, diffuse, spec, mask, coef,
                                                                     parts, wsr, bn, K, R, C, eh, ew, imH, imW, 0.05f, pm, st); });
      snprintf(nm, sizeof nm, "obj_bwd_premap%d", pm);
      run(nm, w32, [&] { return sgr_fused_bwd_recon_p(albedo, normal, rough, axis, lamb, weight, dirs, view, env_gt, mask, coef, (const float*)nullptr, g_d, g_s, g_axis, g_lamb,
                                                      g_weight, parts, wsr, bn, K, R, C, eh, ew, imH, imW, 0.05f, pm, 1.0f, 10.0f, st); });
    }
    set_fwd(trace);
  }
  // inter-kernel gap: forward and backward back to back (separate trace buffers, absolute timestamps)
  TraceRec* trace2; CHECK(hipMalloc(&trace2, sizeof(TraceRec) * max_waves));
  set_fwd(trace); set_bwd(trace2);
  std::vector<TraceRec> h2(max_waves);
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipMemset(trace, 0, sizeof(TraceRec) * max_waves)); CHECK(hipMemset(trace2, 0, sizeof(TraceRec) * max_waves));
    CHECK(hipStreamSynchronize(st));
    hipEvent_t e0, e1, e2; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2));
    CHECK(hipEventRecord(e0, st));
    sgr_fused_fwd_p(albedo, normal, rough, axis, lamb, weight, dirs, view, env, diffuse, spec, bn, K, R, C, eh, ew, imH, imW, 0.05f, 1, st);
    sgr_fused_bwd_sg_p(g_env, g_d, g_s, albedo, normal, rough, axis, lamb, weight, dirs, view, g_axis, g_lamb, g_weight, bn, K, R, C, eh, ew, imH, imW, 0.05f, 1, st);
    CHECK(hipEventRecord(e1, st));
    sgr_fused_fwd_p(albedo, normal, rough, axis, lamb, weight, dirs, view, env, diffuse, spec, bn, K, R, C, eh, ew, imH, imW, 0.05f, 1, st);
    CHECK(hipEventRecord(e2, st));
    CHECK(hipEventSynchronize(e2));
    float ms01, ms12; CHECK(hipEventElapsedTime(&ms01, e0, e1)); CHECK(hipEventElapsedTime(&ms12, e1, e2));
    CHECK(hipMemcpy(h.data(), trace, sizeof(TraceRec) * wfe, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(h2.data(), trace2, sizeof(TraceRec) * w32, hipMemcpyDeviceToHost));
    unsigned long long f0 = ~0ull, f1 = 0, b0 = ~0ull, b1 = 0;
    for (size_t i = 0; i < wfe; ++i) { if (h[i].t0 < f0) f0 = h[i].t0; if (h[i].t1 > f1) f1 = h[i].t1; }
    for (size_t i = 0; i < w32; ++i) { if (h2[i].t0 < b0) b0 = h2[i].t0; if (h2[i].t1 > b1) b1 = h2[i].t1; }
    // the second forward overwrote `trace`; f0/f1 are of the SECOND forward, b0/b1 of the backward before it
    printf("# back-to-back rep %d: events fwd+bwd %.1f us, second fwd %.1f us; bwd wave span %.1f us; gap bwd last wave end -> next fwd first wave start %.2f us; next fwd wave span %.1f us\n",
           rep, ms01 * 1e3, ms12 * 1e3, (b1 - b0) / 100.0, ((long long)f0 - (long long)b1) / 100.0, (f1 - f0) / 100.0);
    if (rep == 2) {
      printf("# fwd_env_after_bwd waves=%zu\n", wfe);
      for (size_t i = 0; i < wfe; ++i)
        printf("fwd_env_after_bwd %zu %llu %llu %u %u %u %u %u %llu %llu\n", i, h[i].t0 - f0, h[i].t1 - f0, h[i].xcc & 0xf, (h[i].hw >> 13) & 7, (h[i].hw >> 12) & 1,
               (h[i].hw >> 8) & 15, (h[i].hw >> 4) & 3, h[i].tp - h[i].t0, h[i].c1 - h[i].c0);
    }
  }
  return 0;
}
