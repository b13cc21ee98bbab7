// torch.ops.sgrender.*: the host layer of the render path as a C++ torch extension (round 4).
//
// north_star: "the hot path is hand-written HIP C++ exposed as a torch extension"; SURVEY.md section 8b: "thin C++ TORCH_LIBRARY
// wrappers own allocation and checks".  This file is that layer and nothing more: every operator checks its arguments, allocates
// its outputs with the caching allocator, looks up the constant tables and makes ONE call (the light objective: one short
// sequence of calls) into the C ABI of libsgrender.so (include/sgrender.h) on the current HIP stream.  Nothing here computes.
//
//   * schemas:   TORCH_LIBRARY(sgrender, m)                   -- what torch.compile / FakeTensorMode / opcheck see
//   * kernels:   TORCH_LIBRARY_IMPL(sgrender, CUDA, m)        -- "CUDA" is the device type HIP tensors carry in PyTorch-ROCm
//   * shapes:    TORCH_LIBRARY_IMPL(sgrender, Meta, m)        -- fake-tensor shape functions, no launch
//   * autograd:  TORCH_LIBRARY_IMPL(sgrender, Autograd, m)    -- torch::autograd::Function nodes whose backward is itself made of
//                                                                registered operators (the one autograd implementation of the package;
//                                                                rounds 2-3 had a Python custom_op AND a Python autograd.Function)
//   * CPU:       a boxed fallback that raises -- there is no CPU path and no fallback to one.
//
// The C ABI stays the drop-in boundary: it is resolved with dlopen at first use ($SGR_LIB, else the libsgrender.so next to this
// file's .so), so development builds of the kernel library can be A/B-ed under the same host layer, and a missing library is a
// loud error, not a silent fallback.
//
// Reference call each public operator stands for (file:line relative to the reference checkout):
//   sg_to_env          output2env.output2env / fromSGtoIm            models.py:371-404
//   render_env         renderingLayer.forwardEnv                     models.py:461-522
//   fused_render       both back to back                             wrapperBRDFLight.py:177 + :194
//   render_loss        LSregressDiffSpec + clamp + masked L2         wrapperBRDFLight.py:170-171,192,197-207, models.py:23-84
//   recon_loss_parts   LSregress-scaled log-L2 env reconstruction    wrapperBRDFLight.py:172-188, models.py:7-21
//   light_heads        decoderLight output activations               models.py:336-346, wrapperBRDFLight.py:167-168
//   light_objective    renW * renderErr + recW * reconstErr, fused   wrapperBRDFLight.py:167-207, trainLight.py:237
//   lsregress_*_coef   LSregress / LSregressDiffSpec coefficients    models.py:7-21, 23-84
//   sg_shading, light_albedo_scale, light_encoder_input             utils.py:156-195, testReal.py:421-432, wrapperBRDFLight.py:138-156
#include <dlfcn.h>
#include <link.h>

#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include <ATen/ATen.h>
#include <ATen/core/dispatch/Dispatcher.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/csrc/autograd/custom_function.h>
#include <torch/library.h>

#include <rccl/rccl.h>      // declarations only: the functions are resolved from the RCCL PyTorch itself has loaded (rccl() below)

#include "../../include/sgrender.h"

namespace {

using at::Tensor;
using OptTensor = std::optional<Tensor>;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;
using T2 = std::tuple<Tensor, Tensor>;
using T3 = std::tuple<Tensor, Tensor, Tensor>;
using T4 = std::tuple<Tensor, Tensor, Tensor, Tensor>;
using T5 = std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor>;
using Cam = at::ArrayRef<double>;

// ------------------------------------------------------------------------------------------------------------------
// the C ABI, resolved at first use
// ------------------------------------------------------------------------------------------------------------------
#define SGR_API_LIST(X)                                                                                                          \
  X(sgr_abi_version) X(sgr_last_error) X(sgr_dirs_floats) X(sgr_fill_direction_table) X(sgr_fill_view_vectors)                   \
  X(sgr_sg_to_env_fwd) X(sgr_render_env_fwd) X(sgr_fused_fwd_tan) X(sgr_sg_to_env_bwd) X(sgr_fused_bwd_sg)                        \
  X(sgr_render_env_bwd_env) X(sgr_render_bwd_brdf) X(sgr_loss_workspace_floats) X(sgr_render_loss_fwd)                           \
  X(sgr_render_loss_fwd_total) X(sgr_render_loss_fwd_total_grads) X(sgr_loss_finalize) X(sgr_objective_finalize) X(sgr_render_loss_bwd_scaled)                      \
  X(sgr_lsregress_coef) X(sgr_lsregress_diffspec_coef) X(sgr_sg_shading) X(sgr_recon_workspace_floats) X(sgr_recon_loss_fwd)     \
  X(sgr_recon_loss_bwd) X(sgr_fused_recon_supported) X(sgr_heads_prologue_supported) X(sgr_fused_recon_workspace_floats)         \
  X(sgr_fused_fwd_recon_seg) X(sgr_light_objective_fwd) X(sgr_light_heads_fwd) X(sgr_light_heads_bwd) X(sgr_rescale_inplace_flip) X(sgr_fused_bwd_recon)    \
  X(sgr_fused_bwd_recon_total) X(sgr_glue_workspace_floats) X(sgr_light_albedo_scale) X(sgr_light_input_fwd)

struct Api {
#define SGR_DECL(name) decltype(&::name) name = nullptr;
  SGR_API_LIST(SGR_DECL)
#undef SGR_DECL
  std::string path;
};

void anchor() {}

const Api& api() {
  static const Api a = [] {
    Api r;
    const char* env = std::getenv("SGR_LIB");
    if (env && env[0]) {
      r.path = env;
    } else {
      Dl_info info{};
      TORCH_CHECK(dladdr(reinterpret_cast<void*>(&anchor), &info) && info.dli_fname, "sgrender: cannot locate the torch extension on disk");
      std::string self = info.dli_fname;
      const auto slash = self.find_last_of('/');
      r.path = (slash == std::string::npos ? std::string(".") : self.substr(0, slash)) + "/libsgrender.so";
    }
    void* h = dlopen(r.path.c_str(), RTLD_NOW | RTLD_LOCAL);
    TORCH_CHECK(h, "sgrender: cannot load ", r.path, " (", dlerror(), "): the HIP library has not been built -- run "
                "__graft_entry__.build() or `make -C inverserenderingofindoorscene_amd/csrc`.  This package has no CPU / PyTorch fallback.");
#define SGR_LOAD(name)                                                                     \
  r.name = reinterpret_cast<decltype(r.name)>(dlsym(h, #name));                            \
  TORCH_CHECK(r.name, "sgrender: ", r.path, " does not export " #name "; stale build?");
    SGR_API_LIST(SGR_LOAD)
#undef SGR_LOAD
    TORCH_CHECK(r.sgr_abi_version() == SGR_ABI_VERSION, "sgrender: ", r.path, " has ABI version ", r.sgr_abi_version(), ", this extension needs ",
                SGR_ABI_VERSION);
    return r;
  }();
  return a;
}

void ok(int rc, const char* what) {
  if (rc != 0) {
    const char* msg = api().sgr_last_error();
    TORCH_CHECK(false, "sgrender: ", what, " failed (code ", rc, "): ", msg ? msg : "");
  }
}

// ------------------------------------------------------------------------------------------------------------------
// argument checks, pointers, stream, constant tables
// ------------------------------------------------------------------------------------------------------------------
constexpr const char* kNoCpu =
    "sgrender: this layer runs only on HIP device tensors (MI355X); there is no CPU path. Move the inputs to the GPU (the reference's "
    "isCuda=True mode).";

void require_one(const Tensor& t, c10::Device& dev, bool& have) {
  TORCH_CHECK(t.is_cuda(), kNoCpu);
  TORCH_CHECK(t.scalar_type() == at::kFloat, "sgrender: fp32 tensors required, got ", t.scalar_type());
  if (!have) {
    dev = t.device();
    have = true;
  } else {
    TORCH_CHECK(t.device() == dev, "sgrender: tensors on different devices (", dev, " vs ", t.device(), ")");
  }
}
c10::Device require_hip(std::initializer_list<const Tensor*> ts) {
  c10::Device dev(c10::kCUDA, 0);
  bool have = false;
  for (const Tensor* t : ts)
    if (t && t->defined()) require_one(*t, dev, have);
  TORCH_CHECK(have, "sgrender: no tensor argument");
  return dev;
}
const Tensor* opt(const OptTensor& t) { return t.has_value() && t->defined() ? &*t : nullptr; }

const float* rp(const Tensor& t) { return t.defined() && t.numel() ? t.const_data_ptr<float>() : nullptr; }
float* wp(const Tensor& t) { return t.defined() && t.numel() ? t.data_ptr<float>() : nullptr; }
const float* rp(const Tensor* t) { return t ? rp(*t) : nullptr; }

void* stream_of(const c10::Device& dev) { return c10::hip::getCurrentHIPStream(dev.index()).stream(); }

struct SgDims { int64_t bn, K, R, C; };
SgDims check_sg(const Tensor& axis, const Tensor& lamb, const Tensor& weight) {
  TORCH_CHECK(axis.dim() == 5 && axis.size(2) == 3, "sgrender: axis must be [bn,SGNum,3,envRow,envCol], got ", axis.sizes());
  const int64_t bn = axis.size(0), k = axis.size(1), R = axis.size(3), C = axis.size(4);
  TORCH_CHECK(lamb.sizes() == at::IntArrayRef({bn, k, R, C}), "sgrender: lamb must be [bn,SGNum,envRow,envCol]=[", bn, ",", k, ",", R, ",", C, "], got ",
              lamb.sizes());
  TORCH_CHECK(weight.sizes() == at::IntArrayRef({bn, 3 * k, R, C}), "sgrender: weight must be [bn,3*SGNum,envRow,envCol]=[", bn, ",", 3 * k, ",", R, ",", C,
              "], got ", weight.sizes());
  TORCH_CHECK(k <= SGR_MAX_LOBES, "sgrender: SGNum > 32 is not supported");
  // the reference's broadcasts accept zero-sized tensors and return zero-sized ones; the kernels have no launch for them and no caller
  // of the path produces them (dataLoader.py batches, testReal.py single images): refused here, for device and meta tensors alike
  TORCH_CHECK(bn > 0 && k > 0 && R > 0 && C > 0, "sgrender: zero-sized SG tensors (batch, SGNum or env grid of 0) are not supported, got axis ", axis.sizes());
  return {bn, k, R, C};
}
struct BrdfDims { int64_t bn, h, w; };
BrdfDims check_brdf(const Tensor& albedo, const Tensor& normal, const Tensor& rough) {
  TORCH_CHECK(albedo.dim() == 4 && albedo.size(1) == 3, "sgrender: diffusePred must be [bn,3,h,w], got ", albedo.sizes());
  const int64_t bn = albedo.size(0), h = albedo.size(2), w = albedo.size(3);
  TORCH_CHECK(normal.sizes() == at::IntArrayRef({bn, 3, h, w}), "sgrender: normalPred must be [", bn, ",3,", h, ",", w, "], got ", normal.sizes());
  TORCH_CHECK(rough.sizes() == at::IntArrayRef({bn, 1, h, w}), "sgrender: roughPred must be [", bn, ",1,", h, ",", w, "], got ", rough.sizes());
  TORCH_CHECK(bn > 0 && h > 0 && w > 0, "sgrender: zero-sized BRDF maps (batch or image of 0) are not supported, got diffusePred ", albedo.sizes());
  return {bn, h, w};
}
void check_cam(Cam cam) { TORCH_CHECK(cam.size() == 3, "sgrender: cameraPos must have three components"); }

// Constant tables (models.py:353-363, 415-452), created lazily on whichever device the inputs live on.  The reference keeps them
// as bare attributes on the layer object, created on the current device and never moved by .to(); keying by device keeps that
// behaviour while making one layer object usable from several ranks / devices.  Filled by the C ABI's own host-side builders (bit-equal
// to the numpy tables of tables.py: tests/test_host_emulation.py), uploaded with a synchronous copy on first use -- so a table is
// complete before any stream can read it, and a first use inside a HIP-graph capture fails loudly instead of caching a capture-pool tensor.
struct TableKey {
  int kind, a, b, dev;
  float f[4];
  bool operator<(const TableKey& o) const {
    return std::tie(kind, a, b, dev, f[0], f[1], f[2], f[3]) < std::tie(o.kind, o.a, o.b, o.dev, o.f[0], o.f[1], o.f[2], o.f[3]);
  }
};
std::mutex g_tables_mutex;
std::map<TableKey, Tensor> g_tables;
std::deque<TableKey> g_view_order;     // view-vector tables (kind 1) in insertion order
constexpr size_t kMaxViewTables = 64;  // testReal.py builds a layer per image size: keep the cache bounded

// The bound applies to the view-vector tables only, oldest first (erasing the smallest key evicted the per-device direction
// table -- kind 0 sorts first -- on every view-table miss once the cache was full: two rebuilds and two synchronous uploads per call).
// The direction tables (one per device and direction grid: a handful) are never evicted.
template <typename Fill>
Tensor table(const TableKey& key, const c10::Device& dev, int64_t n, Fill fill) {
  std::lock_guard<std::mutex> lock(g_tables_mutex);
  auto it = g_tables.find(key);
  if (it != g_tables.end()) return it->second;
  // build and upload FIRST: if either throws (out of memory; a first use inside a HIP-graph capture, which fails loudly by design) the
  // bookkeeping below has not run -- round 5 pushed the key into g_view_order before the upload, so a failed upload left a key without a
  // table, a retry pushed a duplicate, and the live count drifted below the bound
  Tensor host = at::empty({n}, at::TensorOptions().dtype(at::kFloat));
  fill(host.data_ptr<float>());
  Tensor t = host.to(dev);
  if (key.kind == 1) {
    if (g_view_order.size() >= kMaxViewTables) {
      g_tables.erase(g_view_order.front());      // drops the CACHE's reference only: a captured HIP graph keeps its tables alive through
      g_view_order.pop_front();                  // cached_tables() (graphs.py: CapturedStep holds what the capture may have read)
    }
    g_view_order.push_back(key);
  }
  g_tables.emplace(key, t);
  return t;
}
// every table currently cached (graphs.py): a CapturedStep keeps these references for as long as its graph lives, so that the eviction
// of a view table from the bounded cache can never free memory a captured launch still reads
std::vector<Tensor> cached_tables() {
  std::lock_guard<std::mutex> lock(g_tables_mutex);
  std::vector<Tensor> out;
  out.reserve(g_tables.size());
  for (const auto& kv : g_tables) out.push_back(kv.second);
  return out;
}
// test hook (tests/test_gpu_ops.py): number of cached tables of a kind (0 direction tables, 1 view-vector tables)
int64_t table_cache_count(int64_t kind) {
  std::lock_guard<std::mutex> lock(g_tables_mutex);
  int64_t n = 0;
  for (const auto& kv : g_tables) n += kv.first.kind == (int)kind;
  return n;
}
Tensor dirs_table(const c10::Device& dev, int64_t eh, int64_t ew) {
  TORCH_CHECK(eh > 0 && ew > 0, "sgrender: envHeight / envWidth must be positive");
  const TableKey key{0, (int)eh, (int)ew, (int)dev.index(), {0.f, 0.f, 0.f, 0.f}};
  return table(key, dev, api().sgr_dirs_floats((int)eh, (int)ew), [&](float* p) { ok(api().sgr_fill_direction_table(p, (int)eh, (int)ew), "sgr_fill_direction_table"); });
}
Tensor view_table(const c10::Device& dev, int64_t R, int64_t C, double fov, Cam cam) {
  check_cam(cam);
  const float c3[3] = {(float)cam[0], (float)cam[1], (float)cam[2]};
  const TableKey key{1, (int)R, (int)C, (int)dev.index(), {(float)fov, c3[0], c3[1], c3[2]}};
  return table(key, dev, 3 * R * C, [&](float* p) { ok(api().sgr_fill_view_vectors(p, (int)R, (int)C, (float)fov, c3), "sgr_fill_view_vectors"); });
}

Tensor none_like(const Tensor& t) { return at::empty({0}, t.options()); }      // operators return tensors only: "not asked for" is an empty tensor
Tensor defined_or_none(const Tensor& t) { return t; }
bool present(const Tensor& t) { return t.defined() && t.numel() > 0; }
std::vector<double> vec(Cam cam) { return std::vector<double>(cam.begin(), cam.end()); }

template <typename Sig>
auto find_op(const char* name) {
  return c10::Dispatcher::singleton().findSchemaOrThrow(name, "").typed<Sig>();
}

// ==================================================================================================================
// SG -> env image
// ==================================================================================================================
T3 sg_to_env_cuda(const Tensor& axis, const Tensor& lamb, const Tensor& weight, int64_t eh, int64_t ew, bool premap, bool want_tan) {
  const auto dev = require_hip({&axis, &lamb, &weight});
  const c10::DeviceGuard guard(dev);
  const Tensor a = axis.contiguous(), l = lamb.contiguous(), w = weight.contiguous();
  const auto d = check_sg(a, l, w);
  Tensor env = at::empty({d.bn, 3, d.R, d.C, eh, ew}, a.options());
  const bool tan = premap && want_tan;
  Tensor lam_t = tan ? at::empty_like(l) : none_like(l), w_t = tan ? at::empty_like(w) : none_like(w);
  const Tensor dirs = dirs_table(dev, eh, ew);
  ok(api().sgr_sg_to_env_fwd(rp(a), rp(l), rp(w), rp(dirs), wp(env), wp(lam_t), wp(w_t), (int)d.bn, (int)d.K, (int)d.R, (int)d.C, (int)eh, (int)ew,
                             premap ? 1 : 0, stream_of(dev)),
     "sgr_sg_to_env_fwd");
  return {env, lam_t, w_t};
}
T3 sg_to_env_meta(const Tensor& axis, const Tensor& lamb, const Tensor& weight, int64_t eh, int64_t ew, bool premap, bool want_tan) {
  const auto d = check_sg(axis, lamb, weight);
  const bool tan = premap && want_tan;
  return {at::empty({d.bn, 3, d.R, d.C, eh, ew}, axis.options()), tan ? at::empty(lamb.sizes(), lamb.options()) : none_like(lamb),
          tan ? at::empty(weight.sizes(), weight.options()) : none_like(weight)};
}
T3 sg_to_env_bwd_cuda(const Tensor& g_env, const Tensor& axis, const Tensor& lamb, const Tensor& weight, int64_t eh, int64_t ew, int64_t premap) {
  const auto dev = require_hip({&g_env, &axis, &lamb, &weight});
  const c10::DeviceGuard guard(dev);
  const Tensor g = g_env.contiguous(), a = axis.contiguous(), l = lamb.contiguous(), w = weight.contiguous();
  const auto d = check_sg(a, l, w);
  TORCH_CHECK(g.sizes() == at::IntArrayRef({d.bn, 3, d.R, d.C, eh, ew}), "sgrender: the env cotangent must be [bn,3,envRow,envCol,envHeight,envWidth]");
  Tensor ga = at::empty_like(a), gl = at::empty_like(l), gw = at::empty_like(w);
  const Tensor dirs = dirs_table(dev, eh, ew);
  ok(api().sgr_sg_to_env_bwd(rp(g), rp(a), rp(l), rp(w), rp(dirs), wp(ga), wp(gl), wp(gw), (int)d.bn, (int)d.K, (int)d.R, (int)d.C, (int)eh, (int)ew,
                             (int)premap, stream_of(dev)),
     "sgr_sg_to_env_bwd");
  return {ga, gl, gw};
}
T3 like3_meta(const Tensor& a, const Tensor& l, const Tensor& w) {
  return {at::empty(a.sizes(), a.options()), at::empty(l.sizes(), l.options()), at::empty(w.sizes(), w.options())};
}
T3 sg_to_env_bwd_meta(const Tensor&, const Tensor& axis, const Tensor& lamb, const Tensor& weight, int64_t, int64_t, int64_t) {
  return like3_meta(axis, lamb, weight);
}

struct SgToEnvFn : public torch::autograd::Function<SgToEnvFn> {
  static variable_list forward(AutogradContext* ctx, const Tensor& axis, const Tensor& lamb, const Tensor& weight, int64_t eh, int64_t ew, bool premap,
                               bool want_tan) {
    T3 out;
    {
      at::AutoDispatchBelowADInplaceOrView guard;
      static auto op = find_op<T3(const Tensor&, const Tensor&, const Tensor&, int64_t, int64_t, bool, bool)>("sgrender::sg_to_env");
      out = op.call(axis, lamb, weight, eh, ew, premap, want_tan);
    }
    const bool tan = premap && want_tan;
    // the post-tan tensors exist anyway when the caller asked for them (the reference returns them): the backward then reads them
    // instead of re-evaluating 4K tangents per cell (premap mode 2)
    if (tan) ctx->save_for_backward({axis, std::get<1>(out), std::get<2>(out)});
    else ctx->save_for_backward({axis, lamb, weight});
    ctx->saved_data["eh"] = eh;
    ctx->saved_data["ew"] = ew;
    ctx->saved_data["premap"] = (int64_t)(tan ? 2 : (premap ? 1 : 0));
    ctx->set_materialize_grads(false);
    if (!tan) ctx->mark_non_differentiable({std::get<1>(out), std::get<2>(out)});
    return {std::get<0>(out), std::get<1>(out), std::get<2>(out)};
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    const auto saved = ctx->get_saved_variables();
    const Tensor &axis = saved[0], &lamb = saved[1], &weight = saved[2];
    const int64_t eh = ctx->saved_data["eh"].toInt(), ew = ctx->saved_data["ew"].toInt(), premap = ctx->saved_data["premap"].toInt();
    Tensor ga, gl, gw;
    if (g[0].defined()) {
      static auto op = find_op<T3(const Tensor&, const Tensor&, const Tensor&, const Tensor&, int64_t, int64_t, int64_t)>("sgrender::sg_to_env_bwd");
      std::tie(ga, gl, gw) = op.call(g[0], axis, lamb, weight, eh, ew, premap);
    }
    // cotangents of the returned post-tan tensors (nobody in the reference differentiates through them, wrapperBRDFLight.py:177;
    // handled for completeness with elementwise torch).  Only produced in mode 2, where lamb / weight ARE the saved post-tan tensors.
    const double scale = 0.999 * (M_PI / 2.0);
    if (premap == 2 && g.size() > 1 && present(g[1])) {
      Tensor extra = g[1] * scale * (1 + lamb * lamb);
      gl = gl.defined() ? gl + extra : extra;
    }
    if (premap == 2 && g.size() > 2 && present(g[2])) {
      Tensor extra = g[2] * scale * (1 + weight * weight);
      gw = gw.defined() ? gw + extra : extra;
    }
    return {ga, gl, gw, Tensor(), Tensor(), Tensor(), Tensor()};
  }
};
T3 sg_to_env_autograd(const Tensor& axis, const Tensor& lamb, const Tensor& weight, int64_t eh, int64_t ew, bool premap, bool want_tan) {
  auto o = SgToEnvFn::apply(axis, lamb, weight, eh, ew, premap, want_tan);
  return {o[0], o[1], o[2]};
}

// ==================================================================================================================
// env image -> (diffuse, specular)
// ==================================================================================================================
T2 render_env_cuda(const Tensor& albedo, const Tensor& normal, const Tensor& rough, const Tensor& env, double fov, double F0, Cam cam) {
  const auto dev = require_hip({&albedo, &normal, &rough, &env});
  const c10::DeviceGuard guard(dev);
  const Tensor a = albedo.contiguous(), n = normal.contiguous(), r = rough.contiguous(), e = env.contiguous();
  const auto b = check_brdf(a, n, r);
  TORCH_CHECK(e.dim() == 6 && e.size(0) == b.bn && e.size(1) == 3, "sgrender: envmap must be [bn,3,envRow,envCol,envHeight,envWidth], got ", e.sizes());
  const int64_t R = e.size(2), C = e.size(3), eh = e.size(4), ew = e.size(5);
  Tensor diffuse = at::empty({b.bn, 3, R, C}, a.options()), spec = at::empty({b.bn, 3, R, C}, a.options());
  const Tensor dirs = dirs_table(dev, eh, ew), view = view_table(dev, R, C, fov, cam);
  ok(api().sgr_render_env_fwd(rp(a), rp(n), rp(r), rp(e), rp(dirs), rp(view), wp(diffuse), wp(spec), (int)b.bn, (int)R, (int)C, (int)eh, (int)ew, (int)b.h,
                              (int)b.w, (float)F0, stream_of(dev)),
     "sgr_render_env_fwd");
  return {diffuse, spec};
}
T2 render_env_meta(const Tensor& albedo, const Tensor& normal, const Tensor& rough, const Tensor& env, double, double, Cam) {
  const auto b = check_brdf(albedo, normal, rough);
  TORCH_CHECK(env.dim() == 6, "sgrender: envmap must be 6-D");
  return {at::empty({b.bn, 3, env.size(2), env.size(3)}, albedo.options()), at::empty({b.bn, 3, env.size(2), env.size(3)}, albedo.options())};
}
Tensor render_env_bwd_env_cuda(const Tensor& g_diffuse, const Tensor& g_spec, const Tensor& albedo, const Tensor& normal, const Tensor& rough, int64_t eh,
                               int64_t ew, double fov, double F0, Cam cam) {
  const auto dev = require_hip({&g_diffuse, &g_spec, &albedo, &normal, &rough});
  const c10::DeviceGuard guard(dev);
  const Tensor gd = g_diffuse.contiguous(), gs = g_spec.contiguous(), a = albedo.contiguous(), n = normal.contiguous(), r = rough.contiguous();
  const auto b = check_brdf(a, n, r);
  TORCH_CHECK(gd.dim() == 4 && gd.sizes() == gs.sizes() && gd.size(0) == b.bn && gd.size(1) == 3, "sgrender: cotangents must be [bn,3,envRow,envCol]");
  const int64_t R = gd.size(2), C = gd.size(3);
  Tensor g_env = at::empty({b.bn, 3, R, C, eh, ew}, a.options());
  const Tensor dirs = dirs_table(dev, eh, ew), view = view_table(dev, R, C, fov, cam);
  ok(api().sgr_render_env_bwd_env(rp(gd), rp(gs), rp(a), rp(n), rp(r), rp(dirs), rp(view), wp(g_env), (int)b.bn, (int)R, (int)C, (int)eh, (int)ew, (int)b.h,
                                  (int)b.w, (float)F0, stream_of(dev)),
     "sgr_render_env_bwd_env");
  return g_env;
}
Tensor render_env_bwd_env_meta(const Tensor& g_diffuse, const Tensor&, const Tensor&, const Tensor&, const Tensor&, int64_t eh, int64_t ew, double, double, Cam) {
  return at::empty({g_diffuse.size(0), 3, g_diffuse.size(2), g_diffuse.size(3), eh, ew}, g_diffuse.options());
}
T3 render_bwd_brdf_cuda(const Tensor& g_diffuse, const Tensor& g_spec, const Tensor& albedo, const Tensor& normal, const Tensor& rough, const OptTensor& env,
                        const OptTensor& axis, const OptTensor& lamb, const OptTensor& weight, int64_t eh, int64_t ew, double fov, double F0, Cam cam,
                        bool premap) {
  const auto dev = require_hip({&g_diffuse, &g_spec, &albedo, &normal, &rough, opt(env), opt(axis), opt(lamb), opt(weight)});
  const c10::DeviceGuard guard(dev);
  const Tensor gd = g_diffuse.contiguous(), gs = g_spec.contiguous(), a = albedo.contiguous(), n = normal.contiguous(), r = rough.contiguous();
  const bool have_env = opt(env) != nullptr;
  TORCH_CHECK(have_env || (opt(axis) && opt(lamb) && opt(weight)), "sgrender: render_bwd_brdf needs the env image or the SG parameters");
  Tensor e, ax, la, we;
  if (have_env) e = env->contiguous();
  else { ax = axis->contiguous(); la = lamb->contiguous(); we = weight->contiguous(); }
  const auto b = check_brdf(a, n, r);
  TORCH_CHECK(gd.dim() == 4 && gd.sizes() == gs.sizes() && gd.size(0) == b.bn, "sgrender: cotangents must be [bn,3,envRow,envCol]");
  const int64_t R = gd.size(2), C = gd.size(3), K = have_env ? 0 : ax.size(1);
  Tensor ga = at::empty_like(a), gn = at::empty_like(n), gr = at::empty_like(r);
  const Tensor dirs = dirs_table(dev, eh, ew), view = view_table(dev, R, C, fov, cam);
  ok(api().sgr_render_bwd_brdf(rp(gd), rp(gs), rp(a), rp(n), rp(r), rp(e), rp(ax), rp(la), rp(we), rp(dirs), rp(view), wp(ga), wp(gn), wp(gr), (int)b.bn,
                               (int)K, (int)R, (int)C, (int)eh, (int)ew, (int)b.h, (int)b.w, (float)F0, premap ? 1 : 0, stream_of(dev)),
     "sgr_render_bwd_brdf");
  return {ga, gn, gr};
}
T3 render_bwd_brdf_meta(const Tensor&, const Tensor&, const Tensor& albedo, const Tensor& normal, const Tensor& rough, const OptTensor&, const OptTensor&,
                        const OptTensor&, const OptTensor&, int64_t, int64_t, double, double, Cam, bool) {
  return like3_meta(albedo, normal, rough);
}

using BrdfBwdSig = T3(const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const OptTensor&, const OptTensor&, const OptTensor&,
                      const OptTensor&, int64_t, int64_t, double, double, Cam, bool);

struct RenderEnvFn : public torch::autograd::Function<RenderEnvFn> {
  static variable_list forward(AutogradContext* ctx, const Tensor& albedo, const Tensor& normal, const Tensor& rough, const Tensor& env, double fov, double F0,
                               std::vector<double> cam) {
    T2 out;
    {
      at::AutoDispatchBelowADInplaceOrView guard;
      static auto op = find_op<T2(const Tensor&, const Tensor&, const Tensor&, const Tensor&, double, double, Cam)>("sgrender::render_env");
      out = op.call(albedo, normal, rough, env, fov, F0, cam);
    }
    ctx->save_for_backward({albedo, normal, rough, env});
    ctx->saved_data["fov"] = fov;
    ctx->saved_data["F0"] = F0;
    ctx->saved_data["cam"] = cam;
    ctx->set_materialize_grads(false);
    return {std::get<0>(out), std::get<1>(out)};
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    const auto saved = ctx->get_saved_variables();
    const Tensor &albedo = saved[0], &normal = saved[1], &rough = saved[2], &env = saved[3];
    const double fov = ctx->saved_data["fov"].toDouble(), F0 = ctx->saved_data["F0"].toDouble();
    const std::vector<double> cam = ctx->saved_data["cam"].toDoubleVector();
    variable_list out(7);
    if (!g[0].defined() && !g[1].defined()) return out;
    const int64_t eh = env.size(4), ew = env.size(5);
    Tensor zeros;
    if (!g[0].defined() || !g[1].defined()) zeros = at::zeros({env.size(0), 3, env.size(2), env.size(3)}, env.options());
    const Tensor gd = g[0].defined() ? g[0] : zeros, gs = g[1].defined() ? g[1] : zeros;
    if (ctx->needs_input_grad(3)) {
      static auto op = find_op<Tensor(const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, int64_t, int64_t, double, double, Cam)>(
          "sgrender::render_env_bwd_env");
      out[3] = op.call(gd, gs, albedo, normal, rough, eh, ew, fov, F0, cam);
    }
    if (ctx->needs_input_grad(0) || ctx->needs_input_grad(1) || ctx->needs_input_grad(2)) {
      static auto op = find_op<BrdfBwdSig>("sgrender::render_bwd_brdf");
      auto [ga, gn, gr] = op.call(gd, gs, albedo, normal, rough, env, std::nullopt, std::nullopt, std::nullopt, eh, ew, fov, F0, cam, false);
      if (ctx->needs_input_grad(0)) out[0] = ga;
      if (ctx->needs_input_grad(1)) out[1] = gn;
      if (ctx->needs_input_grad(2)) out[2] = gr;
    }
    return out;
  }
};
T2 render_env_autograd(const Tensor& albedo, const Tensor& normal, const Tensor& rough, const Tensor& env, double fov, double F0, Cam cam) {
  auto o = RenderEnvFn::apply(albedo, normal, rough, env, fov, F0, vec(cam));
  return {o[0], o[1]};
}

// ==================================================================================================================
// fused: SG -> (env image), diffuse, specular                                     wrapperBRDFLight.py:177 + :194
// ==================================================================================================================
T5 fused_render_cuda(const Tensor& albedo, const Tensor& normal, const Tensor& rough, const Tensor& axis, const Tensor& lamb, const Tensor& weight, int64_t eh,
                     int64_t ew, double fov, double F0, Cam cam, int64_t premap, bool need_env, bool want_tan) {
  const auto dev = require_hip({&albedo, &normal, &rough, &axis, &lamb, &weight});
  const c10::DeviceGuard guard(dev);
  const Tensor a = albedo.contiguous(), n = normal.contiguous(), r = rough.contiguous();
  const Tensor ax = axis.contiguous(), la = lamb.contiguous(), we = weight.contiguous();
  const auto d = check_sg(ax, la, we);
  const auto b = check_brdf(a, n, r);
  TORCH_CHECK(b.bn == d.bn, "sgrender: BRDF maps and SG parameters disagree on the batch size");
  TORCH_CHECK(premap == 0 || premap == 1 || premap == 3, "sgrender: fused_render premap must be 0 (post-tan), 1 (decoder outputs in [0, 1]) or 3 (raw decoder outputs)");
  Tensor env = need_env ? at::empty({d.bn, 3, d.R, d.C, eh, ew}, a.options()) : none_like(a);
  Tensor diffuse = at::empty({d.bn, 3, d.R, d.C}, a.options()), spec = at::empty({d.bn, 3, d.R, d.C}, a.options());
  const bool tan = premap == 1 && want_tan;
  Tensor lam_t = tan ? at::empty_like(la) : none_like(la), w_t = tan ? at::empty_like(we) : none_like(we);
  const Tensor dirs = dirs_table(dev, eh, ew), view = view_table(dev, d.R, d.C, fov, cam);
  ok(api().sgr_fused_fwd_tan(rp(a), rp(n), rp(r), rp(ax), rp(la), rp(we), rp(dirs), rp(view), wp(env), wp(lam_t), wp(w_t), wp(diffuse), wp(spec), (int)d.bn,
                             (int)d.K, (int)d.R, (int)d.C, (int)eh, (int)ew, (int)b.h, (int)b.w, (float)F0, (int)premap, stream_of(dev)),
     "sgr_fused_fwd");
  return {env, diffuse, spec, lam_t, w_t};
}
T5 fused_render_meta(const Tensor& albedo, const Tensor& normal, const Tensor& rough, const Tensor& axis, const Tensor& lamb, const Tensor& weight, int64_t eh,
                     int64_t ew, double, double, Cam, int64_t premap, bool need_env, bool want_tan) {
  const auto d = check_sg(axis, lamb, weight);
  check_brdf(albedo, normal, rough);
  const bool tan = premap == 1 && want_tan;
  return {need_env ? at::empty({d.bn, 3, d.R, d.C, eh, ew}, albedo.options()) : none_like(albedo), at::empty({d.bn, 3, d.R, d.C}, albedo.options()),
          at::empty({d.bn, 3, d.R, d.C}, albedo.options()), tan ? at::empty(lamb.sizes(), lamb.options()) : none_like(lamb),
          tan ? at::empty(weight.sizes(), weight.options()) : none_like(weight)};
}
T3 fused_render_bwd_sg_cuda(const OptTensor& g_env, const Tensor& g_diffuse, const Tensor& g_spec, const Tensor& albedo, const Tensor& normal, const Tensor& rough,
                            const Tensor& axis, const Tensor& lamb, const Tensor& weight, int64_t eh, int64_t ew, double fov, double F0, Cam cam, int64_t premap) {
  const auto dev = require_hip({opt(g_env), &g_diffuse, &g_spec, &albedo, &normal, &rough, &axis, &lamb, &weight});
  const c10::DeviceGuard guard(dev);
  Tensor ge;
  if (opt(g_env)) ge = g_env->contiguous();
  const Tensor gd = g_diffuse.contiguous(), gs = g_spec.contiguous(), a = albedo.contiguous(), n = normal.contiguous(), r = rough.contiguous();
  const Tensor ax = axis.contiguous(), la = lamb.contiguous(), we = weight.contiguous();
  const auto d = check_sg(ax, la, we);
  const auto b = check_brdf(a, n, r);
  TORCH_CHECK(gd.sizes() == at::IntArrayRef({d.bn, 3, d.R, d.C}) && gs.sizes() == gd.sizes(), "sgrender: diffuse / specular cotangents must be [bn,3,envRow,envCol]");
  TORCH_CHECK(!ge.defined() || ge.sizes() == at::IntArrayRef({d.bn, 3, d.R, d.C, eh, ew}), "sgrender: the env cotangent must be [bn,3,envRow,envCol,envHeight,envWidth]");
  Tensor ga = at::empty_like(ax), gl = at::empty_like(la), gw = at::empty_like(we);
  const Tensor dirs = dirs_table(dev, eh, ew), view = view_table(dev, d.R, d.C, fov, cam);
  ok(api().sgr_fused_bwd_sg(rp(ge), rp(gd), rp(gs), rp(a), rp(n), rp(r), rp(ax), rp(la), rp(we), rp(dirs), rp(view), wp(ga), wp(gl), wp(gw), (int)d.bn,
                            (int)d.K, (int)d.R, (int)d.C, (int)eh, (int)ew, (int)b.h, (int)b.w, (float)F0, (int)premap, stream_of(dev)),
     "sgr_fused_bwd_sg");
  return {ga, gl, gw};
}
T3 fused_render_bwd_sg_meta(const OptTensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor& axis, const Tensor& lamb,
                            const Tensor& weight, int64_t, int64_t, double, double, Cam, int64_t) {
  return like3_meta(axis, lamb, weight);
}

struct FusedRenderFn : public torch::autograd::Function<FusedRenderFn> {
  static variable_list forward(AutogradContext* ctx, const Tensor& albedo, const Tensor& normal, const Tensor& rough, const Tensor& axis, const Tensor& lamb,
                               const Tensor& weight, int64_t eh, int64_t ew, double fov, double F0, std::vector<double> cam, int64_t premap, bool need_env,
                               bool want_tan, bool brdf_grads) {
    T5 out;
    {
      at::AutoDispatchBelowADInplaceOrView guard;
      static auto op = find_op<T5(const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, int64_t, int64_t, double, double, Cam,
                                  int64_t, bool, bool)>("sgrender::fused_render");
      out = op.call(albedo, normal, rough, axis, lamb, weight, eh, ew, fov, F0, cam, premap, need_env, want_tan);
    }
    const bool tan = premap == 1 && want_tan;      // the backward reads the post-tan tensors the forward produced (premap mode 2)
    // the env image, when it exists, feeds the BRDF-map gradients (the env-given kernel is faster than re-evaluating the SG)
    ctx->save_for_backward({albedo, normal, rough, axis, tan ? std::get<3>(out) : lamb, tan ? std::get<4>(out) : weight,
                            (need_env && brdf_grads) ? std::get<0>(out) : Tensor()});
    ctx->saved_data["eh"] = eh;
    ctx->saved_data["ew"] = ew;
    ctx->saved_data["fov"] = fov;
    ctx->saved_data["F0"] = F0;
    ctx->saved_data["cam"] = cam;
    ctx->saved_data["premap"] = (int64_t)(tan ? 2 : premap);
    ctx->mark_non_differentiable({std::get<3>(out), std::get<4>(out)});      // internal hand-off to the backward, not part of the layer's interface
    if (!need_env) ctx->mark_non_differentiable({std::get<0>(out)});
    ctx->set_materialize_grads(false);
    return {std::get<0>(out), std::get<1>(out), std::get<2>(out), std::get<3>(out), std::get<4>(out)};
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    const auto saved = ctx->get_saved_variables();
    const Tensor &albedo = saved[0], &normal = saved[1], &rough = saved[2], &axis = saved[3], &lamb = saved[4], &weight = saved[5], &env_saved = saved[6];
    const int64_t eh = ctx->saved_data["eh"].toInt(), ew = ctx->saved_data["ew"].toInt(), premap = ctx->saved_data["premap"].toInt();
    const double fov = ctx->saved_data["fov"].toDouble(), F0 = ctx->saved_data["F0"].toDouble();
    const std::vector<double> cam = ctx->saved_data["cam"].toDoubleVector();
    variable_list out(15);
    Tensor g_env = present(g[0]) ? g[0] : Tensor();
    if (!g_env.defined() && !g[1].defined() && !g[2].defined()) return out;
    const int64_t bn = axis.size(0), R = axis.size(3), C = axis.size(4);
    Tensor zeros;
    if (!g[1].defined() || !g[2].defined()) zeros = at::zeros({bn, 3, R, C}, axis.options());
    const Tensor gd = g[1].defined() ? g[1] : zeros, gs = g[2].defined() ? g[2] : zeros;
    if (ctx->needs_input_grad(3) || ctx->needs_input_grad(4) || ctx->needs_input_grad(5)) {
      static auto op = find_op<T3(const OptTensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&,
                                  const Tensor&, int64_t, int64_t, double, double, Cam, int64_t)>("sgrender::fused_render_bwd_sg");
      auto [ga, gl, gw] = op.call(g_env.defined() ? OptTensor(g_env) : OptTensor(), gd, gs, albedo, normal, rough, axis, lamb, weight, eh, ew, fov, F0, cam, premap);
      out[3] = ga; out[4] = gl; out[5] = gw;
    }
    if (ctx->needs_input_grad(0) || ctx->needs_input_grad(1) || ctx->needs_input_grad(2)) {
      static auto op = find_op<BrdfBwdSig>("sgrender::render_bwd_brdf");
      T3 r = env_saved.defined()
                 ? op.call(gd, gs, albedo, normal, rough, env_saved, std::nullopt, std::nullopt, std::nullopt, eh, ew, fov, F0, cam, premap == 1)
                 : op.call(gd, gs, albedo, normal, rough, std::nullopt, axis, lamb, weight, eh, ew, fov, F0, cam, premap == 1);
      if (ctx->needs_input_grad(0)) out[0] = std::get<0>(r);
      if (ctx->needs_input_grad(1)) out[1] = std::get<1>(r);
      if (ctx->needs_input_grad(2)) out[2] = std::get<2>(r);
    }
    return out;
  }
};
T5 fused_render_autograd(const Tensor& albedo, const Tensor& normal, const Tensor& rough, const Tensor& axis, const Tensor& lamb, const Tensor& weight, int64_t eh,
                         int64_t ew, double fov, double F0, Cam cam, int64_t premap, bool need_env, bool want_tan) {
  const bool brdf_grads = at::GradMode::is_enabled() && (albedo.requires_grad() || normal.requires_grad() || rough.requires_grad());
  TORCH_CHECK(!(brdf_grads && premap == 3), "sgrender: BRDF-map gradients are not available with premap 3 (run light_heads first)");
  auto o = FusedRenderFn::apply(albedo, normal, rough, axis, lamb, weight, eh, ew, fov, F0, vec(cam), premap, need_env, want_tan, brdf_grads);
  return {o[0], o[1], o[2], o[3], o[4]};
}

// ==================================================================================================================
// scale-invariant regressions and the render loss               wrapperBRDFLight.py:170-171,192,197-207, models.py:7-84
// ==================================================================================================================
Tensor loss_workspace(int64_t bn, const Tensor& like) { return at::empty({api().sgr_loss_workspace_floats((int)bn)}, like.options()); }

Tensor lsregress_coef_cuda(const Tensor& pred, const Tensor& gt) {
  const auto dev = require_hip({&pred, &gt});
  const c10::DeviceGuard guard(dev);
  const Tensor p = pred.contiguous(), g = gt.contiguous();
  TORCH_CHECK(p.dim() >= 1 && p.sizes() == g.sizes(), "sgrender: LSregress needs pred and gt of one shape");
  const int64_t nb = p.size(0);
  Tensor coef = at::empty({nb}, p.options()), ws = loss_workspace(nb, p);
  ok(api().sgr_lsregress_coef(rp(p), rp(g), wp(coef), wp(ws), (int)nb, (long long)(p.numel() / nb), stream_of(dev)), "sgr_lsregress_coef");
  return coef;
}
Tensor lsregress_coef_meta(const Tensor& pred, const Tensor&) { return at::empty({pred.size(0)}, pred.options()); }
Tensor lsregress_diffspec_coef_cuda(const Tensor& diff, const Tensor& spec, const Tensor& im) {
  const auto dev = require_hip({&diff, &spec, &im});
  const c10::DeviceGuard guard(dev);
  const Tensor d = diff.contiguous(), s = spec.contiguous(), i = im.contiguous();
  TORCH_CHECK(d.sizes() == s.sizes() && d.sizes() == i.sizes(), "sgrender: LSregressDiffSpec needs diff, spec and imOrig of one shape");
  const int64_t nb = d.size(0);
  Tensor coef = at::empty({nb, 2}, d.options()), ws = loss_workspace(nb, d);
  ok(api().sgr_lsregress_diffspec_coef(rp(d), rp(s), rp(i), wp(coef), wp(ws), (int)nb, (int)(d.numel() / nb), stream_of(dev)), "sgr_lsregress_diffspec_coef");
  return coef;
}
Tensor lsregress_diffspec_coef_meta(const Tensor& diff, const Tensor&, const Tensor&) { return at::empty({diff.size(0), 2}, diff.options()); }

// render_loss(diffuse, spec, im, seg, R, C, total) -> (loss, scale, parts, rendered, im_small, seg_small, coef)
//   total = true  (one rank):  loss = renderErr, scale = d loss / d numerator; three launches (the third pass forms the value)
//   total = false (sharded):   loss / scale empty; parts = (numerator, raw denominator) of this shard, all-reduced by the caller
// differentiable outputs: loss (total) or parts (sharded; only its numerator carries a gradient)
using T7 = std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor>;
T7 render_loss_cuda(const Tensor& diffuse, const Tensor& spec, const Tensor& im, const Tensor& seg, int64_t R, int64_t C, bool total) {
  const auto dev = require_hip({&diffuse, &spec, &im, &seg});
  const c10::DeviceGuard guard(dev);
  const Tensor d = diffuse.contiguous(), s = spec.contiguous(), i = im.contiguous(), g = seg.contiguous();
  TORCH_CHECK(d.dim() == 4, "sgrender: diffuse/spec must be [bn,3,", R, ",", C, "]");
  const int64_t bn = d.size(0);
  TORCH_CHECK(d.sizes() == at::IntArrayRef({bn, 3, R, C}) && s.sizes() == d.sizes(), "sgrender: diffuse/spec must be [bn,3,", R, ",", C, "]");
  TORCH_CHECK(i.dim() == 4 && i.size(0) == bn && i.size(1) == 3 && g.sizes() == at::IntArrayRef({bn, 1, i.size(2), i.size(3)}),
              "sgrender: im must be [bn,3,h,w] and seg [bn,1,h,w]");
  const int64_t imH = i.size(2), imW = i.size(3);
  const auto o = d.options();
  Tensor im_s = at::empty({bn, 3, R, C}, o), seg_s = at::empty({bn, 1, R, C}, o), rendered = at::empty({bn, 3, R, C}, o), coef = at::empty({bn, 2}, o);
  Tensor parts = at::empty({2}, o), ws = loss_workspace(bn, d);
  Tensor loss = total ? at::empty({}, o) : none_like(d), scale = total ? at::empty({1}, o) : none_like(d);      // separate buffers: no shared version counter
  ok(api().sgr_render_loss_fwd_total(rp(d), rp(s), rp(i), rp(g), wp(im_s), wp(seg_s), wp(rendered), wp(coef), wp(parts), total ? loss.data_ptr<float>() : nullptr,
                                     wp(scale), 3.0f, wp(ws), (int)bn, (int)R, (int)C, (int)imH, (int)imW, stream_of(dev)),
     "sgr_render_loss_fwd");
  return {loss, scale, parts, rendered, im_s, seg_s, coef};
}
T7 render_loss_meta(const Tensor& diffuse, const Tensor&, const Tensor&, const Tensor&, int64_t R, int64_t C, bool total) {
  const int64_t bn = diffuse.size(0);
  const auto o = diffuse.options();
  return {total ? at::empty({}, o) : none_like(diffuse), total ? at::empty({1}, o) : none_like(diffuse), at::empty({2}, o), at::empty({bn, 3, R, C}, o),
          at::empty({bn, 3, R, C}, o), at::empty({bn, 1, R, C}, o), at::empty({bn, 2}, o)};
}
// g_loss (device scalar, nullable = 1) * weight * scale (device scalar, nullable = 1) * d numerator / d{diffuse, spec}
T2 render_loss_bwd_cuda(const OptTensor& g_loss, double weight, const OptTensor& scale, const Tensor& diffuse, const Tensor& spec, const Tensor& im_s,
                        const Tensor& seg_s, const Tensor& coef) {
  const auto dev = require_hip({opt(g_loss), opt(scale), &diffuse, &spec, &im_s, &seg_s, &coef});
  const c10::DeviceGuard guard(dev);
  const Tensor d = diffuse.contiguous(), s = spec.contiguous(), i = im_s.contiguous(), g = seg_s.contiguous(), c = coef.contiguous();
  Tensor gl, sc;
  if (opt(g_loss)) gl = g_loss->contiguous();
  if (opt(scale)) sc = scale->contiguous();
  TORCH_CHECK(d.dim() == 4 && s.sizes() == d.sizes() && i.sizes() == d.sizes(), "sgrender: render_loss_bwd shapes disagree");
  Tensor gd = at::empty_like(d), gs = at::empty_like(s);
  ok(api().sgr_render_loss_bwd_scaled(rp(gl), (float)weight, rp(sc), rp(d), rp(s), rp(i), rp(g), rp(c), wp(gd), wp(gs), (int)d.size(0), (int)d.size(2),
                                      (int)d.size(3), stream_of(dev)),
     "sgr_render_loss_bwd");
  return {gd, gs};
}
T2 render_loss_bwd_meta(const OptTensor&, double, const OptTensor&, const Tensor& diffuse, const Tensor& spec, const Tensor&, const Tensor&, const Tensor&) {
  return {at::empty(diffuse.sizes(), diffuse.options()), at::empty(spec.sizes(), spec.options())};
}
struct RenderLossFn : public torch::autograd::Function<RenderLossFn> {
  static variable_list forward(AutogradContext* ctx, const Tensor& diffuse, const Tensor& spec, const Tensor& im, const Tensor& seg, int64_t R, int64_t C,
                               bool total) {
    T7 out;
    {
      at::AutoDispatchBelowADInplaceOrView guard;
      static auto op = find_op<T7(const Tensor&, const Tensor&, const Tensor&, const Tensor&, int64_t, int64_t, bool)>("sgrender::render_loss");
      out = op.call(diffuse, spec, im, seg, R, C, total);
    }
    auto& [loss, scale, parts, rendered, im_s, seg_s, coef] = out;
    ctx->save_for_backward({diffuse, spec, im_s, seg_s, coef, scale});
    ctx->saved_data["total"] = total;
    if (total) ctx->mark_non_differentiable({scale, parts, rendered, im_s, seg_s, coef});
    else ctx->mark_non_differentiable({loss, scale, rendered, im_s, seg_s, coef});
    ctx->set_materialize_grads(false);      // no zero image for the unused cotangent of `rendered` on every backward
    return {loss, scale, parts, rendered, im_s, seg_s, coef};
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    const bool total = ctx->saved_data["total"].toBool();
    variable_list out(7);
    const Tensor& gin = total ? g[0] : g[2];
    if (!gin.defined()) return out;
    const auto saved = ctx->get_saved_variables();
    static auto op = find_op<T2(const OptTensor&, double, const OptTensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&)>(
        "sgrender::render_loss_bwd");
    // total: cotangent of the loss times d loss / d numerator; sharded: the numerator's own cotangent (first element of g_parts)
    const Tensor gl = gin.to(at::kFloat).reshape({-1});
    auto [gd, gs] = op.call(gl, 1.0, total ? OptTensor(saved[5]) : OptTensor(), saved[0], saved[1], saved[2], saved[3], saved[4]);
    out[0] = gd;
    out[1] = gs;
    return out;
  }
};
T7 render_loss_autograd(const Tensor& diffuse, const Tensor& spec, const Tensor& im, const Tensor& seg, int64_t R, int64_t C, bool total) {
  auto o = RenderLossFn::apply(diffuse, spec, im, seg, R, C, total);
  return {o[0], o[1], o[2], o[3], o[4], o[5], o[6]};
}

// render_loss_finalize(diffuse, spec, parts, im_small, seg_small, coef) -> (loss, scale): the step AFTER the all-reduce of `parts` under
// batch sharding: loss = parts[0] / max(parts[1], 1e-5) / 3 with the rank-summed totals, one launch; the autograd node
// sits here -- its backward is the render-loss backward kernel with the GLOBAL normaliser (scale = d loss / d numerator), so every rank
// gets the gradient of the global loss w.r.t. its shard.  (diffuse / spec / im_small / seg_small / coef: what render_loss(total = False)
// took and returned for this shard.)  Five launches and one collective per step, as in rounds 2-3.
T2 render_loss_finalize_cuda(const Tensor& diffuse, const Tensor& spec, const Tensor& parts, const Tensor& im_s, const Tensor& seg_s, const Tensor& coef) {
  const auto dev = require_hip({&diffuse, &spec, &parts, &im_s, &seg_s, &coef});
  const c10::DeviceGuard guard(dev);
  const Tensor p = parts.contiguous();
  TORCH_CHECK(p.numel() == 2, "sgrender: render_loss_finalize takes the two totals [numerator, denominator]");
  Tensor loss = at::empty({}, p.options()), scale = at::empty({1}, p.options());
  ok(api().sgr_loss_finalize(rp(p), loss.data_ptr<float>(), wp(scale), 3.0f, stream_of(dev)), "sgr_loss_finalize");
  return {loss, scale};
}
T2 render_loss_finalize_meta(const Tensor&, const Tensor&, const Tensor& parts, const Tensor&, const Tensor&, const Tensor&) {
  return {at::empty({}, parts.options()), at::empty({1}, parts.options())};
}
struct RenderLossFinalizeFn : public torch::autograd::Function<RenderLossFinalizeFn> {
  static variable_list forward(AutogradContext* ctx, const Tensor& diffuse, const Tensor& spec, const Tensor& parts, const Tensor& im_s, const Tensor& seg_s,
                               const Tensor& coef) {
    T2 out;
    {
      at::AutoDispatchBelowADInplaceOrView guard;
      static auto op = find_op<T2(const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&)>("sgrender::render_loss_finalize");
      out = op.call(diffuse, spec, parts, im_s, seg_s, coef);
    }
    ctx->save_for_backward({diffuse, spec, im_s, seg_s, coef, std::get<1>(out)});
    ctx->mark_non_differentiable({std::get<1>(out)});
    ctx->set_materialize_grads(false);
    return {std::get<0>(out), std::get<1>(out)};
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    variable_list out(6);
    if (!g[0].defined()) return out;
    const auto saved = ctx->get_saved_variables();
    static auto op = find_op<T2(const OptTensor&, double, const OptTensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&)>(
        "sgrender::render_loss_bwd");
    auto [gd, gs] = op.call(g[0].to(at::kFloat).reshape({-1}), 1.0, saved[5], saved[0], saved[1], saved[2], saved[3], saved[4]);
    out[0] = gd;
    out[1] = gs;
    return out;
  }
};
T2 render_loss_finalize_autograd(const Tensor& diffuse, const Tensor& spec, const Tensor& parts, const Tensor& im_s, const Tensor& seg_s, const Tensor& coef) {
  auto o = RenderLossFinalizeFn::apply(diffuse, spec, parts, im_s, seg_s, coef);
  return {o[0], o[1]};
}

// ==================================================================================================================
// env reconstruction loss, unfused (two streaming passes over a materialised env image)      wrapperBRDFLight.py:172-188
// ==================================================================================================================
T3 recon_loss_parts_cuda(const Tensor& env, const Tensor& env_gt, const Tensor& seg_small, const Tensor& env_ind, double offset) {
  const auto dev = require_hip({&env, &env_gt, &seg_small, &env_ind});
  const c10::DeviceGuard guard(dev);
  const Tensor e = env.contiguous(), g = env_gt.contiguous();
  TORCH_CHECK(e.dim() == 6 && e.sizes() == g.sizes() && e.size(1) == 3, "sgrender: envmapsPred / envmaps must both be [bn,3,envRow,envCol,envHeight,envWidth]");
  const int64_t bn = e.size(0), R = e.size(2), C = e.size(3), eh = e.size(4), ew = e.size(5);
  TORCH_CHECK(seg_small.numel() == bn * R * C && env_ind.numel() == bn, "sgrender: the pooled object mask must be [bn,1,envRow,envCol] and envmapsInd [bn,...]");
  const Tensor sm = seg_small.contiguous().reshape({bn, R * C}), ind = env_ind.contiguous().reshape({bn});
  const auto o = e.options();
  Tensor mask = at::empty({bn, R * C}, o), coef = at::empty({bn}, o), parts = at::empty({2}, o);
  Tensor ws = at::empty({api().sgr_recon_workspace_floats((int)bn, (int)R, (int)C)}, o);
  ok(api().sgr_recon_loss_fwd(rp(e), rp(g), rp(sm), rp(ind), wp(mask), wp(coef), wp(parts), wp(ws), (int)bn, (int)R, (int)C, (int)eh, (int)ew, (float)offset,
                              stream_of(dev)),
     "sgr_recon_loss_fwd");
  return {parts, mask, coef};
}
T3 recon_loss_parts_meta(const Tensor& env, const Tensor&, const Tensor&, const Tensor&, double) {
  const auto o = env.options();
  return {at::empty({2}, o), at::empty({env.size(0), env.size(2) * env.size(3)}, o), at::empty({env.size(0)}, o)};
}
Tensor recon_loss_bwd_cuda(const Tensor& g_num, const Tensor& env, const Tensor& env_gt, const Tensor& mask, const Tensor& coef, double offset) {
  const auto dev = require_hip({&g_num, &env, &env_gt, &mask, &coef});
  const c10::DeviceGuard guard(dev);
  const Tensor e = env.contiguous(), g = env_gt.contiguous(), m = mask.contiguous(), c = coef.contiguous(), gn = g_num.contiguous();
  TORCH_CHECK(e.dim() == 6 && e.sizes() == g.sizes(), "sgrender: recon_loss_bwd shapes disagree");
  Tensor g_env = at::empty_like(e);
  ok(api().sgr_recon_loss_bwd(rp(gn), rp(e), rp(g), rp(m), rp(c), wp(g_env), (int)e.size(0), (int)e.size(2), (int)e.size(3), (int)e.size(4), (int)e.size(5),
                              (float)offset, stream_of(dev)),
     "sgr_recon_loss_bwd");
  return g_env;
}
Tensor recon_loss_bwd_meta(const Tensor&, const Tensor& env, const Tensor&, const Tensor&, const Tensor&, double) { return at::empty(env.sizes(), env.options()); }
struct ReconLossFn : public torch::autograd::Function<ReconLossFn> {
  static variable_list forward(AutogradContext* ctx, const Tensor& env, const Tensor& env_gt, const Tensor& seg_small, const Tensor& env_ind, double offset) {
    T3 out;
    {
      at::AutoDispatchBelowADInplaceOrView guard;
      static auto op = find_op<T3(const Tensor&, const Tensor&, const Tensor&, const Tensor&, double)>("sgrender::recon_loss_parts");
      out = op.call(env, env_gt, seg_small, env_ind, offset);
    }
    ctx->save_for_backward({env, env_gt, std::get<1>(out), std::get<2>(out)});
    ctx->saved_data["offset"] = offset;
    ctx->mark_non_differentiable({std::get<1>(out), std::get<2>(out)});
    ctx->set_materialize_grads(false);
    return {std::get<0>(out), std::get<1>(out), std::get<2>(out)};
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    variable_list out(5);
    if (!g[0].defined()) return out;
    const auto saved = ctx->get_saved_variables();
    static auto op = find_op<Tensor(const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, double)>("sgrender::recon_loss_bwd");
    out[0] = op.call(g[0].to(at::kFloat).reshape({-1}), saved[0], saved[1], saved[2], saved[3], ctx->saved_data["offset"].toDouble());      // reads the numerator's cotangent
    return out;
  }
};
T3 recon_loss_parts_autograd(const Tensor& env, const Tensor& env_gt, const Tensor& seg_small, const Tensor& env_ind, double offset) {
  auto o = ReconLossFn::apply(env, env_gt, seg_small, env_ind, offset);
  return {o[0], o[1], o[2]};
}

// ==================================================================================================================
// decoder output heads                                                      models.py:336-346, wrapperBRDFLight.py:167-168
// ==================================================================================================================
struct HeadDims { int64_t bn, K, R, C; };
HeadDims check_heads(const Tensor& xa, const Tensor& xl, const Tensor& xw) {
  TORCH_CHECK(xa.dim() == 4 && xl.dim() == 4 && xw.dim() == 4 && xa.size(1) % 3 == 0,
              "sgrender: light_heads takes the three decoders' [bn,3K,R,C], [bn,K,R,C], [bn,3K,R,C] outputs");
  const int64_t bn = xa.size(0), K = xa.size(1) / 3, R = xa.size(2), C = xa.size(3);
  TORCH_CHECK(xl.sizes() == at::IntArrayRef({bn, K, R, C}) && xw.sizes() == at::IntArrayRef({bn, 3 * K, R, C}), "sgrender: light_heads shapes disagree: ", xa.sizes(),
              ", ", xl.sizes(), ", ", xw.sizes());
  return {bn, K, R, C};
}
T4 light_heads_cuda(const Tensor& xa, const Tensor& xl, const Tensor& xw, bool need_packed) {
  const auto dev = require_hip({&xa, &xl, &xw});
  const c10::DeviceGuard guard(dev);
  const Tensor a = xa.contiguous(), l = xl.contiguous(), w = xw.contiguous();
  const auto d = check_heads(a, l, w);
  const auto o = a.options();
  Tensor axis = at::empty({d.bn, d.K, 3, d.R, d.C}, o), lamb = at::empty({d.bn, d.K, d.R, d.C}, o), weight = at::empty({d.bn, 3 * d.K, d.R, d.C}, o);
  Tensor packed = need_packed ? at::empty({d.bn, 7 * d.K, d.R, d.C}, o) : none_like(a);
  ok(api().sgr_light_heads_fwd(rp(a), rp(l), rp(w), wp(axis), wp(lamb), wp(weight), wp(packed), (int)d.bn, (int)d.K, (int)d.R, (int)d.C, stream_of(dev)),
     "sgr_light_heads_fwd");
  return {axis, lamb, weight, packed};
}
T4 light_heads_meta(const Tensor& xa, const Tensor& xl, const Tensor& xw, bool need_packed) {
  const auto d = check_heads(xa, xl, xw);
  const auto o = xa.options();
  return {at::empty({d.bn, d.K, 3, d.R, d.C}, o), at::empty({d.bn, d.K, d.R, d.C}, o), at::empty({d.bn, 3 * d.K, d.R, d.C}, o),
          need_packed ? at::empty({d.bn, 7 * d.K, d.R, d.C}, o) : none_like(xa)};
}
T3 light_heads_bwd_cuda(const Tensor& xa, const Tensor& xl, const Tensor& xw, const OptTensor& g_axis, const OptTensor& g_lamb, const OptTensor& g_weight,
                        const OptTensor& g_packed) {
  const auto dev = require_hip({&xa, &xl, &xw, opt(g_axis), opt(g_lamb), opt(g_weight), opt(g_packed)});
  const c10::DeviceGuard guard(dev);
  const Tensor a = xa.contiguous(), l = xl.contiguous(), w = xw.contiguous();
  const auto d = check_heads(a, l, w);
  Tensor ga, gl, gw, gp;
  if (opt(g_axis)) ga = g_axis->contiguous();
  if (opt(g_lamb)) gl = g_lamb->contiguous();
  if (opt(g_weight)) gw = g_weight->contiguous();
  if (opt(g_packed)) gp = g_packed->contiguous();
  Tensor gxa = at::empty_like(a), gxl = at::empty_like(l), gxw = at::empty_like(w);
  ok(api().sgr_light_heads_bwd(rp(a), rp(l), rp(w), rp(ga), rp(gl), rp(gw), rp(gp), wp(gxa), wp(gxl), wp(gxw), (int)d.bn, (int)d.K, (int)d.R, (int)d.C,
                               stream_of(dev)),
     "sgr_light_heads_bwd");
  return {gxa, gxl, gxw};
}
T3 light_heads_bwd_meta(const Tensor& xa, const Tensor& xl, const Tensor& xw, const OptTensor&, const OptTensor&, const OptTensor&, const OptTensor&) {
  return like3_meta(xa, xl, xw);
}
struct LightHeadsFn : public torch::autograd::Function<LightHeadsFn> {
  static variable_list forward(AutogradContext* ctx, const Tensor& xa, const Tensor& xl, const Tensor& xw, bool need_packed) {
    T4 out;
    {
      at::AutoDispatchBelowADInplaceOrView guard;
      static auto op = find_op<T4(const Tensor&, const Tensor&, const Tensor&, bool)>("sgrender::light_heads");
      out = op.call(xa, xl, xw, need_packed);
    }
    ctx->save_for_backward({xa, xl, xw});
    if (!need_packed) ctx->mark_non_differentiable({std::get<3>(out)});
    ctx->set_materialize_grads(false);
    return {std::get<0>(out), std::get<1>(out), std::get<2>(out), std::get<3>(out)};
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    variable_list out(4);
    if (!g[0].defined() && !g[1].defined() && !g[2].defined() && !(g.size() > 3 && present(g[3]))) return out;
    const auto saved = ctx->get_saved_variables();
    static auto op = find_op<T3(const Tensor&, const Tensor&, const Tensor&, const OptTensor&, const OptTensor&, const OptTensor&, const OptTensor&)>(
        "sgrender::light_heads_bwd");
    auto o = [](const Tensor& t) { return present(t) ? OptTensor(t) : OptTensor(); };
    auto [a, l, w] = op.call(saved[0], saved[1], saved[2], o(g[0]), o(g[1]), o(g[2]), g.size() > 3 ? o(g[3]) : OptTensor());
    out[0] = a; out[1] = l; out[2] = w;
    return out;
  }
};
T4 light_heads_autograd(const Tensor& xa, const Tensor& xl, const Tensor& xw, bool need_packed) {
  auto o = LightHeadsFn::apply(xa, xl, xw, need_packed);
  return {o[0], o[1], o[2], o[3]};
}

// ==================================================================================================================
// glue either side of the path (forward only)                utils.py:156-195, testReal.py:421-432, wrapperBRDFLight.py:138-156
// ==================================================================================================================
Tensor sg_shading_cuda(const Tensor& axis, const Tensor& lamb, const Tensor& weight, int64_t eh, int64_t ew, int64_t premap) {
  const auto dev = require_hip({&axis, &lamb, &weight});
  const c10::DeviceGuard guard(dev);
  const Tensor a = axis.contiguous(), l = lamb.contiguous(), w = weight.contiguous();
  const auto d = check_sg(a, l, w);
  Tensor out = at::empty({d.bn, 3, d.R, d.C}, a.options());
  const Tensor dirs = dirs_table(dev, eh, ew);
  ok(api().sgr_sg_shading(rp(a), rp(l), rp(w), rp(dirs), wp(out), (int)d.bn, (int)d.K, (int)d.R, (int)d.C, (int)eh, (int)ew, (int)premap, stream_of(dev)), "sgr_sg_shading");
  return out;
}
Tensor sg_shading_meta(const Tensor& axis, const Tensor& lamb, const Tensor& weight, int64_t, int64_t, int64_t) {
  const auto d = check_sg(axis, lamb, weight);
  return at::empty({d.bn, 3, d.R, d.C}, axis.options());
}
Tensor light_albedo_scale_cuda(const Tensor& dn, const Tensor& d, const Tensor& sn, const Tensor& s, const Tensor& alb) {
  const auto dev = require_hip({&dn, &d, &sn, &s, &alb});
  const c10::DeviceGuard guard(dev);
  const Tensor a = dn.contiguous(), b = d.contiguous(), c = sn.contiguous(), e = s.contiguous(), al = alb.contiguous();
  TORCH_CHECK(a.sizes() == b.sizes() && a.sizes() == c.sizes() && a.sizes() == e.sizes(), "sgrender: light_albedo_scale needs four render images of one shape");
  Tensor out = at::empty({4}, a.options()), ws = at::empty({api().sgr_glue_workspace_floats(1)}, a.options());
  ok(api().sgr_light_albedo_scale(rp(a), rp(b), rp(c), rp(e), rp(al), wp(out), wp(ws), (long long)b.numel(), (long long)al.numel(), stream_of(dev)),
     "sgr_light_albedo_scale");
  return out;
}
Tensor light_albedo_scale_meta(const Tensor& dn, const Tensor&, const Tensor&, const Tensor&, const Tensor&) { return at::empty({4}, dn.options()); }
T3 light_encoder_input_cuda(const Tensor& im, const Tensor& albedo, const Tensor& normal, const Tensor& rough, const Tensor& depth, int64_t H, int64_t W) {
  const auto dev = require_hip({&im, &albedo, &normal, &rough, &depth});
  const c10::DeviceGuard guard(dev);
  const Tensor i = im.contiguous(), a = albedo.contiguous(), n = normal.contiguous(), r = rough.contiguous(), dp = depth.contiguous();
  TORCH_CHECK(i.dim() == 4 && i.size(1) == 3, "sgrender: light_encoder_input takes im/albedo/normal [bn,3,h,w] and rough/depth [bn,1,h,w]");
  const int64_t bn = i.size(0), h = i.size(2), w = i.size(3);
  TORCH_CHECK(a.sizes() == i.sizes() && n.sizes() == i.sizes() && r.sizes() == at::IntArrayRef({bn, 1, h, w}) && dp.sizes() == r.sizes(),
              "sgrender: light_encoder_input takes im/albedo/normal [bn,3,h,w] and rough/depth [bn,1,h,w]");
  const auto o = i.options();
  Tensor out = at::empty({bn, 11, H, W}, o), alb_n = at::empty_like(a), dep_n = at::empty_like(dp), ws = at::empty({api().sgr_glue_workspace_floats((int)bn)}, o);
  ok(api().sgr_light_input_fwd(rp(i), rp(a), rp(n), rp(r), rp(dp), wp(out), wp(alb_n), wp(dep_n), wp(ws), (int)bn, (int)h, (int)w, (int)H, (int)W, stream_of(dev)),
     "sgr_light_input_fwd");
  return {out, alb_n, dep_n};
}
T3 light_encoder_input_meta(const Tensor& im, const Tensor& albedo, const Tensor&, const Tensor&, const Tensor& depth, int64_t H, int64_t W) {
  return {at::empty({im.size(0), 11, H, W}, im.options()), at::empty(albedo.sizes(), albedo.options()), at::empty(depth.sizes(), depth.options())};
}

// ==================================================================================================================
// the whole trainLight objective, env image never materialised              wrapperBRDFLight.py:167-207, trainLight.py:237
// ==================================================================================================================
// Gradients that exist before backward() is called (the objective's heavy backward pass also produces the loss value, so it
// runs in forward): a node that hands them out times the incoming cotangent.  The scaling happens in place on the device
// (sgr_rescale_inplace_flip: a no-op kernel when the cotangent equals what the gradients are scaled by already -- 1 for a
// plain objective.backward()); `applied` holds two slots, the factor currently applied in applied[parity].
void rescale_grads_cuda(Tensor& g_axis, Tensor& g_lamb, Tensor& g_weight, const Tensor& scale, Tensor& applied, int64_t parity) {
  const auto dev = require_hip({&g_axis, &g_lamb, &g_weight, &scale, &applied});
  const c10::DeviceGuard guard(dev);
  TORCH_CHECK(g_axis.is_contiguous() && g_lamb.is_contiguous() && g_weight.is_contiguous() && applied.numel() == 2 && scale.numel() == 1, "sgrender: rescale_grads_ arguments");
  float* xs[3] = {g_axis.data_ptr<float>(), g_lamb.data_ptr<float>(), g_weight.data_ptr<float>()};
  const long long ns[3] = {(long long)g_axis.numel(), (long long)g_lamb.numel(), (long long)g_weight.numel()};
  const Tensor sc = scale.contiguous();
  ok(api().sgr_rescale_inplace_flip(xs, ns, 3, rp(sc), applied.data_ptr<float>(), (int)parity, stream_of(dev)), "sgr_rescale_inplace_flip");
}
void rescale_grads_meta(Tensor&, Tensor&, Tensor&, const Tensor&, Tensor&, int64_t) {}

struct PrecomputedGradsFn : public torch::autograd::Function<PrecomputedGradsFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& value, const Tensor& axis, const Tensor& lamb, const Tensor& weight, const Tensor& g_axis,
                        const Tensor& g_lamb, const Tensor& g_weight, const Tensor& applied) {
    ctx->save_for_backward({g_axis, g_lamb, g_weight, applied});
    ctx->saved_data["parity"] = (int64_t)0;
    ctx->saved_data["handed_out"] = false;
    ctx->set_materialize_grads(false);
    at::AutoDispatchBelowADInplaceOrView guard;
    return value.alias();
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    variable_list out(8);
    if (!g[0].defined()) return out;
    const auto saved = ctx->get_saved_variables();
    Tensor g_axis = saved[0], g_lamb = saved[1], g_weight = saved[2], applied = saved[3];
    const int64_t parity = ctx->saved_data["parity"].toInt();
    if (ctx->saved_data["handed_out"].toBool()) {
      // a second backward through this node (retain_graph): the buffers may be somebody's .grad by now -- leave them alone.  If the
      // first backward came with a zero cotangent the stored gradients were scaled to zero in place and cannot be recovered: say so
      // instead of returning inf / NaN (rare path, so the host sync is acceptable)
      TORCH_CHECK(applied[parity].item<float>() != 0.0f, "sgrender: light_objective was first back-propagated with a zero cotangent; its stored "
                  "gradients are gone -- re-evaluate the objective instead of reusing the graph");
      const Tensor f = g[0].detach() / applied[parity];
      if (ctx->needs_input_grad(1)) out[1] = g_axis * f;
      if (ctx->needs_input_grad(2)) out[2] = g_lamb * f;
      if (ctx->needs_input_grad(3)) out[3] = g_weight * f;
      return out;
    }
    ctx->saved_data["handed_out"] = true;
    static auto op = find_op<void(Tensor&, Tensor&, Tensor&, const Tensor&, Tensor&, int64_t)>("sgrender::rescale_grads_");
    op.call(g_axis, g_lamb, g_weight, g[0].detach().to(at::kFloat).reshape({1}), applied, parity);
    ctx->saved_data["parity"] = (int64_t)(1 - parity);
    if (ctx->needs_input_grad(1)) out[1] = g_axis;
    if (ctx->needs_input_grad(2)) out[2] = g_lamb;
    if (ctx->needs_input_grad(3)) out[3] = g_weight;
    return out;
  }
};
// Python-callable face of the node (the sharded objective assembles its value from stage operators and collectives)
Tensor attach_grads_backend(const Tensor& value, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&) {
  return value.clone();
}
Tensor attach_grads_autograd(const Tensor& value, const Tensor& axis, const Tensor& lamb, const Tensor& weight, const Tensor& g_axis, const Tensor& g_lamb,
                             const Tensor& g_weight, const Tensor& applied) {
  return PrecomputedGradsFn::apply(value, axis, lamb, weight, g_axis, g_lamb, g_weight, applied);
}

struct ObjDims { int64_t bn, K, R, C, h, w, imH, imW; };
ObjDims check_objective(const Tensor& albedo, const Tensor& normal, const Tensor& rough, const Tensor& axis, const Tensor& lamb, const Tensor& weight,
                        const Tensor& im, const Tensor& seg, const Tensor& env_gt, const Tensor& env_ind, int64_t eh, int64_t ew) {
  const auto d = check_sg(axis, lamb, weight);
  const auto b = check_brdf(albedo, normal, rough);
  TORCH_CHECK(b.bn == d.bn && env_gt.sizes() == at::IntArrayRef({d.bn, 3, d.R, d.C, eh, ew}), "sgrender: envmapsBatch must be [bn,3,", d.R, ",", d.C, ",", eh, ",", ew,
              "] and the batch sizes must agree");
  TORCH_CHECK(im.dim() == 4 && im.size(0) == d.bn && im.size(1) == 3 && seg.sizes() == at::IntArrayRef({d.bn, 1, im.size(2), im.size(3)}),
              "sgrender: im must be [bn,3,h,w] and seg [bn,1,h,w]");
  TORCH_CHECK(env_ind.numel() == d.bn, "sgrender: envmapsInd must hold one value per image");
  return {d.bn, d.K, d.R, d.C, b.h, b.w, im.size(2), im.size(3)};
}

// light_objective_fwdbwd(...) -> (objective, render_err, recon_err, rendered, coef, g_axis, g_lamb, g_weight, applied)
// One rank.  need_grad: forward statistics pass -> render loss (3 launches, value included) -> render-loss backward -> the
// objective's backward pass (SG gradients + reconstruction numerator + the scalar tail in its fold): ten launches, eight of them
// small.  !need_grad (round 4, forward-only callers): the last pass runs without its gradient half (sgr_fused_bwd_recon_total
// with NULL gradient outputs) and the render-loss backward is skipped; the gradient outputs are empty.
using T9 = std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor>;
T9 light_objective_fwdbwd_cuda(const Tensor& albedo, const Tensor& normal, const Tensor& rough, const Tensor& axis, const Tensor& lamb, const Tensor& weight,
                               const Tensor& im, const Tensor& seg, const Tensor& env_gt, const Tensor& env_ind, int64_t eh, int64_t ew, double fov, double F0, Cam cam,
                               double ren_w, double rec_w, double offset, bool heads, bool handoff, bool need_grad) {
  const auto dev = require_hip({&albedo, &normal, &rough, &axis, &lamb, &weight, &im, &seg, &env_gt, &env_ind});
  const c10::DeviceGuard guard(dev);
  const Tensor a = albedo.contiguous(), n = normal.contiguous(), r = rough.contiguous(), ax = axis.contiguous(), la = lamb.contiguous(), we = weight.contiguous();
  const Tensor i = im.contiguous(), sg = seg.contiguous(), gt = env_gt.contiguous(), ind = env_ind.contiguous().reshape({-1});
  const auto d = check_objective(a, n, r, ax, la, we, i, sg, gt, ind, eh, ew);
  const auto o = a.options();
  const Api& A = api();
  void* st = stream_of(dev);
  const int bn = (int)d.bn, K = (int)d.K, R = (int)d.R, C = (int)d.C;
  Tensor diffuse = at::empty({d.bn, 3, d.R, d.C}, o), spec = at::empty({d.bn, 3, d.R, d.C}, o), im_s = at::empty({d.bn, 3, d.R, d.C}, o);
  Tensor seg_s = at::empty({d.bn, 1, d.R, d.C}, o), rendered = at::empty({d.bn, 3, d.R, d.C}, o), mask = at::empty({d.bn, d.R * d.C}, o), coef = at::empty({d.bn}, o);
  Tensor coef_ds = at::empty({d.bn, 2}, o), parts_r = at::empty({2}, o), parts_b = at::empty({2}, o), scale_r = at::empty({1}, o);
  Tensor ws = at::empty({A.sgr_fused_recon_workspace_floats(bn, R, C)}, o), ws_r = loss_workspace(d.bn, a);
  Tensor objective = at::empty({}, o), render_err = at::empty({}, o), recon_err = at::empty({}, o);
  handoff = handoff && !heads && need_grad;
  const int pm = heads ? 3 : 1;      // 3: axis / lamb / weight are the decoders' last-convolution outputs (heads as the kernels' prologue)
  Tensor lam_t = handoff ? at::empty_like(la) : Tensor(), w_t = handoff ? at::empty_like(we) : Tensor();      // post-tan values: forward writes, backward reads (premap 2)
  const Tensor dirs = dirs_table(dev, eh, ew), view = view_table(dev, d.R, d.C, fov, cam);
  // forward half in four launches (ABI 5): the statistics kernel (it pools the object mask itself: the env mask needs it before the render-loss
  // pass produces it), then the render loss's three passes -- the first also folds the env statistics per image, the third also writes
  // ren_w * d renderErr / d{diffuse, spec} when gradients are wanted (no fold launch, no loss_bwd launch between the two heavy kernels)
  Tensor g_axis = none_like(a), g_lamb = none_like(a), g_weight = none_like(a), applied = none_like(a), g_d, g_s;
  if (need_grad) {
    g_axis = at::empty_like(ax); g_lamb = at::empty_like(la); g_weight = at::empty_like(we); applied = at::empty({2}, o);
    g_d = at::empty_like(diffuse); g_s = at::empty_like(spec);
  }
  ok(A.sgr_light_objective_fwd(rp(a), rp(n), rp(r), rp(ax), rp(la), rp(we), rp(dirs), rp(view), rp(gt), rp(i), rp(sg), rp(ind), wp(lam_t), wp(w_t), wp(diffuse), wp(spec),
                               wp(mask), wp(coef), wp(im_s), wp(seg_s), wp(rendered), wp(coef_ds), wp(parts_r), render_err.data_ptr<float>(), wp(scale_r), (float)ren_w,
                               wp(g_d), wp(g_s), wp(ws), wp(ws_r), bn, K, R, C, (int)eh, (int)ew, (int)d.imH, (int)d.imW, (int)d.h, (int)d.w, (float)F0, pm, st),
     "sgr_light_objective_fwd");
  ok(A.sgr_fused_bwd_recon_total(rp(a), rp(n), rp(r), rp(ax), handoff ? rp(lam_t) : rp(la), handoff ? rp(w_t) : rp(we), rp(dirs), rp(view), rp(gt), rp(mask), rp(coef),
                                 rp(g_d), rp(g_s), wp(g_axis), wp(g_lamb), wp(g_weight), wp(parts_b), wp(ws), bn, K, R, C, (int)eh, (int)ew, (int)d.h, (int)d.w,
                                 (float)F0, handoff ? 2 : pm, (float)offset, (float)rec_w, rp(render_err), (float)ren_w, objective.data_ptr<float>(),
                                 recon_err.data_ptr<float>(), wp(applied), st),
     "sgr_fused_bwd_recon");
  return {objective, render_err, recon_err, rendered, coef, g_axis, g_lamb, g_weight, applied};
}
T9 light_objective_fwdbwd_meta(const Tensor& albedo, const Tensor& normal, const Tensor& rough, const Tensor& axis, const Tensor& lamb, const Tensor& weight,
                               const Tensor& im, const Tensor& seg, const Tensor& env_gt, const Tensor& env_ind, int64_t eh, int64_t ew, double, double, Cam, double,
                               double, double, bool, bool, bool need_grad) {
  const auto d = check_objective(albedo, normal, rough, axis, lamb, weight, im, seg, env_gt, env_ind, eh, ew);
  const auto o = albedo.options();
  auto g = [&](const Tensor& t) { return need_grad ? at::empty(t.sizes(), t.options()) : none_like(albedo); };
  return {at::empty({}, o), at::empty({}, o), at::empty({}, o), at::empty({d.bn, 3, d.R, d.C}, o), at::empty({d.bn}, o), g(axis), g(lamb), g(weight),
          need_grad ? at::empty({2}, o) : none_like(albedo)};
}
using ObjSig = T9(const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&,
                  int64_t, int64_t, double, double, Cam, double, double, double, bool, bool, bool);

// light_objective(...) -> (objective, renderErr, reconstErr, renderedImPred, envScale); differentiable w.r.t. the SG parameters only
T5 light_objective_backend(const Tensor& albedo, const Tensor& normal, const Tensor& rough, const Tensor& axis, const Tensor& lamb, const Tensor& weight, const Tensor& im,
                           const Tensor& seg, const Tensor& env_gt, const Tensor& env_ind, int64_t eh, int64_t ew, double fov, double F0, Cam cam, double ren_w,
                           double rec_w, double offset, bool heads, bool handoff) {
  static auto op = find_op<ObjSig>("sgrender::light_objective_fwdbwd");
  auto o = op.call(albedo, normal, rough, axis, lamb, weight, im, seg, env_gt, env_ind, eh, ew, fov, F0, cam, ren_w, rec_w, offset, heads, handoff, false);
  return {std::get<0>(o), std::get<1>(o), std::get<2>(o), std::get<3>(o), std::get<4>(o)};
}
T5 light_objective_autograd(const Tensor& albedo, const Tensor& normal, const Tensor& rough, const Tensor& axis, const Tensor& lamb, const Tensor& weight, const Tensor& im,
                            const Tensor& seg, const Tensor& env_gt, const Tensor& env_ind, int64_t eh, int64_t ew, double fov, double F0, Cam cam, double ren_w,
                            double rec_w, double offset, bool heads, bool handoff) {
  const bool grad_mode = at::GradMode::is_enabled();
  TORCH_CHECK(!(grad_mode && (albedo.requires_grad() || normal.requires_grad() || rough.requires_grad() || im.requires_grad() || seg.requires_grad() ||
                              env_gt.requires_grad() || env_ind.requires_grad())),
              "sgrender: light_objective differentiates w.r.t. the SG parameters only (trainLight mode, wrapperBRDFLight.py:194 detaches the BRDF maps)");
  const bool need = grad_mode && (axis.requires_grad() || lamb.requires_grad() || weight.requires_grad());
  T9 o;
  {
    at::AutoDispatchBelowADInplaceOrView guard;
    static auto op = find_op<ObjSig>("sgrender::light_objective_fwdbwd");
    o = op.call(albedo, normal, rough, axis, lamb, weight, im, seg, env_gt, env_ind, eh, ew, fov, F0, cam, ren_w, rec_w, offset, heads, handoff, need);
  }
  Tensor objective = std::get<0>(o);
  if (need) objective = PrecomputedGradsFn::apply(objective, axis, lamb, weight, std::get<5>(o), std::get<6>(o), std::get<7>(o), std::get<8>(o));
  return {objective, std::get<1>(o), std::get<2>(o), std::get<3>(o), std::get<4>(o)};
}

// floats of the fused objective's workspace: sgr_fused_recon_workspace_floats (csrc/sgr_fused_recon.hip) restated as host arithmetic, so that
// the Meta kernels report the real size without loading the kernel library (tests/test_ops_registration.py holds the two together)
int64_t recon_workspace_floats(int64_t bn, int64_t R, int64_t C) {
  const int64_t rc = R * C, tiles32 = (rc + 31) / 32, tiles16 = (rc + 15) / 16;
  return bn + 4 + bn * tiles32 * 3 + bn * tiles16;
}
// stage 2 takes stage 1's tensors as they are: sizes implied by (bn, R, C), checked for device and meta tensors alike (a
// tensor of another shard or shape made the kernels read and write out of bounds instead of raising)
void check_stage1_tensors(int64_t bn, int64_t R, int64_t C, const Tensor& mask, const Tensor& coef, const Tensor& diffuse, const Tensor& spec, const Tensor& im_s,
                          const Tensor& seg_s, const Tensor& coef_ds, const Tensor& sums, const Tensor& ws) {
  const int64_t px = bn * R * C;
  TORCH_CHECK(mask.numel() == px, "sgrender: light_objective_stage2: mask must have bn*R*C = ", px, " elements, got ", mask.sizes());
  TORCH_CHECK(coef.numel() == bn, "sgrender: light_objective_stage2: coef must have bn = ", bn, " elements, got ", coef.sizes());
  TORCH_CHECK(diffuse.numel() == 3 * px && spec.numel() == 3 * px && im_s.numel() == 3 * px,
              "sgrender: light_objective_stage2: diffuse / spec / im_s must be [bn,3,R,C] = ", 3 * px, " elements, got ", diffuse.sizes(), " ", spec.sizes(), " ", im_s.sizes());
  TORCH_CHECK(seg_s.numel() == px, "sgrender: light_objective_stage2: seg_s must be [bn,1,R,C], got ", seg_s.sizes());
  TORCH_CHECK(coef_ds.numel() == 2 * bn, "sgrender: light_objective_stage2: coef_ds must be [bn,2], got ", coef_ds.sizes());
  TORCH_CHECK(sums.numel() == 4, "sgrender: light_objective_stage2: sums must have 4 elements, got ", sums.sizes());
  TORCH_CHECK(ws.numel() == recon_workspace_floats(bn, R, C), "sgrender: light_objective_stage2: ws must be stage 1's workspace of ",
              recon_workspace_floats(bn, R, C), " floats, got ", ws.sizes());
}

// ---- the same objective under batch sharding: three stage operators with the two collectives between them ----
// stage 1: forward statistics pass + render-loss passes -> sums = [num_r, den_r, 0, den_e] for ONE all-reduce before the backward pass
using T12 = std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor>;
T12 light_objective_stage1_cuda(const Tensor& albedo, const Tensor& normal, const Tensor& rough, const Tensor& axis, const Tensor& lamb, const Tensor& weight,
                                const Tensor& im, const Tensor& seg, const Tensor& env_gt, const Tensor& env_ind, int64_t eh, int64_t ew, double fov, double F0, Cam cam,
                                bool heads, bool handoff) {
  const auto dev = require_hip({&albedo, &normal, &rough, &axis, &lamb, &weight, &im, &seg, &env_gt, &env_ind});
  const c10::DeviceGuard guard(dev);
  const Tensor a = albedo.contiguous(), n = normal.contiguous(), r = rough.contiguous(), ax = axis.contiguous(), la = lamb.contiguous(), we = weight.contiguous();
  const Tensor i = im.contiguous(), sg = seg.contiguous(), gt = env_gt.contiguous(), ind = env_ind.contiguous().reshape({-1});
  const auto d = check_objective(a, n, r, ax, la, we, i, sg, gt, ind, eh, ew);
  const auto o = a.options();
  const Api& A = api();
  void* st = stream_of(dev);
  const int bn = (int)d.bn, K = (int)d.K, R = (int)d.R, C = (int)d.C;
  Tensor diffuse = at::empty({d.bn, 3, d.R, d.C}, o), spec = at::empty({d.bn, 3, d.R, d.C}, o), im_s = at::empty({d.bn, 3, d.R, d.C}, o);
  Tensor seg_s = at::empty({d.bn, 1, d.R, d.C}, o), rendered = at::empty({d.bn, 3, d.R, d.C}, o), mask = at::empty({d.bn, d.R * d.C}, o), coef = at::empty({d.bn}, o);
  Tensor coef_ds = at::empty({d.bn, 2}, o), sums = at::empty({4}, o);
  TORCH_CHECK(A.sgr_fused_recon_workspace_floats(bn, R, C) == recon_workspace_floats(d.bn, d.R, d.C), "sgrender: workspace size disagrees with the kernel library");
  Tensor ws = at::empty({recon_workspace_floats(d.bn, d.R, d.C)}, o), ws_r = loss_workspace(d.bn, a);
  handoff = handoff && !heads;
  Tensor lam_t = handoff ? at::empty_like(la) : none_like(a), w_t = handoff ? at::empty_like(we) : none_like(a);
  const Tensor dirs = dirs_table(dev, eh, ew), view = view_table(dev, d.R, d.C, fov, cam);
  float* sp = sums.data_ptr<float>();
  ok(A.sgr_fused_fwd_recon_seg(rp(a), rp(n), rp(r), rp(ax), rp(la), rp(we), rp(dirs), rp(view), rp(gt), rp(sg), (int)d.imH, (int)d.imW, rp(ind), wp(lam_t), wp(w_t),
                               wp(diffuse), wp(spec), wp(mask), wp(coef), sp + 2 /* (0, env-mask sum) */, wp(ws), bn, K, R, C, (int)eh, (int)ew, (int)d.h, (int)d.w,
                               (float)F0, heads ? 3 : 1, st),
     "sgr_fused_fwd_recon");
  ok(A.sgr_render_loss_fwd_total(rp(diffuse), rp(spec), rp(i), rp(sg), wp(im_s), wp(seg_s), wp(rendered), wp(coef_ds), sp /* (num_r, den_r) */, nullptr, nullptr, 3.0f,
                                 wp(ws_r), bn, R, C, (int)d.imH, (int)d.imW, st),
     "sgr_render_loss_fwd");
  return {diffuse, spec, mask, coef, im_s, seg_s, rendered, coef_ds, sums, ws, lam_t, w_t};
}
T12 light_objective_stage1_meta(const Tensor& albedo, const Tensor& normal, const Tensor& rough, const Tensor& axis, const Tensor& lamb, const Tensor& weight,
                                const Tensor& im, const Tensor& seg, const Tensor& env_gt, const Tensor& env_ind, int64_t eh, int64_t ew, double, double, Cam, bool heads,
                                bool handoff) {
  const auto d = check_objective(albedo, normal, rough, axis, lamb, weight, im, seg, env_gt, env_ind, eh, ew);
  const auto o = albedo.options();
  const bool h = handoff && !heads;
  auto img = [&] { return at::empty({d.bn, 3, d.R, d.C}, o); };
  return {img(), img(), at::empty({d.bn, d.R * d.C}, o), at::empty({d.bn}, o), img(), at::empty({d.bn, 1, d.R, d.C}, o), img(), at::empty({d.bn, 2}, o), at::empty({4}, o),
          at::empty({recon_workspace_floats(d.bn, d.R, d.C)}, o), h ? at::empty(lamb.sizes(), o) : none_like(albedo), h ? at::empty(weight.sizes(), o) : none_like(albedo)};
}
// stage 2 (sums = the rank-summed vector): render-loss value + backward with the global normaliser, the objective's backward pass
// with the global env-mask sum -> (render_err, g_axis, g_lamb, g_weight, parts_b = (num_e of this shard, its mask sum))
T5 light_objective_stage2_cuda(const Tensor& albedo, const Tensor& normal, const Tensor& rough, const Tensor& axis, const Tensor& lamb, const Tensor& weight,
                               const Tensor& env_gt, const Tensor& mask, const Tensor& coef, const Tensor& diffuse, const Tensor& spec, const Tensor& im_s,
                               const Tensor& seg_s, const Tensor& coef_ds, const Tensor& sums, Tensor& ws /* written: schema Tensor(a!) */, const Tensor& lam_t, const Tensor& w_t, int64_t eh,
                               int64_t ew, double fov, double F0, Cam cam, double ren_w, double rec_w, double offset, bool heads, bool need_grad) {
  const auto dev = require_hip({&albedo, &normal, &rough, &axis, &lamb, &weight, &env_gt, &mask, &coef, &diffuse, &spec, &im_s, &seg_s, &coef_ds, &sums, &ws});
  const c10::DeviceGuard guard(dev);
  const Tensor a = albedo.contiguous(), n = normal.contiguous(), r = rough.contiguous(), ax = axis.contiguous(), la = lamb.contiguous(), we = weight.contiguous();
  const Tensor gt = env_gt.contiguous();
  const auto d = check_sg(ax, la, we);
  const auto b = check_brdf(a, n, r);
  TORCH_CHECK(sums.is_contiguous() && sums.numel() == 4 && mask.is_contiguous() && coef.is_contiguous() && diffuse.is_contiguous() && spec.is_contiguous() &&
                  im_s.is_contiguous() && seg_s.is_contiguous() && coef_ds.is_contiguous() && ws.is_contiguous(), "sgrender: light_objective_stage2 takes stage 1's tensors as they are");
  check_stage1_tensors(d.bn, d.R, d.C, mask, coef, diffuse, spec, im_s, seg_s, coef_ds, sums, ws);
  TORCH_CHECK(gt.numel() == d.bn * 3 * d.R * d.C * eh * ew, "sgrender: light_objective_stage2: env_gt must be [bn,3,R,C,eh,ew], got ", gt.sizes());
  const auto o = a.options();
  const Api& A = api();
  void* st = stream_of(dev);
  const int bn = (int)d.bn, K = (int)d.K, R = (int)d.R, C = (int)d.C;
  const bool handoff = present(lam_t) && present(w_t) && need_grad;
  Tensor render_err = at::empty({}, o), scale_r = at::empty({1}, o), parts_b = at::empty({2}, o);
  const float* sp = sums.const_data_ptr<float>();
  ok(A.sgr_loss_finalize(sp, render_err.data_ptr<float>(), wp(scale_r), 3.0f, st), "sgr_loss_finalize");
  Tensor g_axis = none_like(a), g_lamb = none_like(a), g_weight = none_like(a), g_d, g_s;
  if (need_grad) {
    g_axis = at::empty_like(ax); g_lamb = at::empty_like(la); g_weight = at::empty_like(we);
    g_d = at::empty_like(diffuse); g_s = at::empty_like(spec);
    ok(A.sgr_render_loss_bwd_scaled(nullptr, (float)ren_w, rp(scale_r), rp(diffuse), rp(spec), rp(im_s), rp(seg_s), rp(coef_ds), wp(g_d), wp(g_s), bn, R, C, st),
       "sgr_render_loss_bwd");
  }
  const Tensor dirs = dirs_table(dev, eh, ew), view = view_table(dev, d.R, d.C, fov, cam);
  ok(A.sgr_fused_bwd_recon(rp(a), rp(n), rp(r), rp(ax), handoff ? rp(lam_t) : rp(la), handoff ? rp(w_t) : rp(we), rp(dirs), rp(view), rp(gt), rp(mask), rp(coef),
                           sp + 3 /* the global env-mask sum */, rp(g_d), rp(g_s), wp(g_axis), wp(g_lamb), wp(g_weight), wp(parts_b), ws.mutable_data_ptr<float>(), bn, K, R,
                           C, (int)eh, (int)ew, (int)b.h, (int)b.w, (float)F0, handoff ? 2 : (heads ? 3 : 1), (float)offset, (float)rec_w, st),
     "sgr_fused_bwd_recon");
  return {render_err, g_axis, g_lamb, g_weight, parts_b};
}
T5 light_objective_stage2_meta(const Tensor& albedo, const Tensor& normal, const Tensor& rough, const Tensor& axis, const Tensor& lamb, const Tensor& weight, const Tensor&,
                               const Tensor& mask, const Tensor& coef, const Tensor& diffuse, const Tensor& spec, const Tensor& im_s, const Tensor& seg_s, const Tensor& coef_ds,
                               const Tensor& sums, Tensor& ws, const Tensor&, const Tensor&, int64_t, int64_t, double, double, Cam, double, double, double, bool, bool need_grad) {
  const auto d = check_sg(axis, lamb, weight);
  check_brdf(albedo, normal, rough);
  check_stage1_tensors(d.bn, d.R, d.C, mask, coef, diffuse, spec, im_s, seg_s, coef_ds, sums, ws);
  const auto o = albedo.options();
  auto g = [&](const Tensor& t) { return need_grad ? at::empty(t.sizes(), o) : none_like(albedo); };
  return {at::empty({}, o), g(axis), g(lamb), g(weight), at::empty({2}, o)};
}
// stage 3 (num_e = the rank-summed reconstruction numerator): the objective's scalar tail
T2 light_objective_stage3_cuda(const Tensor& render_err, const Tensor& num_e, const Tensor& sums, double ren_w, double rec_w, int64_t eh, int64_t ew) {
  const auto dev = require_hip({&render_err, &num_e, &sums});
  const c10::DeviceGuard guard(dev);
  TORCH_CHECK(num_e.numel() >= 1 && sums.numel() == 4 && sums.is_contiguous(), "sgrender: light_objective_stage3 arguments");
  const auto o = render_err.options();
  Tensor pe = at::empty({2}, o), objective = at::empty({}, o), recon_err = at::empty({}, o);
  pe.slice(0, 0, 1).copy_(num_e.reshape({-1}).slice(0, 0, 1));
  pe.slice(0, 1, 2).copy_(sums.slice(0, 3, 4));
  ok(api().sgr_objective_finalize(rp(render_err), rp(pe), (float)ren_w, (float)rec_w, 3.0f * (float)(eh * ew), objective.data_ptr<float>(), recon_err.data_ptr<float>(),
                                  stream_of(dev)),
     "sgr_objective_finalize");
  return {objective, recon_err};
}
T2 light_objective_stage3_meta(const Tensor& render_err, const Tensor&, const Tensor&, double, double, int64_t, int64_t) {
  return {at::empty({}, render_err.options()), at::empty({}, render_err.options())};
}

// ------------------------------------------------------------------------------------------------------------------
void no_cpu_path(const c10::OperatorHandle&, torch::jit::Stack*) { TORCH_CHECK(false, kNoCpu); }

}  // namespace

// ==================================================================================================================
// in-stream collectives ("optional native variant: ncclAllReduce on the same HIP stream from the extension")
// ==================================================================================================================
// The batch-sharded losses couple the ranks through one all-reduce of a handful of floats between two kernels of the step
// (wrapperBRDFLight.py:192,205-207 under sharding).  Through c10d that is a Python call, ProcessGroupNCCL's bookkeeping and -- at
// world 1, measured in round 5 -- 9 us per all-reduce on the step.  Here the extension owns an RCCL communicator per process group
// (bootstrapped once by the host layer: ncclGetUniqueId on rank 0, the 128 bytes broadcast through the existing c10d group) and
// `allreduce_sum_` enqueues ncclAllReduce on the CURRENT HIP stream: no side stream, no event pair, capturable in a HIP graph.
// The RCCL used is the one already in the process (PyTorch-ROCm links it): found among the loaded objects, never a second copy.
struct Rccl {
  decltype(&::ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&::ncclCommInitRank) CommInitRank = nullptr;
  decltype(&::ncclAllReduce) AllReduce = nullptr;
  decltype(&::ncclCommDestroy) CommDestroy = nullptr;
  decltype(&::ncclGetErrorString) GetErrorString = nullptr;
  std::string path;
};
int find_rccl(struct dl_phdr_info* info, size_t, void* out) {
  if (info->dlpi_name && std::strstr(info->dlpi_name, "librccl")) {
    *static_cast<std::string*>(out) = info->dlpi_name;
    return 1;
  }
  return 0;
}
const Rccl& rccl() {
  static const Rccl r = [] {
    Rccl x;
    dl_iterate_phdr(&find_rccl, &x.path);
    TORCH_CHECK(!x.path.empty(), "sgrender: no librccl in this process (PyTorch-ROCm loads it with its distributed backend); the in-stream all-reduce needs it");
    void* h = dlopen(x.path.c_str(), RTLD_NOW | RTLD_NOLOAD);
    TORCH_CHECK(h, "sgrender: cannot re-open ", x.path, ": ", dlerror());
#define SGR_RCCL(name)                                                                \
  x.name = reinterpret_cast<decltype(x.name)>(dlsym(h, "nccl" #name));                \
  TORCH_CHECK(x.name, "sgrender: ", x.path, " does not export nccl" #name);
    SGR_RCCL(GetUniqueId) SGR_RCCL(CommInitRank) SGR_RCCL(AllReduce) SGR_RCCL(CommDestroy) SGR_RCCL(GetErrorString)
#undef SGR_RCCL
    return x;
  }();
  return r;
}
void rccl_ok(ncclResult_t rc, const char* what) { TORCH_CHECK(rc == ncclSuccess, "sgrender: ", what, " failed: ", rccl().GetErrorString(rc)); }

struct CommEntry { ncclComm_t comm; int device, rank, world; };
std::mutex g_comm_mutex;
std::map<int64_t, CommEntry> g_comms;
int64_t g_next_comm = 1;

Tensor comm_unique_id() {
  static_assert(sizeof(ncclUniqueId) == NCCL_UNIQUE_ID_BYTES, "ncclUniqueId is an opaque byte array");
  ncclUniqueId id;
  rccl_ok(rccl().GetUniqueId(&id), "ncclGetUniqueId");
  Tensor t = at::empty({(int64_t)sizeof(id)}, at::TensorOptions().dtype(at::kByte));
  std::memcpy(t.data_ptr<uint8_t>(), &id, sizeof(id));
  return t;
}
// collective over the ranks of the communicator being made (ncclCommInitRank synchronises with them); device = this rank's GPU
int64_t comm_init(const Tensor& uid, int64_t rank, int64_t world, int64_t device) {
  TORCH_CHECK(uid.device().is_cpu() && uid.scalar_type() == at::kByte && uid.is_contiguous() && uid.numel() == (int64_t)sizeof(ncclUniqueId),
              "sgrender: comm_init needs the ", sizeof(ncclUniqueId), " bytes of comm_unique_id() as a CPU uint8 tensor");
  TORCH_CHECK(world >= 1 && rank >= 0 && rank < world, "sgrender: comm_init: rank ", rank, " of ", world);
  ncclUniqueId id;
  std::memcpy(&id, uid.const_data_ptr<uint8_t>(), sizeof(id));
  const c10::DeviceGuard guard(c10::Device(c10::kCUDA, (c10::DeviceIndex)device));
  ncclComm_t comm = nullptr;
  rccl_ok(rccl().CommInitRank(&comm, (int)world, id, (int)rank), "ncclCommInitRank");
  std::lock_guard<std::mutex> lock(g_comm_mutex);
  const int64_t h = g_next_comm++;
  g_comms[h] = CommEntry{comm, (int)device, (int)rank, (int)world};
  return h;
}
void comm_destroy(int64_t handle) {
  CommEntry e{};
  {
    std::lock_guard<std::mutex> lock(g_comm_mutex);
    auto it = g_comms.find(handle);
    if (it == g_comms.end()) return;
    e = it->second;
    g_comms.erase(it);
  }
  const c10::DeviceGuard guard(c10::Device(c10::kCUDA, (c10::DeviceIndex)e.device));
  rccl_ok(rccl().CommDestroy(e.comm), "ncclCommDestroy");
}
int64_t comm_world_size(int64_t handle) {
  std::lock_guard<std::mutex> lock(g_comm_mutex);
  auto it = g_comms.find(handle);
  TORCH_CHECK(it != g_comms.end(), "sgrender: unknown communicator handle ", handle);
  return it->second.world;
}
// t <- sum over the ranks of t, enqueued on the current stream of t's device (fp32 / fp64, contiguous)
void allreduce_sum_cuda(Tensor& t, int64_t handle) {
  TORCH_CHECK(t.is_cuda() && t.is_contiguous(), "sgrender: allreduce_sum_ needs a contiguous device tensor");
  TORCH_CHECK(t.scalar_type() == at::kFloat || t.scalar_type() == at::kDouble, "sgrender: allreduce_sum_ needs fp32 or fp64, got ", t.scalar_type());
  CommEntry e{};
  {
    std::lock_guard<std::mutex> lock(g_comm_mutex);
    auto it = g_comms.find(handle);
    TORCH_CHECK(it != g_comms.end(), "sgrender: unknown communicator handle ", handle);
    e = it->second;
  }
  TORCH_CHECK(t.device().index() == e.device, "sgrender: allreduce_sum_: the tensor lives on device ", (int)t.device().index(), ", the communicator on ", e.device);
  if (t.numel() == 0) return;
  const c10::DeviceGuard guard(t.device());
  rccl_ok(rccl().AllReduce(t.data_ptr(), t.data_ptr(), (size_t)t.numel(), t.scalar_type() == at::kFloat ? ncclFloat32 : ncclFloat64, ncclSum, e.comm,
                           (hipStream_t)stream_of(t.device())),
          "ncclAllReduce");
}
void allreduce_sum_meta(Tensor&, int64_t) {}

TORCH_LIBRARY(sgrender, m) {
  m.def("sg_to_env(Tensor axis, Tensor lamb, Tensor weight, int eh, int ew, bool premap, bool want_tan=False) -> (Tensor, Tensor, Tensor)");
  m.def("sg_to_env_bwd(Tensor g_env, Tensor axis, Tensor lamb, Tensor weight, int eh, int ew, int premap) -> (Tensor, Tensor, Tensor)");
  m.def("render_env(Tensor albedo, Tensor normal, Tensor rough, Tensor env, float fov, float F0, float[] cam) -> (Tensor, Tensor)");
  m.def("render_env_bwd_env(Tensor g_diffuse, Tensor g_spec, Tensor albedo, Tensor normal, Tensor rough, int eh, int ew, float fov, float F0, float[] cam) -> Tensor");
  m.def("render_bwd_brdf(Tensor g_diffuse, Tensor g_spec, Tensor albedo, Tensor normal, Tensor rough, Tensor? env, Tensor? axis, Tensor? lamb, Tensor? weight, "
        "int eh, int ew, float fov, float F0, float[] cam, bool premap) -> (Tensor, Tensor, Tensor)");
  m.def("fused_render(Tensor albedo, Tensor normal, Tensor rough, Tensor axis, Tensor lamb, Tensor weight, int eh, int ew, float fov, float F0, float[] cam, "
        "int premap, bool need_env, bool want_tan=False) -> (Tensor, Tensor, Tensor, Tensor, Tensor)");
  m.def("fused_render_bwd_sg(Tensor? g_env, Tensor g_diffuse, Tensor g_spec, Tensor albedo, Tensor normal, Tensor rough, Tensor axis, Tensor lamb, Tensor weight, "
        "int eh, int ew, float fov, float F0, float[] cam, int premap) -> (Tensor, Tensor, Tensor)");
  m.def("lsregress_coef(Tensor pred, Tensor gt) -> Tensor");
  m.def("lsregress_diffspec_coef(Tensor diff, Tensor spec, Tensor im) -> Tensor");
  m.def("render_loss(Tensor diffuse, Tensor spec, Tensor im, Tensor seg, int R, int C, bool total) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)");
  m.def("render_loss_finalize(Tensor diffuse, Tensor spec, Tensor parts, Tensor im_s, Tensor seg_s, Tensor coef) -> (Tensor, Tensor)");
  m.def("render_loss_bwd(Tensor? g_loss, float weight, Tensor? scale, Tensor diffuse, Tensor spec, Tensor im_s, Tensor seg_s, Tensor coef) -> (Tensor, Tensor)");
  m.def("recon_loss_parts(Tensor env, Tensor env_gt, Tensor seg_small, Tensor env_ind, float offset) -> (Tensor, Tensor, Tensor)");
  m.def("recon_loss_bwd(Tensor g_num, Tensor env, Tensor env_gt, Tensor mask, Tensor coef, float offset) -> Tensor");
  m.def("light_heads(Tensor x_axis, Tensor x_lamb, Tensor x_weight, bool need_packed) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def("light_heads_bwd(Tensor x_axis, Tensor x_lamb, Tensor x_weight, Tensor? g_axis, Tensor? g_lamb, Tensor? g_weight, Tensor? g_packed) -> (Tensor, Tensor, Tensor)");
  m.def("sg_shading(Tensor axis, Tensor lamb, Tensor weight, int eh, int ew, int premap) -> Tensor");
  m.def("light_albedo_scale(Tensor diffuse_scaled, Tensor diffuse, Tensor spec_scaled, Tensor spec, Tensor albedo) -> Tensor");
  m.def("light_encoder_input(Tensor im, Tensor albedo, Tensor normal, Tensor rough, Tensor depth, int H, int W) -> (Tensor, Tensor, Tensor)");
  m.def("rescale_grads_(Tensor(a!) g_axis, Tensor(b!) g_lamb, Tensor(c!) g_weight, Tensor scale, Tensor(d!) applied, int parity) -> ()");
  m.def("attach_grads(Tensor value, Tensor axis, Tensor lamb, Tensor weight, Tensor g_axis, Tensor g_lamb, Tensor g_weight, Tensor applied) -> Tensor");
  m.def("light_objective_fwdbwd(Tensor albedo, Tensor normal, Tensor rough, Tensor axis, Tensor lamb, Tensor weight, Tensor im, Tensor seg, Tensor env_gt, Tensor env_ind, "
        "int eh, int ew, float fov, float F0, float[] cam, float ren_w, float rec_w, float offset, bool heads, bool handoff, bool need_grad) -> "
        "(Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)");
  m.def("light_objective(Tensor albedo, Tensor normal, Tensor rough, Tensor axis, Tensor lamb, Tensor weight, Tensor im, Tensor seg, Tensor env_gt, Tensor env_ind, "
        "int eh, int ew, float fov, float F0, float[] cam, float ren_w, float rec_w, float offset, bool heads, bool handoff) -> (Tensor, Tensor, Tensor, Tensor, Tensor)");
  m.def("light_objective_stage1(Tensor albedo, Tensor normal, Tensor rough, Tensor axis, Tensor lamb, Tensor weight, Tensor im, Tensor seg, Tensor env_gt, Tensor env_ind, "
        "int eh, int ew, float fov, float F0, float[] cam, bool heads, bool handoff) -> "
        "(Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)");
  m.def("light_objective_stage2(Tensor albedo, Tensor normal, Tensor rough, Tensor axis, Tensor lamb, Tensor weight, Tensor env_gt, Tensor mask, Tensor coef, "
        "Tensor diffuse, Tensor spec, Tensor im_s, Tensor seg_s, Tensor coef_ds, Tensor sums, Tensor(a!) ws, Tensor lam_t, Tensor w_t, int eh, int ew, float fov, float F0, "
        "float[] cam, float ren_w, float rec_w, float offset, bool heads, bool need_grad) -> (Tensor, Tensor, Tensor, Tensor, Tensor)");
  m.def("light_objective_stage3(Tensor render_err, Tensor num_e, Tensor sums, float ren_w, float rec_w, int eh, int ew) -> (Tensor, Tensor)");
  // host-side queries (no tensors, no dispatch key): cached constant tables of a kind (test hook), the objective's workspace size
  m.def("table_cache_count(int kind) -> int", &table_cache_count);
  m.def("cached_tables() -> Tensor[]", &cached_tables);
  m.def("recon_workspace_floats(int bn, int R, int C) -> int", &recon_workspace_floats);
  // in-stream collectives: communicator management is host-side (no dispatch key); the all-reduce itself is a device operator
  m.def("comm_unique_id() -> Tensor", &comm_unique_id);
  m.def("comm_init(Tensor uid, int rank, int world, int device) -> int", &comm_init);
  m.def("comm_destroy(int comm) -> ()", &comm_destroy);
  m.def("comm_world_size(int comm) -> int", &comm_world_size);
  m.def("allreduce_sum_(Tensor(a!) t, int comm) -> ()");
}

TORCH_LIBRARY_IMPL(sgrender, CUDA, m) {
  m.impl("sg_to_env", &sg_to_env_cuda);
  m.impl("sg_to_env_bwd", &sg_to_env_bwd_cuda);
  m.impl("render_env", &render_env_cuda);
  m.impl("render_env_bwd_env", &render_env_bwd_env_cuda);
  m.impl("render_bwd_brdf", &render_bwd_brdf_cuda);
  m.impl("fused_render", &fused_render_cuda);
  m.impl("fused_render_bwd_sg", &fused_render_bwd_sg_cuda);
  m.impl("lsregress_coef", &lsregress_coef_cuda);
  m.impl("lsregress_diffspec_coef", &lsregress_diffspec_coef_cuda);
  m.impl("render_loss", &render_loss_cuda);
  m.impl("render_loss_bwd", &render_loss_bwd_cuda);
  m.impl("render_loss_finalize", &render_loss_finalize_cuda);
  m.impl("recon_loss_parts", &recon_loss_parts_cuda);
  m.impl("recon_loss_bwd", &recon_loss_bwd_cuda);
  m.impl("light_heads", &light_heads_cuda);
  m.impl("light_heads_bwd", &light_heads_bwd_cuda);
  m.impl("sg_shading", &sg_shading_cuda);
  m.impl("light_albedo_scale", &light_albedo_scale_cuda);
  m.impl("light_encoder_input", &light_encoder_input_cuda);
  m.impl("rescale_grads_", &rescale_grads_cuda);
  m.impl("attach_grads", &attach_grads_backend);
  m.impl("light_objective_fwdbwd", &light_objective_fwdbwd_cuda);
  m.impl("light_objective", &light_objective_backend);
  m.impl("light_objective_stage1", &light_objective_stage1_cuda);
  m.impl("light_objective_stage2", &light_objective_stage2_cuda);
  m.impl("light_objective_stage3", &light_objective_stage3_cuda);
  m.impl("allreduce_sum_", &allreduce_sum_cuda);
}

TORCH_LIBRARY_IMPL(sgrender, Meta, m) {
  m.impl("sg_to_env", &sg_to_env_meta);
  m.impl("sg_to_env_bwd", &sg_to_env_bwd_meta);
  m.impl("render_env", &render_env_meta);
  m.impl("render_env_bwd_env", &render_env_bwd_env_meta);
  m.impl("render_bwd_brdf", &render_bwd_brdf_meta);
  m.impl("fused_render", &fused_render_meta);
  m.impl("fused_render_bwd_sg", &fused_render_bwd_sg_meta);
  m.impl("lsregress_coef", &lsregress_coef_meta);
  m.impl("lsregress_diffspec_coef", &lsregress_diffspec_coef_meta);
  m.impl("render_loss", &render_loss_meta);
  m.impl("render_loss_bwd", &render_loss_bwd_meta);
  m.impl("render_loss_finalize", &render_loss_finalize_meta);
  m.impl("recon_loss_parts", &recon_loss_parts_meta);
  m.impl("recon_loss_bwd", &recon_loss_bwd_meta);
  m.impl("light_heads", &light_heads_meta);
  m.impl("light_heads_bwd", &light_heads_bwd_meta);
  m.impl("sg_shading", &sg_shading_meta);
  m.impl("light_albedo_scale", &light_albedo_scale_meta);
  m.impl("light_encoder_input", &light_encoder_input_meta);
  m.impl("rescale_grads_", &rescale_grads_meta);
  m.impl("attach_grads", &attach_grads_backend);
  m.impl("light_objective_fwdbwd", &light_objective_fwdbwd_meta);
  m.impl("light_objective", &light_objective_backend);
  m.impl("light_objective_stage1", &light_objective_stage1_meta);
  m.impl("light_objective_stage2", &light_objective_stage2_meta);
  m.impl("light_objective_stage3", &light_objective_stage3_meta);
  m.impl("allreduce_sum_", &allreduce_sum_meta);
}

TORCH_LIBRARY_IMPL(sgrender, Autograd, m) {
  m.impl("sg_to_env", &sg_to_env_autograd);
  m.impl("render_env", &render_env_autograd);
  m.impl("fused_render", &fused_render_autograd);
  m.impl("render_loss", &render_loss_autograd);
  m.impl("render_loss_finalize", &render_loss_finalize_autograd);
  m.impl("recon_loss_parts", &recon_loss_parts_autograd);
  m.impl("light_heads", &light_heads_autograd);
  m.impl("attach_grads", &attach_grads_autograd);
  m.impl("light_objective", &light_objective_autograd);
}

// no CPU path: every operator raises on CPU tensors (a namespace cannot carry a backend fallback, hence one registration each)
TORCH_LIBRARY_IMPL(sgrender, CPU, m) {
  for (const char* name : {"sg_to_env", "sg_to_env_bwd", "render_env", "render_env_bwd_env", "render_bwd_brdf", "fused_render", "fused_render_bwd_sg", "lsregress_coef",
                           "lsregress_diffspec_coef", "render_loss", "render_loss_bwd", "render_loss_finalize", "recon_loss_parts", "recon_loss_bwd", "light_heads", "light_heads_bwd", "sg_shading",
                           "light_albedo_scale", "light_encoder_input", "rescale_grads_", "attach_grads", "light_objective_fwdbwd", "light_objective",
                           "light_objective_stage1", "light_objective_stage2", "light_objective_stage3", "allreduce_sum_"})
    m.impl(name, torch::CppFunction::makeFromBoxedFunction<&no_cpu_path>());
}
