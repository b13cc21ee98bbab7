// Host-side error plumbing shared by the C-ABI entry points.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/sgrender.h"

namespace sgr {
void set_error(const char* msg);
int sgr_check(int hip_rc, const char* who);
void note_pending_error();

// The env-statistics fold of the fused light objective (recon_fold0_image, sgr_recon_fold.h) as a side job of the render loss's first pass:
// one extra workgroup per image.  ws == nullptr: no job.
struct FoldJob { const float* ws; float* coef; float* den_img; int nblk; };
// the three render-loss passes (sgr_loss.hip), shared by sgr_render_loss_fwd_total(_grads) and sgr_light_objective_fwd (sgr_fused_recon.hip)
int render_loss_fwd_launch(const float* diffuse, const float* spec, const float* im, const float* seg, float* im_small, float* seg_small,
                           float* rendered, float* coef, float* parts, float* loss, float* scale, float divisor, float weight,
                           float* g_diffuse, float* g_spec, float* workspace, int bn, int R, int C, int imH, int imW, FoldJob job, void* stream);
}  // namespace sgr

// Every entry point starts with SGR_REQUIRE.  Its first act is note_pending_error(): a (non-sticky) HIP error some earlier
// call on this thread left pending -- ours or another library's -- is taken off the error slot so that the hipGetLastError()
// after OUR launch reports our launch and nothing else, but it is not discarded: it is remembered (thread-local) and
// appended to the message of the next failure this library reports, and sgr_last_error() shows it until then.
#define SGR_REQUIRE(cond, msg)          \
  do {                                  \
    ::sgr::note_pending_error();        \
    if (!(cond)) {                      \
      ::sgr::set_error(msg);            \
      return SGR_ERR_BAD_ARG;           \
    }                                   \
  } while (0)

#define SGR_SUPPORTED(cond, msg)        \
  do {                                  \
    if (!(cond)) {                      \
      ::sgr::set_error(msg);            \
      return SGR_ERR_UNSUPPORTED;       \
    }                                   \
  } while (0)

// SGR_GENERIC=1 forces the table-driven generic kernels (tuning / test knob); read once per process
#include <stdlib.h>
static inline bool sgr_generic_forced() {
  static const bool on = [] { const char* e = getenv("SGR_GENERIC"); return e != nullptr && e[0] != 0 && !(e[0] == '0' && e[1] == 0); }();
  return on;
}
