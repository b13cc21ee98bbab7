// Host-side error plumbing shared by the C-ABI entry points.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/sgrender.h"

namespace sgr {
void set_error(const char* msg);
int sgr_check(int hip_rc, const char* who);
}  // namespace sgr

#define SGR_REQUIRE(cond, msg)          \
  do {                                  \
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
  static const bool on = [] { const char* e = getenv("SGR_GENERIC"); return e != nullptr; }();
  return on;
}
