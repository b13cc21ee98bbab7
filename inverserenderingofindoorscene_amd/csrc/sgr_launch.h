// Host-side error plumbing shared by the C-ABI entry points.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/sgrender.h"

namespace sgr {
void set_error(const char* msg);
int sgr_check(int hip_rc, const char* who);
}  // namespace sgr

// Every entry point starts with SGR_REQUIRE: it also drops a (non-sticky) error some other library left pending on this
// thread, so that the hipGetLastError() after our own launch reports OUR launch and nothing else.
#define SGR_REQUIRE(cond, msg)          \
  do {                                  \
    (void)hipGetLastError();            \
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
