// ABI version, error reporting and the host-side constant tables of the render path.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "sgr_launch.h"

namespace sgr {

static thread_local char g_err[384] = "";
static thread_local char g_foreign[128] = "";     // a HIP error found pending at one of our entry points (not caused by it)

void note_pending_error() {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_foreign, sizeof(g_foreign), " [a HIP error was already pending on this thread when sgrender was entered: %s]", hipGetErrorString(e));
    if (!g_err[0]) snprintf(g_err, sizeof(g_err), "no sgrender failure%s", g_foreign);
  }
}

void set_error(const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s%s", msg, g_foreign);
  g_foreign[0] = 0;
}

int sgr_check(int hip_rc, const char* who) {
  if (hip_rc != 0) {
    snprintf(g_err, sizeof(g_err), "%s: %s%s", who, hipGetErrorString((hipError_t)hip_rc), g_foreign);
    g_foreign[0] = 0;
  }
  return hip_rc;
}

}  // namespace sgr

extern "C" int sgr_abi_version(void) { return SGR_ABI_VERSION; }
extern "C" const char* sgr_last_error(void) { return sgr::g_err; }

extern "C" int sgr_dirs_padded(int J) { return (J + 31) / 32 * 32; }
static int rows_padded(int eh) { return (eh + 1) / 2 * 2; }
extern "C" int sgr_dirs_floats(int eh, int ew) { return 4 * sgr_dirs_padded(eh * ew) + 8 * rows_padded(eh) + 8 * ew; }

// models.py:353-363 (output2env.__init__) and models.py:437-452 (renderingLayer.__init__):
// float64 arithmetic, results stored as float32; the evaluation order of the scalar
// expressions follows the Python source so the rounded values are the same.
extern "C" int sgr_fill_direction_table(float* out, int eh, int ew) {
  SGR_REQUIRE(out && eh > 0 && ew > 0, "sgr_fill_direction_table: bad argument");
  const int J = eh * ew, Jp = sgr_dirs_padded(J);
  memset(out, 0, sizeof(float) * (size_t)sgr_dirs_floats(eh, ew));
  // separable form: l_j = (s_e ca_a, s_e sa_a, c_e); rows (s, c, omega, s^2, 2sc, c^2), cols (ca, sa, ca^2, 2 ca sa, sa^2)
  float* rows = out + 4 * (size_t)Jp;
  float* cols = rows + 8 * (size_t)rows_padded(eh);
  for (int e = 0; e < eh; ++e) {
    const double el = (((double)e + 0.5) / (double)eh) * M_PI / 2.0;
    const double sd = sin(el), cd = cos(el);
    float* o = rows + 8 * (size_t)e;
    o[0] = (float)sd; o[1] = (float)cd; o[2] = (float)(sd * M_PI * M_PI / (double)ew / (double)eh);
    o[3] = (float)(sd * sd); o[4] = (float)(2.0 * sd * cd); o[5] = (float)(cd * cd);
  }
  // cols: first half row only (the second half is its negation): [ew/2][2] = (ca, sa), then at float
  // offset ew: [ew/2][4] = (ca^2, 2 ca sa, sa^2, 0)
  for (int a = 0; a < ew / 2; ++a) {
    const double az = ((((double)a + 0.5) / (double)ew) - 0.5) * 2.0 * M_PI;
    const double cad = cos(az), sad = sin(az);
    cols[2 * a] = (float)cad; cols[2 * a + 1] = (float)sad;
    float* o = cols + ew + 4 * (size_t)a;
    o[0] = (float)(cad * cad); o[1] = (float)(2.0 * cad * sad); o[2] = (float)(sad * sad);
    // at float offset 4*ew: [ew/4][4] = (ca_a, ca_a+1, sa_a, sa_a+1) per azimuth pair (packed-math kernels, sgr_pk.inl)
    float* pr = cols + 4 * (size_t)ew + 4 * (size_t)(a / 2);
    pr[a & 1] = (float)cad; pr[2 + (a & 1)] = (float)sad;
  }
  for (int e = 0; e < eh; ++e) {
    const double el = (((double)e + 0.5) / (double)eh) * M_PI / 2.0;
    for (int a = 0; a < ew; ++a) {
      const double az = ((((double)a + 0.5) / (double)ew) - 0.5) * 2.0 * M_PI;
      float* o = out + 4 * (size_t)(e * ew + a);
      o[0] = (float)(sin(el) * cos(az));
      o[1] = (float)(sin(el) * sin(az));
      o[2] = (float)cos(el);
      o[3] = (float)(sin(el) * M_PI * M_PI / (double)ew / (double)eh);
    }
  }
  return SGR_OK;
}

static double linspace_at(double start, double stop, int n, int i) {
  if (n == 1) return start;
  if (i == n - 1) return stop;
  const double step = (stop - start) / (double)(n - 1);
  return (double)i * step + start;
}

// models.py:415-430.  The pixel grid is float64 -> float32; the camera offset and the
// normalisation are float32 operations.
extern "C" int sgr_fill_view_vectors(float* out, int R, int C, float fov_deg, const float* cam) {
  SGR_REQUIRE(out && R > 0 && C > 0, "sgr_fill_view_vectors: bad argument");
  const double fov = (double)fov_deg / 180.0 * M_PI;
  const double xr = 1.0 * tan(fov / 2.0);
  const double yr = (double)R / (double)C * xr;
  const float c0 = cam ? cam[0] : 0.f, c1 = cam ? cam[1] : 0.f, c2 = cam ? cam[2] : 0.f;
  const size_t RC = (size_t)R * C;
  for (int r = 0; r < R; ++r) {
    const float py = (float)linspace_at(-yr, yr, R, R - 1 - r);   // np.flip(y, axis=0)
    for (int c = 0; c < C; ++c) {
      const float px = (float)linspace_at(-xr, xr, C, c);
      const float vx = c0 - px, vy = c1 - py, vz = c2 - (-1.0f);
      const float nn = fmaxf((vx * vx + vy * vy) + vz * vz, 1e-12f);
      const float n = sqrtf(nn);
      out[0 * RC + (size_t)r * C + c] = vx / n;
      out[1 * RC + (size_t)r * C + c] = vy / n;
      out[2 * RC + (size_t)r * C + c] = vz / n;
    }
  }
  return SGR_OK;
}
