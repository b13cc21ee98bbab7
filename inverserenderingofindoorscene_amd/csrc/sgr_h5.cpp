// libsgrender_h5.so: the cascade hand-off container (include/sgrender_h5.h) -- float32 "data" datasets in HDF5 files with h5py's LZF
// filter, what utils.py:92-99 writes and dataLoader.py:277-283 reads in the reference.  Host code only (g++; no HIP, no torch).
//
// libhdf5 is located at RUN time (dlopen) and driven through a dozen of its C entry points, declared below with the HDF5 1.10+ ABI
// (hid_t is a 64-bit integer since 1.10) so that building this file needs no HDF5 headers.  The LZF coder is this file's own: the stream
// format of liblzf (what h5py's filter 32000 links) is small and public --
//     ctrl < 32          : a run of ctrl + 1 literal bytes follows
//     ctrl >= 32         : back reference; len = ctrl >> 5 (7: + the next byte), offset = ((ctrl & 31) << 8 | next byte) + 1;
//                          copy len + 2 bytes from `offset` bytes back in the OUTPUT, byte by byte (overlap = run-length coding)
// -- and any stream in it decodes anywhere, so the encoder here (one hash probe per position over a 64K-entry table of 3-byte
// prefixes, greedy, matches of 3..264 bytes within 8 KiB) need not produce liblzf's exact bytes.
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/sgrender_h5.h"

// ---------------------------------------------------------------------------------------------------------------------------------------
// LZF
// ---------------------------------------------------------------------------------------------------------------------------------------
namespace {
constexpr size_t kMaxLit = 32, kMaxOff = 1u << 13, kMaxRef = (1u << 8) + (1u << 3);      // literal run, window, match length (264)
constexpr unsigned kHashBits = 16;

inline uint32_t hash3(const uint8_t* p) {
  const uint32_t v = ((uint32_t)p[0] << 16) | ((uint32_t)p[1] << 8) | p[2];
  return (v * 2654435761u) >> (32 - kHashBits);
}
}  // namespace

extern "C" size_t sgr_lzf_compress(const void* in_, size_t in_len, void* out_, size_t out_len) {
  const uint8_t* in = static_cast<const uint8_t*>(in_);
  uint8_t* out = static_cast<uint8_t*>(out_);
  if (!in || !out || in_len == 0 || out_len < 2) return 0;
  std::vector<uint32_t> htab((size_t)1 << kHashBits, 0xffffffffu);      // position of the last occurrence of a 3-byte prefix
  size_t ip = 0, op = 1, lit = 0;                                        // out[op - lit - 1] is the pending literal run's control byte
  auto end_run = [&]() {                                                 // close the pending literal run (drop its control byte if empty)
    if (lit) out[op - lit - 1] = (uint8_t)(lit - 1); else --op;
  };
  auto put_literal = [&](uint8_t b) -> bool {
    if (op >= out_len) return false;
    ++lit;
    out[op++] = b;
    if (lit == kMaxLit) {                                                // a full run: close it, reserve the next control byte
      out[op - lit - 1] = (uint8_t)(lit - 1);
      lit = 0;
      ++op;
    }
    return true;
  };
  while (ip + 2 < in_len) {
    const uint32_t h = hash3(in + ip);
    const uint32_t ref = htab[h];
    htab[h] = (uint32_t)ip;
    if (ref != 0xffffffffu && ip - ref <= kMaxOff && in[ref] == in[ip] && in[ref + 1] == in[ip + 1] && in[ref + 2] == in[ip + 2]) {
      size_t len = 3;
      const size_t maxlen = in_len - ip < kMaxRef ? in_len - ip : kMaxRef;
      while (len < maxlen && in[ref + len] == in[ip + len]) ++len;
      if (op + 3 + 1 >= out_len) return 0;                               // the reference (3 bytes at most) + the next control byte
      end_run();
      const size_t off = ip - ref - 1, l = len - 2;                      // encoded: offset - 1, length - 2
      if (l < 7) {
        out[op++] = (uint8_t)((off >> 8) + (l << 5));
      } else {
        out[op++] = (uint8_t)((off >> 8) + (7u << 5));
        out[op++] = (uint8_t)(l - 7);
      }
      out[op++] = (uint8_t)(off & 0xff);
      lit = 0;
      ++op;                                                              // control byte of the next literal run
      // the positions the match covers enter the table too (so that runs keep finding recent references)
      const size_t stop = ip + len;
      for (++ip; ip < stop; ++ip)
        if (ip + 2 < in_len) htab[hash3(in + ip)] = (uint32_t)ip;
    } else if (!put_literal(in[ip++])) {
      return 0;
    }
  }
  while (ip < in_len)
    if (!put_literal(in[ip++])) return 0;
  end_run();
  return op <= out_len ? op : 0;
}

// `*full` (nullable): the OUTPUT buffer was too small -- the one failure a caller may answer by growing it; every other return of 0 is a
// malformed stream (truncated literal run / back reference, a reference before the start of the output)
static size_t lzf_decode(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_len, bool* full) {
  if (full) *full = false;
  if (!in || !out) return 0;
  size_t ip = 0, op = 0;
  while (ip < in_len) {
    const unsigned ctrl = in[ip++];
    if (ctrl < 32) {
      const size_t n = ctrl + 1;
      if (ip + n > in_len) return 0;
      if (op + n > out_len) { if (full) *full = true; return 0; }
      memcpy(out + op, in + ip, n);
      ip += n;
      op += n;
    } else {
      size_t len = ctrl >> 5;
      if (ip >= in_len) return 0;
      if (len == 7) {
        len += in[ip++];
        if (ip >= in_len) return 0;
      }
      const size_t off = ((size_t)(ctrl & 0x1f) << 8) + in[ip++] + 1;
      len += 2;
      if (off > op) return 0;
      if (op + len > out_len) { if (full) *full = true; return 0; }
      const uint8_t* src = out + op - off;
      for (size_t i = 0; i < len; ++i) out[op + i] = src[i];              // byte by byte: the ranges may overlap
      op += len;
    }
  }
  return op;
}
extern "C" size_t sgr_lzf_decompress(const void* in_, size_t in_len, void* out_, size_t out_len) {
  return lzf_decode(static_cast<const uint8_t*>(in_), in_len, static_cast<uint8_t*>(out_), out_len, nullptr);
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// libhdf5, by dlopen
// ---------------------------------------------------------------------------------------------------------------------------------------
namespace {
typedef int64_t hid_t;
typedef int herr_t;
typedef int htri_t;
typedef unsigned long long hsize_t;
typedef int H5Z_filter_t;
typedef size_t (*H5Z_func_t)(unsigned flags, size_t cd_nelmts, const unsigned cd_values[], size_t nbytes, size_t* buf_size, void** buf);
struct H5Z_class2_t {      // H5Zpublic.h, H5Z_CLASS_T_VERS 1
  int version;
  H5Z_filter_t id;
  unsigned encoder_present, decoder_present;
  const char* name;
  htri_t (*can_apply)(hid_t, hid_t, hid_t);
  herr_t (*set_local)(hid_t, hid_t, hid_t);
  H5Z_func_t filter;
};
constexpr unsigned H5F_ACC_RDONLY = 0x0000u, H5F_ACC_TRUNC = 0x0002u;
constexpr hid_t H5P_DEFAULT = 0, H5S_ALL = 0;
constexpr unsigned H5Z_FLAG_OPTIONAL = 0x0001, H5Z_FLAG_REVERSE = 0x0100;
constexpr H5Z_filter_t kLzfFilter = 32000;                                // h5py's registered filter id (H5PY_FILTER_LZF)
constexpr unsigned kLzfRevision = 4, kLzfVersion = 0x0105;                // H5PY_FILTER_LZF_VERSION, liblzf's LZF_VERSION: cd_values[0], [1]
enum { H5T_FLOAT = 1 };

struct Hdf5 {
  void* lib = nullptr;
  std::string tried;
  unsigned ver[3] = {0, 0, 0};
  herr_t (*H5open)();
  herr_t (*H5get_libversion)(unsigned*, unsigned*, unsigned*);
  herr_t (*H5Eset_auto2)(hid_t, void*, void*);
  hid_t (*H5Fcreate)(const char*, unsigned, hid_t, hid_t);
  hid_t (*H5Fopen)(const char*, unsigned, hid_t);
  herr_t (*H5Fclose)(hid_t);
  hid_t (*H5Screate_simple)(int, const hsize_t*, const hsize_t*);
  herr_t (*H5Sclose)(hid_t);
  int (*H5Sget_simple_extent_ndims)(hid_t);
  int (*H5Sget_simple_extent_dims)(hid_t, hsize_t*, hsize_t*);
  hid_t (*H5Pcreate)(hid_t);
  herr_t (*H5Pclose)(hid_t);
  herr_t (*H5Pset_chunk)(hid_t, int, const hsize_t*);
  int (*H5Pget_chunk)(hid_t, int, hsize_t*);
  herr_t (*H5Pset_filter)(hid_t, H5Z_filter_t, unsigned, size_t, const unsigned*);
  int (*H5Pget_nfilters)(hid_t);
  H5Z_filter_t (*H5Pget_filter2)(hid_t, unsigned, unsigned*, size_t*, unsigned*, size_t, char*, unsigned*);
  int (*H5Pget_layout)(hid_t);
  hid_t (*H5Dcreate2)(hid_t, const char*, hid_t, hid_t, hid_t, hid_t, hid_t);
  hid_t (*H5Dopen2)(hid_t, const char*, hid_t);
  herr_t (*H5Dclose)(hid_t);
  herr_t (*H5Dwrite)(hid_t, hid_t, hid_t, hid_t, hid_t, const void*);
  herr_t (*H5Dread)(hid_t, hid_t, hid_t, hid_t, hid_t, void*);
  hid_t (*H5Dget_space)(hid_t);
  hid_t (*H5Dget_type)(hid_t);
  hid_t (*H5Dget_create_plist)(hid_t);
  int (*H5Tget_class)(hid_t);
  size_t (*H5Tget_size)(hid_t);
  herr_t (*H5Tclose)(hid_t);
  herr_t (*H5Zregister)(const void*);
  htri_t (*H5Zfilter_avail)(H5Z_filter_t);
  hid_t native_float = -1, ieee_f32le = -1, dataset_create = -1;          // H5T_NATIVE_FLOAT, H5T_IEEE_F32LE, H5P_DATASET_CREATE (globals)
};
Hdf5 g;
std::once_flag g_once;
bool g_ok = false;
std::mutex g_mutex;                    // the conda build of libhdf5 is not thread-safe
thread_local std::string t_err;

int fail(int code, const std::string& msg) {
  t_err = msg;
  return code;
}

// h5py's filter (lzf_filter.c) restated: compress into a buffer of the chunk's size -- no gain => return 0, and with H5Z_FLAG_OPTIONAL HDF5
// stores the chunk raw and says so in its filter mask --, decompress into cd_values[2] bytes (the chunk size, set at creation)
size_t lzf_filter(unsigned flags, size_t cd_nelmts, const unsigned cd_values[], size_t nbytes, size_t* buf_size, void** buf) {
  if (!(flags & H5Z_FLAG_REVERSE)) {
    void* out = malloc(nbytes ? nbytes : 1);
    if (!out) return 0;
    const size_t n = sgr_lzf_compress(*buf, nbytes, out, nbytes);
    if (n == 0) {
      free(out);
      return 0;
    }
    free(*buf);
    *buf = out;
    *buf_size = nbytes;
    return n;
  }
  // The chunk size travels in cd_values[2] (h5py sets it at creation): ONE attempt into exactly that.  Only a file without it is decoded by
  // growing the buffer, and only while the decoder says "output full" -- a malformed stream fails at once instead of walking through 32
  // doublings of malloc (h5py grows on E2BIG alone), and the growth stops at 64x the compressed size's first guess.
  const bool known = cd_nelmts >= 3 && cd_values[2] != 0;
  size_t out_size = known ? cd_values[2] : *buf_size;
  for (int attempt = 0; attempt < (known ? 1 : 7); ++attempt) {
    void* out = malloc(out_size ? out_size : 1);
    if (!out) return 0;
    bool full = false;
    const size_t n = lzf_decode(static_cast<const uint8_t*>(*buf), nbytes, static_cast<uint8_t*>(out), out_size, &full);
    if (n) {
      free(*buf);
      *buf = out;
      *buf_size = out_size;
      return n;
    }
    free(out);
    if (!full) return 0;
    out_size = out_size ? out_size * 2 : 64;
  }
  return 0;
}
const H5Z_class2_t kLzfClass = {1, kLzfFilter, 1, 1, "lzf", nullptr, nullptr, lzf_filter};

bool validate(std::string& why);

void load_once() {
  std::vector<std::string> paths;
  if (const char* e = getenv("SGR_HDF5_LIB")) paths.push_back(e);
  for (const char* p : {"libhdf5.so", "libhdf5.so.103", "libhdf5.so.200", "libhdf5.so.310", "libhdf5_serial.so", "libhdf5_serial.so.103",
                        "/opt/conda/lib/libhdf5.so.103", "/opt/conda/lib/libhdf5.so", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so"})
    paths.push_back(p);
  // every candidate is opened AND validated (symbols, version >= 1.10, the *_g globals, the filter): one that fails is closed and the next
  // is tried -- an old unversioned libhdf5.so of a dev package in front of a good /opt/conda/lib/libhdf5.so.103 no longer ends the search
  // (ADVICE round 5).  The reason per path goes into `tried`.
  for (const auto& p : paths) {
    g.tried += (g.tried.empty() ? "" : ", ") + p;
    void* lib = dlopen(p.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!lib) { g.tried += " (not found)"; continue; }
    g.lib = lib;
    std::string why;
    if (validate(why)) return;
    g.tried += " (" + why + ")";
    dlclose(lib);
    g.lib = nullptr;
  }
}

// resolves the API in g.lib and registers the filter; false + reason when this library cannot serve
bool validate(std::string& why) {
  bool all = true;
#define SGR_SYM(name)                                                    \
  do {                                                                   \
    g.name = reinterpret_cast<decltype(g.name)>(dlsym(g.lib, #name));    \
    if (!g.name) { all = false; why += std::string(why.empty() ? "" : "; ") + "missing " #name; } \
  } while (0)
  SGR_SYM(H5open); SGR_SYM(H5get_libversion); SGR_SYM(H5Eset_auto2); SGR_SYM(H5Fcreate); SGR_SYM(H5Fopen); SGR_SYM(H5Fclose);
  SGR_SYM(H5Screate_simple); SGR_SYM(H5Sclose); SGR_SYM(H5Sget_simple_extent_ndims); SGR_SYM(H5Sget_simple_extent_dims);
  SGR_SYM(H5Pcreate); SGR_SYM(H5Pclose); SGR_SYM(H5Pset_chunk); SGR_SYM(H5Pget_chunk); SGR_SYM(H5Pset_filter); SGR_SYM(H5Pget_nfilters);
  SGR_SYM(H5Pget_filter2); SGR_SYM(H5Pget_layout); SGR_SYM(H5Dcreate2); SGR_SYM(H5Dopen2); SGR_SYM(H5Dclose); SGR_SYM(H5Dwrite); SGR_SYM(H5Dread);
  SGR_SYM(H5Dget_space); SGR_SYM(H5Dget_type); SGR_SYM(H5Dget_create_plist); SGR_SYM(H5Tget_class); SGR_SYM(H5Tget_size); SGR_SYM(H5Tclose);
  SGR_SYM(H5Zregister); SGR_SYM(H5Zfilter_avail);
#undef SGR_SYM
  if (!all) return false;
  if (g.H5open() < 0) { why = "H5open failed"; return false; }
  g.H5get_libversion(&g.ver[0], &g.ver[1], &g.ver[2]);
  if (g.ver[0] != 1 || g.ver[1] < 10) {      // 64-bit hid_t since 1.10
    why = "HDF5 " + std::to_string(g.ver[0]) + "." + std::to_string(g.ver[1]) + ", need >= 1.10";
    return false;
  }
  auto global = [&](const char* name) -> hid_t {
    const hid_t* p = reinterpret_cast<const hid_t*>(dlsym(g.lib, name));
    return p ? *p : -1;
  };
  g.native_float = global("H5T_NATIVE_FLOAT_g");
  g.ieee_f32le = global("H5T_IEEE_F32LE_g");
  g.dataset_create = global("H5P_CLS_DATASET_CREATE_ID_g");
  if (g.native_float < 0 || g.ieee_f32le < 0 || g.dataset_create < 0) {
    why = "HDF5 globals not found";
    return false;
  }
  g.H5Eset_auto2(0, nullptr, nullptr);      // no error stacks on stderr: failures are reported through return codes
  if (g.H5Zfilter_avail(kLzfFilter) <= 0 && g.H5Zregister(&kLzfClass) < 0) {
    why = "H5Zregister(lzf) failed";
    return false;
  }
  g_ok = true;
  return true;
}

bool ready() {
  std::call_once(g_once, load_once);
  if (!g_ok) t_err = "sgrender_h5: libhdf5 is not available (tried: " + g.tried + "); set SGR_HDF5_LIB to its path";
  return g_ok;
}

// h5py/_hl/filters.py: guess_chunk, for a fixed-shape dataset
void guess_chunk(int ndims, const hsize_t* shape, size_t typesize, hsize_t* chunks) {
  constexpr double kBase = 16 * 1024, kMin = 8 * 1024, kMax = 1024 * 1024;
  double dset = (double)typesize;
  for (int i = 0; i < ndims; ++i) {
    chunks[i] = shape[i] ? shape[i] : 1024;
    dset *= (double)chunks[i];
  }
  double target = kBase * exp2(log10(dset / (1024.0 * 1024.0)));
  if (target > kMax) target = kMax; else if (target < kMin) target = kMin;
  for (int idx = 0;; ++idx) {
    double bytes = (double)typesize, prod = 1.0;
    for (int i = 0; i < ndims; ++i) { bytes *= (double)chunks[i]; prod *= (double)chunks[i]; }
    if ((bytes < target || fabs(bytes - target) / target < 0.5) && bytes < kMax) break;
    if (prod == 1.0) break;
    hsize_t& c = chunks[idx % ndims];
    c = (hsize_t)ceil((double)c / 2.0);
  }
}
}  // namespace

extern "C" int sgr_h5_abi_version(void) { return SGR_H5_ABI_VERSION; }
extern "C" const char* sgr_h5_last_error(void) { return t_err.c_str(); }
extern "C" int sgr_h5_available(unsigned version[3]) {
  std::lock_guard<std::mutex> lock(g_mutex);
  const bool ok = ready();
  if (version) { version[0] = g.ver[0]; version[1] = g.ver[1]; version[2] = g.ver[2]; }
  return ok ? 1 : 0;
}

extern "C" int sgr_h5_write_f32(const char* path, const char* name, const float* data, int ndims, const unsigned long long* dims, int compression) {
  if (!path || !name || !data || !dims || ndims < 1 || ndims > SGR_H5_MAX_DIMS) return fail(SGR_H5_ERR_ARGUMENT, "sgr_h5_write_f32: NULL argument or rank outside 1..8");
  for (int i = 0; i < ndims; ++i)
    if (dims[i] == 0) return fail(SGR_H5_ERR_ARGUMENT, "sgr_h5_write_f32: zero-sized dimension");
  std::lock_guard<std::mutex> lock(g_mutex);
  if (!ready()) return SGR_H5_ERR_UNAVAILABLE;
  hsize_t shape[SGR_H5_MAX_DIMS], chunk[SGR_H5_MAX_DIMS];
  for (int i = 0; i < ndims; ++i) shape[i] = dims[i];
  guess_chunk(ndims, shape, sizeof(float), chunk);
  size_t chunk_bytes = sizeof(float);
  for (int i = 0; i < ndims; ++i) chunk_bytes *= (size_t)chunk[i];
  hid_t file = -1, space = -1, plist = -1, dset = -1;
  int rc = SGR_H5_OK;
  std::string why;
  do {
    file = g.H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    if (file < 0) { rc = SGR_H5_ERR_IO; why = "cannot create the file"; break; }
    space = g.H5Screate_simple(ndims, shape, nullptr);
    plist = g.H5Pcreate(g.dataset_create);
    if (space < 0 || plist < 0) { rc = SGR_H5_ERR_IO; why = "H5Screate_simple / H5Pcreate failed"; break; }
    if (g.H5Pset_chunk(plist, ndims, chunk) < 0) { rc = SGR_H5_ERR_IO; why = "H5Pset_chunk failed"; break; }
    if (compression) {
      const unsigned cd[3] = {kLzfRevision, kLzfVersion, (unsigned)chunk_bytes};      // what h5py's set_local callback stores
      if (g.H5Pset_filter(plist, kLzfFilter, H5Z_FLAG_OPTIONAL, 3, cd) < 0) { rc = SGR_H5_ERR_IO; why = "H5Pset_filter(lzf) failed"; break; }
    }
    dset = g.H5Dcreate2(file, name, g.ieee_f32le, space, H5P_DEFAULT, plist, H5P_DEFAULT);
    if (dset < 0) { rc = SGR_H5_ERR_IO; why = "H5Dcreate2 failed"; break; }
    if (g.H5Dwrite(dset, g.native_float, H5S_ALL, H5S_ALL, H5P_DEFAULT, data) < 0) { rc = SGR_H5_ERR_IO; why = "H5Dwrite failed"; break; }
  } while (false);
  if (dset >= 0 && g.H5Dclose(dset) < 0 && rc == SGR_H5_OK) { rc = SGR_H5_ERR_IO; why = "H5Dclose failed"; }
  if (plist >= 0) g.H5Pclose(plist);
  if (space >= 0) g.H5Sclose(space);
  if (file >= 0 && g.H5Fclose(file) < 0 && rc == SGR_H5_OK) { rc = SGR_H5_ERR_IO; why = "H5Fclose failed"; }
  return rc == SGR_H5_OK ? rc : fail(rc, std::string("sgr_h5_write_f32(") + path + "): " + why);
}

namespace {
struct Opened {
  hid_t file = -1, dset = -1;
  ~Opened() {
    if (dset >= 0) g.H5Dclose(dset);
    if (file >= 0) g.H5Fclose(file);
  }
};
int open_dataset(const char* fn, const char* path, const char* name, Opened& o) {
  o.file = g.H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
  if (o.file < 0) return fail(SGR_H5_ERR_IO, std::string(fn) + "(" + path + "): cannot open the file");
  o.dset = g.H5Dopen2(o.file, name, H5P_DEFAULT);
  if (o.dset < 0) return fail(SGR_H5_ERR_IO, std::string(fn) + "(" + path + "): no dataset '" + name + "'");
  return SGR_H5_OK;
}
int shape_of(const char* fn, const char* path, hid_t dset, int* ndims, hsize_t* dims) {
  const hid_t space = g.H5Dget_space(dset);
  if (space < 0) return fail(SGR_H5_ERR_IO, std::string(fn) + "(" + path + "): H5Dget_space failed");
  const int nd = g.H5Sget_simple_extent_ndims(space);
  int rc = SGR_H5_OK;
  if (nd < 0 || nd > SGR_H5_MAX_DIMS) rc = fail(SGR_H5_ERR_FORMAT, std::string(fn) + "(" + path + "): rank outside 0..8");
  else { g.H5Sget_simple_extent_dims(space, dims, nullptr); *ndims = nd; }
  g.H5Sclose(space);
  return rc;
}
}  // namespace

extern "C" int sgr_h5_shape(const char* path, const char* name, int* ndims, unsigned long long dims[SGR_H5_MAX_DIMS]) {
  if (!path || !name || !ndims || !dims) return fail(SGR_H5_ERR_ARGUMENT, "sgr_h5_shape: NULL argument");
  std::lock_guard<std::mutex> lock(g_mutex);
  if (!ready()) return SGR_H5_ERR_UNAVAILABLE;
  Opened o;
  if (int rc = open_dataset("sgr_h5_shape", path, name, o)) return rc;
  hsize_t d[SGR_H5_MAX_DIMS] = {0};
  if (int rc = shape_of("sgr_h5_shape", path, o.dset, ndims, d)) return rc;
  for (int i = 0; i < *ndims; ++i) dims[i] = d[i];
  return SGR_H5_OK;
}

extern "C" int sgr_h5_read_f32(const char* path, const char* name, float* out, unsigned long long capacity) {
  if (!path || !name || !out) return fail(SGR_H5_ERR_ARGUMENT, "sgr_h5_read_f32: NULL argument");
  std::lock_guard<std::mutex> lock(g_mutex);
  if (!ready()) return SGR_H5_ERR_UNAVAILABLE;
  Opened o;
  if (int rc = open_dataset("sgr_h5_read_f32", path, name, o)) return rc;
  int nd = 0;
  hsize_t d[SGR_H5_MAX_DIMS] = {0};
  if (int rc = shape_of("sgr_h5_read_f32", path, o.dset, &nd, d)) return rc;
  unsigned long long n = 1;
  for (int i = 0; i < nd; ++i) n *= d[i];
  if (n > capacity) return fail(SGR_H5_ERR_FORMAT, std::string("sgr_h5_read_f32(") + path + "): the dataset holds " + std::to_string(n) + " values, the buffer " + std::to_string(capacity));
  const hid_t type = g.H5Dget_type(o.dset);
  const bool f32 = type >= 0 && g.H5Tget_class(type) == H5T_FLOAT && g.H5Tget_size(type) == 4;
  if (type >= 0) g.H5Tclose(type);
  if (!f32) return fail(SGR_H5_ERR_FORMAT, std::string("sgr_h5_read_f32(") + path + "): dataset '" + name + "' is not float32");
  if (g.H5Dread(o.dset, g.native_float, H5S_ALL, H5S_ALL, H5P_DEFAULT, out) < 0)
    return fail(SGR_H5_ERR_IO, std::string("sgr_h5_read_f32(") + path + "): H5Dread failed (corrupt chunk or unknown filter)");
  return SGR_H5_OK;
}

extern "C" int sgr_h5_dataset_info(const char* path, const char* name, int* filter_id, int* ndims, unsigned long long chunk[SGR_H5_MAX_DIMS]) {
  if (!path || !name || !filter_id || !ndims || !chunk) return fail(SGR_H5_ERR_ARGUMENT, "sgr_h5_dataset_info: NULL argument");
  std::lock_guard<std::mutex> lock(g_mutex);
  if (!ready()) return SGR_H5_ERR_UNAVAILABLE;
  Opened o;
  if (int rc = open_dataset("sgr_h5_dataset_info", path, name, o)) return rc;
  const hid_t plist = g.H5Dget_create_plist(o.dset);
  if (plist < 0) return fail(SGR_H5_ERR_IO, "sgr_h5_dataset_info: H5Dget_create_plist failed");
  *filter_id = 0;
  *ndims = 0;
  if (g.H5Pget_nfilters(plist) > 0) {
    unsigned flags = 0, cfg = 0;
    size_t n = 0;
    *filter_id = (int)g.H5Pget_filter2(plist, 0, &flags, &n, nullptr, 0, nullptr, &cfg);
  }
  if (g.H5Pget_layout(plist) == 2 /* H5D_CHUNKED */) {
    hsize_t c[SGR_H5_MAX_DIMS] = {0};
    const int nd = g.H5Pget_chunk(plist, SGR_H5_MAX_DIMS, c);
    if (nd > 0 && nd <= SGR_H5_MAX_DIMS) {
      *ndims = nd;
      for (int i = 0; i < nd; ++i) chunk[i] = c[i];
    }
  }
  g.H5Pclose(plist);
  return SGR_H5_OK;
}
