/* C ABI of libsgrender_h5.so -- the cascade hand-off container, host side only.
 *
 * What it replaces in the reference (file:line relative to the reference checkout):
 *   utils.py:92-99            writeH5ToFile(imBatch, nameBatch): one HDF5 file per image, ONE dataset named "data" holding the
 *                             float32 array imBatch[n] ([ch, H, W]), h5py `compression='lzf'`
 *   dataLoader.py:277-283     loadH5(imName): np.array(h5py.File(imName, 'r').get('data')), None on any failure
 *   outputBRDFLight.py:246-301  the cascade-0 export: imenv_*_0.h5 [7*SGNum = 84, 120, 160] (the packed raw SG parameters,
 *                             wrapperBRDFLight.py:167-168,216-223), imdiffuse_*_0.h5 / imspecular_*_0.h5 [3, 120, 160], and the BRDF
 *                             maps imbaseColor / imnormal / imroughness / imdepth_*_0.h5; read back by dataLoader.py:97-105,160-...
 *
 * The files are real HDF5 written through libhdf5 (located at run time: $SGR_HDF5_LIB, then the usual sonames, then /opt/conda/lib -- the
 * image ships HDF5 1.10.6 there but no h5py for its main interpreter) with h5py's LZF filter (registered id 32000, cd_values
 * {revision 4, LZF_VERSION 0x0105, chunk bytes}, H5Z_FLAG_OPTIONAL -- an incompressible chunk is stored raw, as h5py does), h5py's chunk-shape
 * heuristic, and this library's own LZF coder (the liblzf stream format: literal runs of <= 32 bytes, back references of 3..264 bytes over
 * <= 8 KiB).  Files written here open in h5py (`compression == 'lzf'`), files written by h5py 3.3.0 read here bit for bit
 * (tests/test_h5_handoff.py, fixtures tests/golden/h5/ made by oracle/make_golden_h5.py under /opt/conda/bin/python3.9).
 *
 * Conventions: every function returns 0 on success, a negative SGR_H5_* code otherwise; sgr_h5_last_error() describes the last failure of
 * the calling thread.  No torch types, no device pointers: the tensors of the path live in HBM, the caller copies them to the host
 * (the reference does: `.data.cpu().numpy()`, utils.py:96).  Thread-safe as far as libhdf5 is (the conda build is not thread-safe: the
 * library serialises its calls with one mutex). */
#ifndef SGRENDER_H5_H
#define SGRENDER_H5_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGR_H5_ABI_VERSION 1
#define SGR_H5_MAX_DIMS 8

#define SGR_H5_OK 0
#define SGR_H5_ERR_ARGUMENT (-1)    /* NULL pointer, rank out of range, zero-sized dimension */
#define SGR_H5_ERR_UNAVAILABLE (-2) /* libhdf5 could not be loaded (message names the paths tried) */
#define SGR_H5_ERR_IO (-3)          /* file / dataset cannot be created, opened, read or written */
#define SGR_H5_ERR_FORMAT (-4)      /* dataset "data" is not float32, or rank / size does not fit the caller's buffer */

int sgr_h5_abi_version(void);
/* 1 when libhdf5 is loaded (loads it on first call), 0 otherwise; `version` (may be NULL) receives major, minor, release */
int sgr_h5_available(unsigned version[3]);
const char* sgr_h5_last_error(void);

/* utils.py:96-98 for one image: create / truncate `path`, dataset `name` (the reference: "data") of float32 with `dims[0..ndims)`,
 * chunked like h5py's auto-chunking; compression: 1 = lzf (the reference), 0 = none (contiguous chunks, no filter). */
int sgr_h5_write_f32(const char* path, const char* name, const float* data, int ndims, const unsigned long long* dims, int compression);

/* dataLoader.py:277-283 in two steps: the shape, then the values into the caller's buffer of `capacity` floats. */
int sgr_h5_shape(const char* path, const char* name, int* ndims, unsigned long long dims[SGR_H5_MAX_DIMS]);
int sgr_h5_read_f32(const char* path, const char* name, float* out, unsigned long long capacity);
/* filter pipeline of the dataset: `filter_id` = first filter (32000 for lzf; 0 when none), `chunk` = its chunk shape */
int sgr_h5_dataset_info(const char* path, const char* name, int* filter_id, int* ndims, unsigned long long chunk[SGR_H5_MAX_DIMS]);

/* The LZF coder itself (the liblzf stream format), exposed for the tests: returns the number of bytes produced, 0 when the output does
 * not fit (compress: "store raw") or the input is malformed (decompress). */
size_t sgr_lzf_compress(const void* in, size_t in_len, void* out, size_t out_len);
size_t sgr_lzf_decompress(const void* in, size_t in_len, void* out, size_t out_len);

#ifdef __cplusplus
}
#endif
#endif
