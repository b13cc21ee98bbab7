#!/bin/bash
# Trace build for tools/wavetrace (no GPU needed): a copy of the kernel sources (inverserenderingofindoorscene_amd/csrc_trace/, git-ignored) with
# trace points added to the objective's backward and a setter exported from sgr_fused_recon.hip, compiled with -DSGR_TRACE into
# inverserenderingofindoorscene_amd/variants/libsgrender_trace.so.  The tracked sources are not touched (their hash stamps the counter records).
set -eu
cd "$(dirname "$0")/../inverserenderingofindoorscene_amd"
rm -rf csrc_trace && mkdir csrc_trace
cp csrc/*.hip csrc/*.inl csrc/*.h csrc/Makefile csrc_trace/
python3 - <<'PY'
p = 'csrc_trace/sgr_fused_recon.hip'
s = open(p).read()
anchor = '  static_assert(NG == 2 || NG == 4, "two or four lane groups per pixel");\n'
assert anchor in s
s = s.replace(anchor, anchor + '  SGR_TRACE_BEGIN\n', 1)
i = s.index('void sg_bwd_recon_pk_kernel')
j = s.index('  auto row_loop = [&](auto ortho_c) {\n', i)
s = s[:j] + '  SGR_TRACE_MARK\n' + s[j:]
j = s.index('// x *= s / applied, skipped entirely when')
k = s.rindex('}\n', 0, j)
s = s[:k] + '  SGR_TRACE_END\n' + s[k:]
s += '''
#ifdef SGR_TRACE
extern "C" int sgr_debug_trace_recon(void* device_buffer) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(sgr::g_trace), &device_buffer, sizeof(void*));
}
#endif
'''
open(p, 'w').write(s)
PY
mkdir -p variants
make -C csrc_trace -j8 OBJDIR=build OUT=../variants/libsgrender_trace.so EXTRA=-DSGR_TRACE > /dev/null
ls -la variants/libsgrender_trace.so
