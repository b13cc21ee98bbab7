#!/bin/bash
# Registers / scratch / LDS / instruction mix of the kernels of one translation unit, from the device ISA (no GPU needed):
#   tools/kernel_resources.sh sgr_fused_recon.hip [kernel-name filter] [extra hipcc flags...]
# (the same figures sit in the code object's notes: llvm-readelf --notes on the unbundled gfx950 object)
set -e
TU=$1; PAT=${2:-}; shift; shift || true
OUT=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -fno-math-errno -fno-slp-vectorize -Wno-unused-function -w \
  -S --cuda-device-only "$@" -o $OUT/k.s "$(dirname $0)/../inverserenderingofindoorscene_amd/csrc/$TU" 2>/dev/null
python "$(dirname $0)/isa_stats.py" $OUT/k.s "$PAT" | cut -c1-300
rm -rf $OUT
