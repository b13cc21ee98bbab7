#!/bin/bash
# Builds the ablation variants of the kernel library (no GPU needed): inverserenderingofindoorscene_amd/variants/libsgrender_abl<bits>.so with
# -DSGR_ABLATE=<bits> (csrc/sgr_pk.inl: a data-movement component removed -- results wrong, timing = that component's cost).
#   tools/ablate.sh 1 2 8 16 ...      then on the GPU box: KBENCH_ONLY=... tools/kbench inverserenderingofindoorscene_amd/variants/libsgrender_abl8.so
set -eu
cd "$(dirname "$0")/../inverserenderingofindoorscene_amd/csrc"
mkdir -p ../variants
for b in "$@"; do
  make -j8 OBJDIR=build_v_abl$b OUT=../variants/libsgrender_abl$b.so EXTRA=-DSGR_ABLATE=$b > /dev/null 2>&1
  ls -la ../variants/libsgrender_abl$b.so
done
