#!/bin/bash
# quick session: parity tests + kernel bench
set -u
mkdir -p gpurun_out
LIB=inverserenderingofindoorscene_amd/libsgrender.so
echo "== pytest gpu"; timeout 900 python -m pytest tests -q -m gpu -x > gpurun_out/pytest_gpu.txt 2>&1; tail -5 gpurun_out/pytest_gpu.txt
echo "== kbench"; timeout 300 ./tools/kbench $LIB 16 20 > gpurun_out/kbench_fast.txt 2>&1; cat gpurun_out/kbench_fast.txt
for v in inverserenderingofindoorscene_amd/variants/*.so; do
  [ -f "$v" ] || continue
  echo "== kbench $v"; timeout 300 ./tools/kbench $v 16 20 > gpurun_out/kbench_$(basename $v .so).txt 2>&1; head -8 gpurun_out/kbench_$(basename $v .so).txt
done
