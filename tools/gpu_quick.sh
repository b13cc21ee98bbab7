#!/bin/bash
# quick session: parity tests + kernel bench (+ A/B of the split backward)
set -u
mkdir -p gpurun_out
LIB=inverserenderingofindoorscene_amd/libsgrender.so
echo "== pytest gpu"; timeout 900 python -m pytest tests -q -m gpu -x > gpurun_out/pytest_gpu.txt 2>&1; tail -5 gpurun_out/pytest_gpu.txt
echo "== kbench"; timeout 300 ./tools/kbench $LIB 16 20 > gpurun_out/kbench_fast.txt 2>&1; cat gpurun_out/kbench_fast.txt
echo "== kbench SGR_BWD_SPLIT=0"; SGR_BWD_SPLIT=0 timeout 300 ./tools/kbench $LIB 16 20 2>&1 | sed -n '6,8p'
echo "== bench"; timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.txt 2>&1; tail -1 gpurun_out/bench.txt | cut -c1-400
