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
echo "== trainlight example"; timeout 300 python examples/train_light_synthetic.py --batch 16 --steps 8 > gpurun_out/trainlight.txt 2>&1; tail -3 gpurun_out/trainlight.txt
echo "== bench"; timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.txt 2>&1; tail -1 gpurun_out/bench.txt | cut -c1-600
echo "== bench config5"; timeout 600 python bench.py --config 5 --steps 10 --warmup 2 > gpurun_out/bench_cfg5.txt 2>&1; tail -1 gpurun_out/bench_cfg5.txt | cut -c1-700
