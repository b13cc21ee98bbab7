#!/bin/bash
# One GPU-box session: parity tests, kernel-level bench (variants), bench.py, rocprof summary -> gpurun_out/
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
LIB=inverserenderingofindoorscene_amd/libsgrender.so
echo "== pytest gpu"; timeout 900 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu.txt 2>&1; tail -25 gpurun_out/pytest_gpu.txt
echo "== kbench fast"; timeout 300 ./tools/kbench $LIB 16 20 > gpurun_out/kbench_fast.txt 2>&1; cat gpurun_out/kbench_fast.txt
echo "== kbench fast TJ32"; SGR_FWD_TJ=32 timeout 300 ./tools/kbench $LIB 16 20 > gpurun_out/kbench_fast_tj32.txt 2>&1; head -4 gpurun_out/kbench_fast_tj32.txt
for v in inverserenderingofindoorscene_amd/variants/*.so; do
  [ -f "$v" ] || continue
  echo "== kbench $v"; timeout 300 ./tools/kbench $v 16 20 > gpurun_out/kbench_$(basename $v .so).txt 2>&1; head -8 gpurun_out/kbench_$(basename $v .so).txt
done
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.txt 2>&1; tail -2 gpurun_out/smoke.txt
echo "== bench"; timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.txt 2>&1; tail -2 gpurun_out/bench.txt
echo "== rocprof"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.txt 2>&1; cd $GRAFT_REPO_ROOT
for f in $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); do cp $f gpurun_out/kernel_stats.csv; head -12 $f; done
echo "== trainlight example"; timeout 300 python examples/train_light_synthetic.py --batch 16 --steps 8 > gpurun_out/trainlight.txt 2>&1; tail -4 gpurun_out/trainlight.txt
