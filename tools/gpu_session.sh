#!/bin/bash
# One GPU-box session: microbench, parity tests, bench, rocprof summaries -> gpurun_out/
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocm-smi" ; rocm-smi --showproductname 2>/dev/null | head -8
echo "== microbench"; timeout 120 ./tools/microbench > gpurun_out/microbench.txt 2>&1; tail -40 gpurun_out/microbench.txt
echo "== pytest gpu"; timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.txt 2>&1; tail -30 gpurun_out/pytest_gpu.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.txt 2>&1; tail -5 gpurun_out/smoke.txt
echo "== bench"; timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.txt 2>&1; tail -5 gpurun_out/bench.txt
echo "== bench noenv"; timeout 600 python bench.py --steps 20 --warmup 3 --no-env --no-cpu-baseline > gpurun_out/bench_noenv.txt 2>&1; tail -3 gpurun_out/bench_noenv.txt
echo "== rocprof"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.txt 2>&1; cd $GRAFT_REPO_ROOT
find gpurun_out/prof -name "*kernel_stats*" | head; for f in $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); do head -12 $f; done
