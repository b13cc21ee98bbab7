#!/bin/bash
# round-2 baseline session: parity tests, kbench, bench (batch 16 + 64), rocprof kernel stats
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
LIB=inverserenderingofindoorscene_amd/libsgrender.so
echo "== pytest gpu"; timeout 900 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu.txt 2>&1; tail -5 gpurun_out/pytest_gpu.txt
echo "== kbench"; timeout 300 ./tools/kbench $LIB 16 20 > gpurun_out/kbench.txt 2>&1; cat gpurun_out/kbench.txt
echo "== bench"; t0=$SECONDS; timeout 600 python bench.py > gpurun_out/bench.txt 2>&1; echo "bench wall $((SECONDS-t0)) s"; tail -1 gpurun_out/bench.txt
echo "== bench batch 64"; timeout 600 python bench.py --batch 64 --no-cpu-baseline > gpurun_out/bench_b64.txt 2>&1; tail -1 gpurun_out/bench_b64.txt | cut -c1-400
echo "== rocprof"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.txt 2>&1; cd $GRAFT_REPO_ROOT
for f in $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); do cp $f gpurun_out/kernel_stats.csv; head -8 $f | cut -c1-160; done
