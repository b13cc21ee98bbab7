#!/bin/bash
# round-3 session E: full GPU tests, bench (full line with config 3 + objective legs), kbench, rocprof of the bench loop,
# PMC traffic + SQ counters for config 2 and config 5
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
LIB=inverserenderingofindoorscene_amd/libsgrender.so
echo "== pytest gpu"; t0=$SECONDS; timeout 1500 python -m pytest tests -q -m gpu --durations=6 -s > gpurun_out/pytest_gpu.txt 2>&1; echo "pytest wall $((SECONDS-t0)) s"; tail -12 gpurun_out/pytest_gpu.txt | cut -c1-300
echo "== kbench"; timeout 300 ./tools/kbench $LIB 16 20 > gpurun_out/kbench.txt 2>&1; cat gpurun_out/kbench.txt
echo "== bench"; t0=$SECONDS; timeout 900 python bench.py > gpurun_out/bench.txt 2>&1; echo "bench wall $((SECONDS-t0)) s"; tail -1 gpurun_out/bench.txt | cut -c1-3000
echo "== bench, four-launch loss"; SGR_LOSS_FUSED=0 timeout 300 python bench.py --layer-only --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('four-launch loss: layer', d['ms_per_step'], 'with loss', d['config']['ms_per_step_with_render_loss'])"
echo "== rocprof"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 20 --reps 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.txt 2>&1; cd $GRAFT_REPO_ROOT
for f in $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); do cp $f gpurun_out/kernel_stats.csv; head -30 $f | cut -c1-220; done
find gpurun_out/prof -name "*kernel_trace.csv" -size +1M -delete
echo "== pmc traffic config 2"; bash tools/pmc_traffic.sh config2_batch16_env | grep -E "fwd_pk|sg_bwd_pk"
echo "== pmc sq config 2"; bash tools/pmc_sq.sh config2_batch16_env | grep -E "fwd_pk|sg_bwd_pk"
echo "== pmc traffic config 5"; bash tools/pmc_traffic.sh config5_batch4_env --config 5 | grep -E "fwd_pk|sg_bwd_pk"
echo "== pmc sq config 5"; bash tools/pmc_sq.sh config5_batch4_env --config 5 | grep -E "fwd_pk|sg_bwd_pk"
