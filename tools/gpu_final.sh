#!/bin/bash
# end-of-round evidence: tests, smoke, kbench, bench (plain + torchrun), rocprof kernel stats, PMC traffic (config 2 and 5),
# SQ counters, training example, config 5, batch sweep, clocks under load.  Everything lands in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
LIB=inverserenderingofindoorscene_amd/libsgrender.so
echo "== pytest gpu"; t0=$SECONDS; timeout 1200 python -m pytest tests -q -m gpu --durations=6 > gpurun_out/pytest_gpu.txt 2>&1; echo "pytest wall $((SECONDS-t0)) s"; tail -10 gpurun_out/pytest_gpu.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.txt
echo "== kbench"; timeout 300 ./tools/kbench $LIB 16 20 > gpurun_out/kbench.txt 2>&1; cat gpurun_out/kbench.txt
echo "== bench"; t0=$SECONDS; timeout 600 python bench.py > gpurun_out/bench.txt 2>&1; echo "bench wall $((SECONDS-t0)) s" | tee gpurun_out/bench_time.txt; tail -1 gpurun_out/bench.txt | cut -c1-400
echo "== clocks under load"; (timeout 60 python bench.py --layer-only --no-cpu-baseline --steps 2000 --reps 3 > gpurun_out/bench_long.txt 2>&1 &) ; sleep 14; for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -4; sleep 2; done | tee gpurun_out/clocks_under_load.txt; wait; sleep 12; tail -1 gpurun_out/bench_long.txt | cut -c1-200
echo "== bench torchrun world=1 (RCCL init / barrier / all-reduce path)"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --no-cpu-baseline --layer-only > gpurun_out/bench_torchrun1.txt 2>&1; tail -1 gpurun_out/bench_torchrun1.txt | cut -c1-300
echo "== rocprof"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 20 --reps 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.txt 2>&1; cd $GRAFT_REPO_ROOT
for f in $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); do cp $f gpurun_out/kernel_stats.csv; head -14 $f | cut -c1-200; done
find gpurun_out/prof -name "*kernel_trace.csv" -size +1M -delete
echo "== pmc traffic config 2"; bash tools/pmc_traffic.sh config2_batch16_env
echo "== pmc traffic config 5"; bash tools/pmc_traffic.sh config5_batch4_env --config 5
echo "== sq counters"; bash tools/pmc_sq.sh 2>&1 | grep -A1 -E "fwd_pk_kernel<12, 2, true, true>|sg_bwd_pk_kernel<2, true, true>" | cut -c1-400
echo "== trainlight example (fused objective + HIP heads | unfused + torch heads)"
timeout 300 python examples/train_light_synthetic.py --batch 16 --steps 23 2>&1 | tail -1 | tee gpurun_out/trainlight_fused.txt
timeout 300 python examples/train_light_synthetic.py --batch 16 --steps 23 --unfused --torch-heads 2>&1 | tail -1 | tee gpurun_out/trainlight_unfused.txt
echo "== config 5"; timeout 600 python bench.py --config 5 --no-cpu-baseline --layer-only > gpurun_out/bench_config5.txt 2>&1; tail -1 gpurun_out/bench_config5.txt | cut -c1-300
echo "== batch sweep"; for b in 5 8 16 32 64; do timeout 300 python bench.py --batch $b --no-cpu-baseline --layer-only 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print('batch $b', d['value'], 'Mpix/s', d['ms_per_step'], 'ms/step  fwd', k['forward (sgr_fused_fwd)']['ms'], 'bwd', k['backward (sgr_fused_bwd_sg)']['ms'], 'with loss', d['config']['Mpix_per_s_with_render_loss'])"; done | tee gpurun_out/bench_batch_sweep.txt
