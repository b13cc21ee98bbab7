#!/bin/bash
# end-of-round evidence: tests, bench (plain + torchrun world 1), rocprof kernel stats, PMC traffic, kbench
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
LIB=inverserenderingofindoorscene_amd/libsgrender.so
echo "== pytest gpu"; timeout 900 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu.txt 2>&1; tail -3 gpurun_out/pytest_gpu.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== kbench"; timeout 300 ./tools/kbench $LIB 16 20 > gpurun_out/kbench_fast.txt 2>&1; cat gpurun_out/kbench_fast.txt
echo "== bench"; t0=$SECONDS; timeout 600 python bench.py > gpurun_out/bench.txt 2>&1; echo "bench wall $((SECONDS-t0)) s" | tee gpurun_out/bench_time.txt; tail -1 gpurun_out/bench.txt
echo "== bench torchrun world=1 (RCCL init / barrier / all-reduce path)"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_torchrun1.txt 2>&1; tail -1 gpurun_out/bench_torchrun1.txt | cut -c1-300
echo "== rocprof"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.txt 2>&1; cd $GRAFT_REPO_ROOT
for f in $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); do cp $f gpurun_out/kernel_stats.csv; head -14 $f; done
echo "== pmc traffic"; bash tools/pmc_traffic.sh
echo "== trainlight example (fused objective + HIP heads | unfused + torch heads)"
timeout 300 python examples/train_light_synthetic.py --batch 16 --steps 23 2>&1 | tail -1 | tee gpurun_out/trainlight_fused.txt
timeout 300 python examples/train_light_synthetic.py --batch 16 --steps 23 --unfused --torch-heads 2>&1 | tail -1 | tee gpurun_out/trainlight_unfused.txt
echo "== config 5"; timeout 600 python bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_config5.txt 2>&1; tail -1 gpurun_out/bench_config5.txt | cut -c1-200
echo "== issue-cost microbenchmarks"; timeout 120 ./tools/ubench3 > gpurun_out/ubench3.txt 2>&1; timeout 120 ./tools/ubench4 > gpurun_out/ubench4.txt 2>&1; tail -3 gpurun_out/ubench4.txt
