#!/bin/bash
# quick session: every GPU test except the full-size sweeps, bench (plain + torchrun world 1)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu (without test_gpu_fullsize.py)"; timeout 900 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_fullsize.py --durations=8 > gpurun_out/pytest_gpu_quick.txt 2>&1; tail -16 gpurun_out/pytest_gpu_quick.txt
echo "== bench"; t0=$SECONDS; timeout 600 python bench.py > gpurun_out/bench.txt 2>&1; echo "bench wall $((SECONDS-t0)) s"; tail -1 gpurun_out/bench.txt | cut -c1-1800
echo "== bench torchrun world=1"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --layer-only > gpurun_out/bench_torchrun1.txt 2>&1; tail -1 gpurun_out/bench_torchrun1.txt | cut -c1-900
