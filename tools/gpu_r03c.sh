#!/bin/bash
# round-3 session C: forward dispatch by batch size (one pixel per lane vs half-wave with 128-byte env segments), spec-only
# exchange in the backward, GPU tests
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print('$1', d['value'], 'Mpix/s', d['ms_per_step'], 'ms/step  fwd', k['forward (sgr_fused_fwd)']['ms'], 'bwd', k['backward (sgr_fused_bwd_sg)']['ms'])"; }
for b in 4 8 16 32 64; do
  st=100; wu=300; if [ $b -ge 32 ]; then st=40; wu=80; fi
  for mode in pk pkhalf2w pkhalf3w pkhalf3; do
    SGR_FWD_MODE=$mode timeout 300 python bench.py --layer-only --no-cpu-baseline --batch $b --steps $st --warmup $wu 2>&1 | tail -1 | line "batch=$b $mode"
  done
done | tee gpurun_out/fwd_mode_sweep.txt
echo "== pytest gpu (all but fullsize)"; t0=$SECONDS; timeout 1200 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_fullsize.py --durations=5 > gpurun_out/pytest_gpu_quick.txt 2>&1; echo "pytest wall $((SECONDS-t0)) s"; tail -8 gpurun_out/pytest_gpu_quick.txt
echo "== pytest parity with pkhalf3w / pkhalf2w"; for m in pkhalf3w pkhalf2w; do SGR_FWD_MODE=$m timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -2; done
echo "== config 5 layer"; timeout 600 python bench.py --config 5 --no-cpu-baseline --layer-only 2>&1 | tail -1 | line "config5"
