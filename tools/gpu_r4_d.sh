#!/bin/bash
# round 4, session D: A/B of library variants in the bench loop and kbench (layer kernels), plus the host-overhead floor
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
LIB=inverserenderingofindoorscene_amd/libsgrender.so
for lib in $LIB inverserenderingofindoorscene_amd/variants/*.so $LIB; do
  echo "== $lib"
  timeout 120 ./tools/kbench $lib 16 20 2>&1 | grep -E "sgr_fused_fwd \(env|sgr_fused_bwd_sg \(g_env|sgr_fused_fwd_recon  |sgr_fused_bwd_recon  |sgr_sg_to_env_bwd" | tee -a gpurun_out/kbench_d.txt
  SGR_LIB=$PWD/$lib timeout 300 python bench.py --no-cpu-baseline --layer-only --reps 7 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print('   bench', d['value'], 'Mpix/s', d['ms_per_step'], 'ms  fwd', k['forward (sgr_fused_fwd)']['ms'], 'bwd', k['backward (sgr_fused_bwd_sg)']['ms'], d['config']['ms_per_step_repetitions'])" | tee -a gpurun_out/kbench_d.txt
done
echo "== parity subset on the variants"
for lib in inverserenderingofindoorscene_amd/variants/*.so; do SGR_LIB=$PWD/$lib timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -m gpu -x 2>&1 | tail -2; done
echo "== host overhead"; timeout 300 python tools/host_overhead.py 2>&1 | tail -10 | tee gpurun_out/host_overhead.txt
