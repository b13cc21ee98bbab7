#!/bin/bash
# A/B of compile-time variants: kernel timing (kbench, warm) and the bench loop (working set cycling through HBM)
set -u
mkdir -p gpurun_out
for lib in inverserenderingofindoorscene_amd/libsgrender.so inverserenderingofindoorscene_amd/variants/*.so; do   # variants: make -C .../csrc OBJDIR=/tmp/x OUT=../variants/libsgrender_x.so EXTRA=-DSGR_...=n
  echo "== $lib"
  timeout 120 ./tools/kbench $lib 16 20 2>&1 | grep -E "sgr_fused_fwd \(env|sgr_fused_bwd_sg \(g_env|sgr_fused_fwd_recon|sgr_fused_bwd_recon"
  SGR_LIB=$PWD/$lib timeout 200 python bench.py --no-cpu-baseline --layer-only --reps 5 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print('   bench', d['value'], 'Mpix/s', d['ms_per_step'], 'ms  fwd', k['forward (sgr_fused_fwd)']['ms'], 'bwd', k['backward (sgr_fused_bwd_sg)']['ms'], d['config']['ms_per_step_repetitions'])"
done
