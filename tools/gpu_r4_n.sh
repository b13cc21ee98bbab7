#!/bin/bash
# round-4 session N: machine-scheduler strategies (-mllvm -amdgpu-sched-strategy=...) A/B through $SGR_LIB, one box
set -u
mkdir -p gpurun_out
B=inverserenderingofindoorscene_amd/libsgrender.so
V=inverserenderingofindoorscene_amd/variants
KEYS="sgr_fused_fwd (env written)|sgr_fused_bwd_sg \(g_env|sgr_fused_fwd_recon  |sgr_fused_bwd_recon  |sgr_fused_fwd_recon \(premap 3|sgr_fused_bwd_recon \(premap 3|sgr_render_env_fwd|sgr_sg_to_env_fwd|sgr_render_bwd_brdf \(env"
for lib in $B $V/libsgrender_max-ilp.so $V/libsgrender_max-memory-clause.so $B $V/libsgrender_max-ilp.so; do
  echo "== $lib"; timeout 300 ./tools/kbench $lib 16 20 2>&1 | grep -E "$KEYS" | cut -c1-60
done | tee gpurun_out/r04n_sched_kbench.txt
for lib in $B $V/libsgrender_max-ilp.so $V/libsgrender_max-memory-clause.so $B; do
  echo "== $lib"; SGR_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --layer-only 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print(d['value'], 'Mpix/s', d['ms_per_step'], 'ms/step  fwd', k['forward (sgr_fused_fwd)']['ms'], 'bwd', k['backward (sgr_fused_bwd_sg)']['ms'])"
done | tee gpurun_out/r04n_sched_bench.txt
