#!/bin/bash
# round-4 session Q: prologue wave priority per kernel family (the trace build, which has none, ran the objective's half-wave forward faster)
set -u
mkdir -p gpurun_out
B=inverserenderingofindoorscene_amd/libsgrender.so
V=inverserenderingofindoorscene_amd/variants/libsgrender_prio0.so
for lib in $B $V $B $V; do
  echo "== $lib"; timeout 300 ./tools/kbench $lib 16 20 2>&1 | grep -E "sgr_fused_fwd \(env written|sgr_fused_bwd_sg \(g_env|sgr_fused_fwd_recon  |sgr_fused_bwd_recon  |sgr_fused_fwd_recon \(premap 3|sgr_fused_bwd_recon \(premap 3|fwd \(env, premap 3" | cut -c1-60
done | tee gpurun_out/r04q_prio_kbench.txt
for lib in $B $V $B $V; do
  echo "== $lib"; SGR_LIB=$lib timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; k=d['kernels']; print(d['value'], d['ms_per_step'], 'fwd', k['forward (sgr_fused_fwd)']['ms'], 'bwd', k['backward (sgr_fused_bwd_sg)']['ms'], 'obj', c['ms_per_step_light_objective_fused'], 'fwd-only', c['ms_per_step_light_objective_forward_only'], {k:v for k,v in c['config3'].items() if k.startswith('ms_')})"
done | tee gpurun_out/r04q_prio_bench.txt
