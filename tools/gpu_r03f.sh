#!/bin/bash
# round-3 session F: fused objective for 24 lobes / 16x32 (tests), register-fence A/B of the objective backward, config-5 objective legs
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
V=inverserenderingofindoorscene_amd/variants
echo "== objective tests"; timeout 1200 python -m pytest tests/test_gpu_objective.py tests/test_gpu_fullsize.py -q -m gpu -x -s 2>&1 | tail -15 | cut -c1-400
echo "== kbench recon: fences 2 (default) / 1 / 0"
for spec in "f2 inverserenderingofindoorscene_amd/libsgrender.so" "f1 $V/libsgrender_rf1.so" "f0 $V/libsgrender_rf0.so"; do set -- $spec; echo "-- $1 warm"; timeout 200 ./tools/kbench $2 16 20 2>&1 | grep -E "fused_bwd_recon|fused_fwd_recon "; echo "-- $1 cold"; KBENCH_COLD=1 timeout 200 ./tools/kbench $2 16 10 2>&1 | grep -E "fused_bwd_recon "; done
echo "== bench objective legs: fences 2 / 1 / 0"
for spec in "f2 inverserenderingofindoorscene_amd/libsgrender.so" "f1 $V/libsgrender_rf1.so" "f0 $V/libsgrender_rf0.so"; do set -- $spec; SGR_LIB=$2 timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print('$1 objective fused', c['ms_per_step_light_objective_fused'], 'unfused', c['ms_per_step_light_objective_unfused'], 'config3', c['config3'].get('ms_per_step_config3'), 'layer', d['ms_per_step'])"; done
echo "== config 5 bench (objective legs)"; timeout 900 python bench.py --config 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print('config5 layer', d['value'], 'Mpix/s', d['ms_per_step'], 'ms; objective fused', c['ms_per_step_light_objective_fused'], 'unfused', c['ms_per_step_light_objective_unfused'], 'config3', c['config3'])"
