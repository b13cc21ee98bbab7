#!/bin/bash
# objective kernels: parity tests, kernel timing packed vs scalar, bench, training example
set -u
mkdir -p gpurun_out
LIB=inverserenderingofindoorscene_amd/libsgrender.so
timeout 900 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_fullsize.py > gpurun_out/pytest_gpu_quick.txt 2>&1; tail -6 gpurun_out/pytest_gpu_quick.txt
echo "== kbench packed"; timeout 120 ./tools/kbench $LIB 16 20 2>&1 | grep -E "fused_fwd_recon|fused_bwd_recon"
echo "== kbench scalar"; SGR_F1_MODE=scalar SGR_B1_MODE=scalar timeout 120 ./tools/kbench $LIB 16 20 2>&1 | grep -E "fused_fwd_recon|fused_bwd_recon"
echo "== bench"; timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print(d['value'], d['ms_per_step'], 'obj fused', c['ms_per_step_light_objective_fused'], 'unfused', c['ms_per_step_light_objective_unfused'], 'graph', c['ms_per_step_light_objective_fused_hipgraph_replay'])"
echo "== trainlight"; timeout 300 python examples/train_light_synthetic.py --batch 16 --steps 23 2>&1 | tail -1
SGR_F1_MODE=scalar SGR_B1_MODE=scalar timeout 300 python examples/train_light_synthetic.py --batch 16 --steps 23 2>&1 | tail -1
