#!/bin/bash
# round-4 session S: the objective's forward half in four launches (sgr_light_objective_fwd)
set -u
mkdir -p gpurun_out
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_losses.py tests/test_gpu_objective.py tests/test_gpu_graph.py tests/test_gpu_sharded.py tests/test_gpu_wrapper.py tests/test_gpu_ops.py tests/test_gpu_fullsize.py -q -m gpu -x 2>&1 | tail -4 | cut -c1-300
for i in 1 2; do
echo "== bench"; timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print(d['value'], d['ms_per_step'], {k:c[k] for k in c if k.startswith('ms_per_step_') and k!='ms_per_step_repetitions'}); print({k:v for k,v in c['config3'].items() if k.startswith('ms_')})"
done | tee gpurun_out/r04s_bench.txt
echo "== example"; timeout 300 python examples/train_light_synthetic.py --batch 16 --steps 43 2>&1 | tail -1
