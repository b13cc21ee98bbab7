#!/bin/bash
# round-4 session S/T: the objective in fewer launches; streaming passes of the render loss with their loads in flight together
set -u
mkdir -p gpurun_out
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_losses.py tests/test_gpu_objective.py tests/test_gpu_graph.py tests/test_gpu_sharded.py tests/test_gpu_wrapper.py tests/test_gpu_ops.py tests/test_gpu_fullsize.py -q -m gpu -x 2>&1 | tail -4 | cut -c1-300
for i in 1 2; do
echo "== bench"; timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print(d['value'], d['ms_per_step'], {k:c[k] for k in c if k.startswith('ms_per_step_') and k!='ms_per_step_repetitions'}); print({k:v for k,v in c['config3'].items() if k.startswith('ms_')})"
done | tee gpurun_out/r04t_bench.txt
echo "== kbench"; timeout 300 ./tools/kbench inverserenderingofindoorscene_amd/libsgrender.so 16 20 2>&1 | grep -E "render_loss|lsregress" | cut -c1-100
echo "== example"; timeout 300 python examples/train_light_synthetic.py --batch 16 --steps 43 2>&1 | tail -1
