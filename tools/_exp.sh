#!/bin/bash
export TMPDIR=/tmp
timeout 200 ./tools/kbench inverserenderingofindoorscene_amd/libsgrender.so 16 50 2>&1 | grep -E "premap 3|sgr_fused_fwd_recon  |sgr_fused_bwd_recon  "
timeout 900 python -m pytest tests/test_gpu_objective.py tests/test_gpu_heads.py -q -m gpu 2>&1 | tail -5
timeout 300 python bench.py --steps 50 --warmup 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value',d['value'],d['config'].get('config3'))
"
