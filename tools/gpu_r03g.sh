#!/bin/bash
set -u
export TMPDIR=/tmp
V=inverserenderingofindoorscene_amd/variants
for rep in 1 2; do
for spec in "f2 inverserenderingofindoorscene_amd/libsgrender.so" "f1 $V/libsgrender_bf1.so"; do set -- $spec; SGR_LIB=$2 timeout 300 python bench.py --no-cpu-baseline --layer-only 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print('$1 layer', d['ms_per_step'], 'fwd', k['forward (sgr_fused_fwd)']['ms'], 'bwd', k['backward (sgr_fused_bwd_sg)']['ms'])"; done; done
SGR_LIB=$V/libsgrender_bf1.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -2
for spec in "f2 inverserenderingofindoorscene_amd/libsgrender.so" "f1 $V/libsgrender_bf1.so"; do set -- $spec; SGR_LIB=$2 timeout 300 python bench.py --no-cpu-baseline --layer-only --config 5 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print('$1 config5 layer', d['ms_per_step'], 'fwd', k['forward (sgr_fused_fwd)']['ms'], 'bwd', k['backward (sgr_fused_bwd_sg)']['ms'])"; done
