#!/bin/bash
# round-3 session B: what bounds the forward in the bench loop?  tan hand-off on/off, env flush width (64- vs 128-byte segments),
# mixed grid on/off -- all through bench.py (the loop is the truth: kbench relaunches on warm buffers)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
V=inverserenderingofindoorscene_amd/variants
run() { name=$1; shift; env "$@" timeout 300 python bench.py --layer-only --no-cpu-baseline > gpurun_out/bench_$name.txt 2>&1; tail -1 gpurun_out/bench_$name.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print('$name', d['value'], 'Mpix/s', d['ms_per_step'], 'ms/step  fwd', k['forward (sgr_fused_fwd)']['ms'], 'bwd', k['backward (sgr_fused_bwd_sg)']['ms'], 'with loss', d['config']['ms_per_step_with_render_loss'], d['config']['ms_per_step_repetitions'])" || tail -3 gpurun_out/bench_$name.txt; }
run default SGR_DUMMY=1
run nomixed SGR_FWD_MIXED=0
run notan SGR_TAN_HANDOFF=0
run notan_nomixed SGR_TAN_HANDOFF=0 SGR_FWD_MIXED=0
run pkhalf2 SGR_FWD_MODE=pkhalf2
run pkhalf3 SGR_FWD_MODE=pkhalf3
run pkhalf2w SGR_FWD_MODE=pkhalf2w
run pkhalf3w SGR_FWD_MODE=pkhalf3w
run notan_pkhalf2w SGR_FWD_MODE=pkhalf2w SGR_TAN_HANDOFF=0
run notan_pkhalf3w SGR_FWD_MODE=pkhalf3w SGR_TAN_HANDOFF=0
run tj32 SGR_LIB=$V/libsgrender_tj32.so SGR_FWD_MIXED=0
run notan_tj32 SGR_LIB=$V/libsgrender_tj32.so SGR_FWD_MIXED=0 SGR_TAN_HANDOFF=0
run default_again SGR_DUMMY=1
echo "== batch 64, default / pkhalf2w / notan"
for spec in "b64_default SGR_DUMMY=1" "b64_pkhalf2w SGR_FWD_MODE=pkhalf2w" "b64_notan SGR_TAN_HANDOFF=0" "b64_notan_pkhalf2w SGR_TAN_HANDOFF=0 SGR_FWD_MODE=pkhalf2w"; do set -- $spec; name=$1; shift; env "$@" timeout 300 python bench.py --layer-only --no-cpu-baseline --batch 64 --steps 30 --warmup 60 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print('$name', d['value'], 'Mpix/s', d['ms_per_step'], 'ms/step  fwd', k['forward (sgr_fused_fwd)']['ms'], 'bwd', k['backward (sgr_fused_bwd_sg)']['ms'])"; done
echo "== objective legs: tan hand-off on / off"
for t in 1 0; do SGR_TAN_HANDOFF=$t timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print('tan=$t objective fused', c['ms_per_step_light_objective_fused'], 'unfused', c['ms_per_step_light_objective_unfused'], 'graph', c['ms_per_step_light_objective_fused_hipgraph_replay'], 'layer', d['ms_per_step'])"; done
echo "== pytest parity with pkhalf2w"; SGR_FWD_MODE=pkhalf2w timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -3
