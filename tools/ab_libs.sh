#!/bin/bash
# Same-box A/B of kernel libraries, alternating:  tools/ab_libs.sh <tag> lib1.so lib2.so ...   (the LAST one is the build under test:
# the parity suites run on it).  kbench (warm relaunches) x3, GPU parity tests, then the bench loop (driver flags) x2 per library.
set -u
TAG=$1; shift
LIBS="$@"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== kbench A/B"
for i in 1 2 3; do for lib in $LIBS; do
  timeout 300 ./tools/kbench $lib 16 20 2>&1 | grep -E "sgr_fused_fwd \(env|sgr_fused_bwd_sg \(g_env|sgr_light_objective_fwd|sgr_fused_bwd_recon  |sgr_fused_bwd_recon \(premap 3" | sed "s|^|$(basename $lib) |"
done; done | tee gpurun_out/${TAG}_kbench.txt
echo "== parity (last library = the in-tree build)"
timeout 1200 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_bench_launch.py 2>&1 | tail -6 | tee gpurun_out/${TAG}_pytest.txt
echo "== bench loop A/B (driver flags)"
for i in 1 2; do for lib in $LIBS; do
  SGR_LIB=$PWD/$lib timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-config5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$(basename $lib)', 'value', d['value'], 'ms', d['ms_per_step'], 'fwd_us', c['fwd_us'], 'bwd_us', c['bwd_us'], 'with loss', c['ms_with_loss'], 'obj', c.get('obj_ms'), 'obj_fwd_us', c.get('obj_fwd_us'), 'obj_bwd_us', c.get('obj_bwd_us'), 'cfg3', c.get('cfg3_ms'), 'cfg3_bwd_us', c.get('cfg3_bwd_us'), 'cfg4', c.get('cfg4_ms'))"
done; done | tee gpurun_out/${TAG}_bench.txt
