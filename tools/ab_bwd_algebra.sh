#!/bin/bash
# A/B of the round-6 backward algebra (dL/dlam = a . S - w . q in the epilogue): previous library vs this build, same box, alternating.
#   kbench (warm relaunches), parity suites on the new build, then the bench loop (driver flags) with each library.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
NEW=inverserenderingofindoorscene_amd/libsgrender.so
OLD=inverserenderingofindoorscene_amd/variants/libsgrender_prev.so
echo "== kbench A/B"
DEF=inverserenderingofindoorscene_amd/variants/libsgrender_defer.so
for i in 1 2 3; do for lib in $OLD $NEW $DEF; do
  KBENCH_ONLY=bwd timeout 200 ./tools/kbench $lib 16 20 2>&1 | grep -E "sgr_fused_bwd_sg \(g_env|sgr_fused_bwd_recon" | sed "s|^|$(basename $lib) |"
done; done | tee gpurun_out/ab_bwd_algebra_kbench.txt
echo "== parity (new build)"
timeout 1200 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_bench_launch.py 2>&1 | tail -15 | tee gpurun_out/ab_bwd_algebra_pytest.txt
echo "== bench loop A/B (driver flags)"
for i in 1 2; do for lib in $OLD $NEW $DEF; do
  SGR_LIB=$PWD/$lib timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-config5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$(basename $lib)', 'value', d['value'], 'ms', d['ms_per_step'], 'fwd_us', c['fwd_us'], 'bwd_us', c['bwd_us'], 'with loss', c['ms_with_loss'], 'obj', c.get('obj_ms'), 'obj_bwd_us', c.get('obj_bwd_us'), 'cfg3', c.get('cfg3_ms'), 'cfg3_bwd_us', c.get('cfg3_bwd_us'), 'cfg4', c.get('cfg4_ms'))"
done; done | tee gpurun_out/ab_bwd_algebra_bench.txt
