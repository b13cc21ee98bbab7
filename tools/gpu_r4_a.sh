#!/bin/bash
# round 4, session A: the C++ torch extension on the GPU (full test suite, smoke, host overhead) and an A/B of the objective-backward
# variants (kbench warm + the bench loop's objective / config-3 legs per library)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
LIB=inverserenderingofindoorscene_amd/libsgrender.so
echo "== pytest gpu"; t0=$SECONDS; timeout 1500 python -m pytest tests -q -m gpu --durations=8 -x > gpurun_out/pytest_gpu.txt 2>&1; echo "pytest wall $((SECONDS-t0)) s"; tail -25 gpurun_out/pytest_gpu.txt | cut -c1-300
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.txt | cut -c1-600
echo "== host overhead"; timeout 300 python tools/host_overhead.py 2>&1 | tail -3 | tee gpurun_out/host_overhead.txt
for lib in $LIB inverserenderingofindoorscene_amd/variants/*.so; do
  echo "== $lib"
  timeout 120 ./tools/kbench $lib 16 20 2>&1 | grep -E "sgr_fused_fwd_recon|sgr_fused_bwd_recon" | tee -a gpurun_out/kbench_variants.txt
  SGR_LIB=$PWD/$lib timeout 300 python bench.py --no-cpu-baseline --reps 5 2>&1 | tail -1 > gpurun_out/bench_$(basename $lib .so).json
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_$(basename $lib .so).json")); c=d["config"]
print("   bench", d["value"], "Mpix/s", d["ms_per_step"], "ms | with loss", c["ms_per_step_with_render_loss"], "| objective fused", c["ms_per_step_light_objective_fused"], "unfused", c["ms_per_step_light_objective_unfused"], "| config3", c["config3"])
PY
done
