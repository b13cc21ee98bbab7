#!/bin/bash
# round 4, session C: the packed half-wave forwardEnv kernel (vs round 1's scalar one: SGR_RENDER_ENV=scalar) and the ring-of-three
# ground-truth tiles of the objective backward (PMC traffic before / after), with the tests that cover them
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
LIB=inverserenderingofindoorscene_amd/libsgrender.so
echo "== pytest (parity, objective, fullsize, wrapper, losses)"; t0=$SECONDS
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_objective.py tests/test_gpu_fullsize.py tests/test_gpu_wrapper.py tests/test_gpu_losses.py tests/test_gpu_sharded.py -q -m gpu > gpurun_out/pytest_c.txt 2>&1
echo "pytest wall $((SECONDS-t0)) s"; tail -5 gpurun_out/pytest_c.txt | cut -c1-300
echo "== kbench"; timeout 300 ./tools/kbench $LIB 16 20 2>&1 | grep -E "render_env_fwd|fused_bwd_recon|fused_fwd_recon" | tee gpurun_out/kbench_c.txt
echo "== kbench, scalar forwardEnv"; SGR_RENDER_ENV=scalar timeout 300 ./tools/kbench $LIB 16 20 2>&1 | grep -E "render_env_fwd" | tee -a gpurun_out/kbench_c.txt
echo "== kbench cold"; KBENCH_COLD=1 timeout 300 ./tools/kbench $LIB 16 20 2>&1 | grep -E "render_env_fwd|fused_bwd_recon|fused_fwd_recon" | tee -a gpurun_out/kbench_c.txt
echo "== kbench cold, scalar forwardEnv"; KBENCH_COLD=1 SGR_RENDER_ENV=scalar timeout 300 ./tools/kbench $LIB 16 20 2>&1 | grep -E "render_env_fwd" | tee -a gpurun_out/kbench_c.txt
for lib in inverserenderingofindoorscene_amd/variants/*.so; do echo "== $lib"; timeout 300 ./tools/kbench $lib 16 20 2>&1 | grep -E "fused_bwd_recon" | tee -a gpurun_out/kbench_c.txt; done
echo "== pmc traffic config 2, objective"; bash tools/pmc_traffic.sh config2_batch16_objective --pmc-workload objective | grep -E "fwd_pk|sg_bwd_recon"
echo "== pmc traffic config 5, objective"; bash tools/pmc_traffic.sh config5_batch4_objective --config 5 --pmc-workload objective | grep -E "fwd_pk|sg_bwd_recon"
echo "== bench (objective legs)"; timeout 600 python bench.py --no-cpu-baseline --reps 5 2>&1 | tail -1 > gpurun_out/bench_c.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_c.json")); c=d["config"]
print("   bench", d["value"], "Mpix/s", d["ms_per_step"], "ms | with loss", c["ms_per_step_with_render_loss"], "| objective fused", c["ms_per_step_light_objective_fused"], "unfused", c["ms_per_step_light_objective_unfused"])
c3=c["config3"]; print("   config3", c3.get("ms_per_step_config3"), c3.get("ms_per_step_config3_standalone_heads"), c3.get("ms_per_step_config3_hipgraph"))
for k in ("kernels_config3","kernels_config3_standalone_heads"):
    for n,v in (c3.get(k) or {}).items():
        r=v["roofline"]; print("   ",k,n,r["kernel"][:60],r["avg_launch_ms"],"ms frac",r["frac"],"traffic",r["traffic"],r["limited_by"])
PY
