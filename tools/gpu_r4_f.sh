#!/bin/bash
# round 4, session F: packed half-wave dL/dEnv kernel of forwardEnv (vs the generic table-driven one: SGR_GENV=generic) + the tests that cover it
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
LIB=inverserenderingofindoorscene_amd/libsgrender.so
echo "== pytest (parity, wrapper, ops, losses)"; timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wrapper.py tests/test_gpu_ops.py tests/test_gpu_losses.py tests/test_gpu_sharded.py -q -m gpu 2>&1 | tail -4 | cut -c1-300
for mode in "" generic; do for cold in 0 1; do echo "== kbench SGR_GENV=$mode cold=$cold"; SGR_GENV=$mode KBENCH_COLD=$cold timeout 300 ./tools/kbench $LIB 16 20 2>&1 | grep -E "render_env_bwd_env|render_env_fwd" | tee -a gpurun_out/kbench_f.txt; done; done
