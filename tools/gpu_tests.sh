#!/bin/bash
# the whole GPU test suite, with durations and the printed error tables
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
t0=$SECONDS
timeout 2400 python -m pytest tests -q -m gpu --durations=12 -s ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.txt 2>&1
echo "pytest wall $((SECONDS-t0)) s"
grep -E "worst|whole|WHOLE|vs the reference-made|seed [0-9]: worst|e_ref" gpurun_out/pytest_gpu.txt | cut -c1-1500
tail -40 gpurun_out/pytest_gpu.txt | cut -c1-600
