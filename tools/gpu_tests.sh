#!/bin/bash
# the driver's GPU tier as it will run it: pytest -m gpu (windowed full-size oracles), then smoke()
set -u
mkdir -p gpurun_out
t0=$SECONDS; timeout 1200 python -m pytest tests -q -m gpu --durations=10 > gpurun_out/pytest_gpu.txt 2>&1; echo "pytest wall $((SECONDS-t0)) s"; tail -18 gpurun_out/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
