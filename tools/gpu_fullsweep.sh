#!/bin/bash
# whole-image fp64 oracle sweep of the full-size tests (6-7 minutes of host time) + the kernel-variant tests
set -u
mkdir -p gpurun_out
t0=$SECONDS; SGR_FULL_SWEEP=1 timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -s -k "config2 or config5 or one_image" > gpurun_out/fullsize_sweep.txt 2>&1; echo "wall $((SECONDS-t0)) s" >> gpurun_out/fullsize_sweep.txt; tail -6 gpurun_out/fullsize_sweep.txt | cut -c1-600
timeout 600 python -m pytest tests/test_gpu_variants.py -q -m gpu 2>&1 | tail -3
