#!/bin/bash
# round-4 session M: HIP-graph replay tests, the forward-tail / loss-pooling overlap bound (review item 8)
set -u
mkdir -p gpurun_out
echo "== graph tests"; timeout 600 python -m pytest tests/test_gpu_graph.py -q -m gpu 2>&1 | tail -15 | cut -c1-300 | tee gpurun_out/r04m_graph_tests.txt
echo "== tail overlap"; timeout 300 python tools/tail_overlap.py 2>&1 | tail -8 | tee gpurun_out/r04m_tail_overlap.txt
