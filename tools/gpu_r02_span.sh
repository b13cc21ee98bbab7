#!/bin/bash
# round-2 session: row-span kernels -- parity vs one-group-per-wave launches, kernel timing, bench A/B, wave trace
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
LIB=inverserenderingofindoorscene_amd/libsgrender.so
echo "== pytest span + sharded"; timeout 400 python -m pytest tests/test_gpu_span.py tests/test_gpu_sharded.py -x -q > gpurun_out/pytest_span.txt 2>&1; tail -15 gpurun_out/pytest_span.txt
echo "== kbench pair bn=16"; KBENCH_PAIR=1 timeout 120 ./tools/kbench $LIB 16 20 2>&1 | tee gpurun_out/kbench_pair16.txt
echo "== kbench pair bn=64"; KBENCH_PAIR=1 timeout 120 ./tools/kbench $LIB 64 10 2>&1 | tee gpurun_out/kbench_pair64.txt
for v in inverserenderingofindoorscene_amd/variants/libsgrender_nosync.so; do
  [ -f "$v" ] || continue
  echo "== kbench pair $v"; KBENCH_PAIR=1 timeout 120 ./tools/kbench $v 16 20 2>&1 | tee gpurun_out/kbench_pair16_$(basename $v .so).txt
done
echo "== bench span"; timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_span.txt 2>&1; tail -1 gpurun_out/bench_span.txt | cut -c1-260; tail -1 gpurun_out/bench_span.txt | grep -o '"kernels".*' | cut -c1-400
echo "== bench SGR_SPAN=0"; SGR_SPAN=0 timeout 300 python bench.py --no-cpu-baseline --layer-only > gpurun_out/bench_nospan.txt 2>&1; tail -1 gpurun_out/bench_nospan.txt | cut -c1-260; tail -1 gpurun_out/bench_nospan.txt | grep -o '"kernels".*' | cut -c1-400
echo "== bench span batch 64"; timeout 300 python bench.py --no-cpu-baseline --layer-only --batch 64 > gpurun_out/bench_span_b64.txt 2>&1; tail -1 gpurun_out/bench_span_b64.txt | cut -c1-260; tail -1 gpurun_out/bench_span_b64.txt | grep -o '"kernels".*' | cut -c1-400
echo "== wavetrace"; timeout 120 ./tools/wavetrace inverserenderingofindoorscene_amd/variants/libsgrender_trace.so 16 > gpurun_out/wavetrace.txt 2>gpurun_out/wavetrace.err; python tools/wavetrace_report.py gpurun_out/wavetrace.txt > gpurun_out/wavetrace_report.txt 2>&1; grep -A6 "span:" gpurun_out/wavetrace_report.txt | cut -c1-260; grep "^#" gpurun_out/wavetrace_report.txt
rm -f gpurun_out/wavetrace.txt
