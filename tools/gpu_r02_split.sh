#!/bin/bash
# round-2 session: tail-split kernels -- parity vs plain launches, kernel timing at several split counts, bench A/B, wave trace
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
LIB=inverserenderingofindoorscene_amd/libsgrender.so
echo "== pytest split + sharded"; timeout 400 python -m pytest tests/test_gpu_split.py tests/test_gpu_sharded.py -x -q > gpurun_out/pytest_split.txt 2>&1; tail -15 gpurun_out/pytest_split.txt
for S in auto 0 352 704 1056 1408 2048; do
  echo "== kbench pair bn=16 SGR_SPLIT=$S"
  if [ $S = auto ]; then KBENCH_PAIR=1 timeout 120 ./tools/kbench $LIB 16 20 2>&1 | grep -v "^#" | tee gpurun_out/kbench_pair16_$S.txt
  else SGR_SPLIT=$S KBENCH_PAIR=1 timeout 120 ./tools/kbench $LIB 16 20 2>&1 | grep -v "^#" | tee gpurun_out/kbench_pair16_$S.txt; fi
done
echo "== kbench pair bn=64"; KBENCH_PAIR=1 timeout 120 ./tools/kbench $LIB 64 10 2>&1 | tee gpurun_out/kbench_pair64.txt
echo "== kbench pair bn=5"; KBENCH_PAIR=1 timeout 120 ./tools/kbench $LIB 5 20 2>&1 | tee gpurun_out/kbench_pair5.txt
echo "== bench split"; timeout 300 python bench.py --no-cpu-baseline --layer-only > gpurun_out/bench_split.txt 2>&1; tail -1 gpurun_out/bench_split.txt | cut -c1-260; tail -1 gpurun_out/bench_split.txt | grep -o '"kernels".*' | cut -c1-400
echo "== bench SGR_SPLIT=0"; SGR_SPLIT=0 timeout 300 python bench.py --no-cpu-baseline --layer-only > gpurun_out/bench_nosplit.txt 2>&1; tail -1 gpurun_out/bench_nosplit.txt | cut -c1-260; tail -1 gpurun_out/bench_nosplit.txt | grep -o '"kernels".*' | cut -c1-400
echo "== wavetrace"; WAVETRACE_SPLIT_ONLY=1 timeout 120 ./tools/wavetrace inverserenderingofindoorscene_amd/variants/libsgrender_trace.so 16 > gpurun_out/wavetrace.txt 2>gpurun_out/wavetrace.err; python tools/wavetrace_report.py gpurun_out/wavetrace.txt > gpurun_out/wavetrace_report.txt 2>&1; grep -v "prologue" gpurun_out/wavetrace_report.txt | cut -c1-260
rm -f gpurun_out/wavetrace.txt
