#!/bin/bash
# round-3 session A: MFMA co-issue microbenchmark, kernel A/B (round-2 library vs packed pre-map / tan hand-off / mixed grid /
# LDS exchange), quick GPU tests, bench loops
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
LIB=inverserenderingofindoorscene_amd/libsgrender.so
V=inverserenderingofindoorscene_amd/variants
echo "== ubench_mfma"; timeout 200 ./tools/ubench_mfma > gpurun_out/ubench_mfma.txt 2>&1; cat gpurun_out/ubench_mfma.txt
echo "== kbench r02"; timeout 200 ./tools/kbench $V/libsgrender_r02.so 16 20 > gpurun_out/kbench_r02.txt 2>&1; grep -E "^#|fused_fwd|fused_bwd|sg_to_env" gpurun_out/kbench_r02.txt
echo "== kbench new"; timeout 200 ./tools/kbench $LIB 16 20 > gpurun_out/kbench_new.txt 2>&1; cat gpurun_out/kbench_new.txt
echo "== kbench new, SGR_FWD_MIXED=0"; SGR_FWD_MIXED=0 timeout 200 ./tools/kbench $LIB 16 20 2>&1 | grep -E "fused_fwd" | tee gpurun_out/kbench_new_nomixed.txt
echo "== kbench xchg"; timeout 200 ./tools/kbench $V/libsgrender_xchg.so 16 20 2>&1 | grep -E "fused_bwd|sg_to_env_bwd" | tee gpurun_out/kbench_xchg.txt
echo "== kbench cold: r02 / new / new nomixed / xchg"
for spec in "r02 $V/libsgrender_r02.so 1" "new $LIB 1" "nomixed $LIB 0" "xchg $V/libsgrender_xchg.so 1"; do set -- $spec; echo "-- $1"; KBENCH_COLD=1 SGR_FWD_MIXED=$3 timeout 300 ./tools/kbench $2 16 10 2>&1 | grep -E "fused_fwd|fused_bwd" | tee gpurun_out/kbench_cold_$1.txt; done
echo "== pytest gpu (quick)"; t0=$SECONDS; timeout 1200 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_fullsize.py --durations=5 > gpurun_out/pytest_gpu_quick.txt 2>&1; echo "pytest wall $((SECONDS-t0)) s"; tail -12 gpurun_out/pytest_gpu_quick.txt
echo "== bench new (full line)"; timeout 600 python bench.py > gpurun_out/bench_new.txt 2>&1; tail -1 gpurun_out/bench_new.txt | cut -c1-2400
echo "== bench new, no mixed"; SGR_FWD_MIXED=0 timeout 300 python bench.py --layer-only --no-cpu-baseline > gpurun_out/bench_nomixed.txt 2>&1; tail -1 gpurun_out/bench_nomixed.txt | cut -c1-700
echo "== bench xchg"; SGR_LIB=$V/libsgrender_xchg.so timeout 300 python bench.py --layer-only --no-cpu-baseline > gpurun_out/bench_xchg.txt 2>&1; tail -1 gpurun_out/bench_xchg.txt | cut -c1-700
echo "== pytest with xchg library (parity + objective)"; SGR_LIB=$V/libsgrender_xchg.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_objective.py -q -m gpu -x 2>&1 | tail -4
