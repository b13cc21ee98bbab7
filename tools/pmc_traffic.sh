#!/bin/bash
# HBM traffic of the hot kernels from PMC counters (run on the GPU box):
#   two separate rocprofv3 passes (FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2), counters only
#   with --kernel-trace, as MI355X_MICROARCH.md prescribes.  Output: gpurun_out/pmc/*.csv + traffic.json
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/parse_pmc.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
