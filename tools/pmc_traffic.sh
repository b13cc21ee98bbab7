#!/bin/bash
# HBM traffic of the hot kernels from PMC counters (run on the GPU box):
#   two separate rocprofv3 passes (FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2), counters only
#   with --kernel-trace, as MI355X_MICROARCH.md prescribes.
#   usage: tools/pmc_traffic.sh [tag] [bench.py args...]   e.g.  tools/pmc_traffic.sh config5_batch4_env --config 5
#   Output: gpurun_out/pmc_<tag>/*.csv, summary.txt, and gpurun_out/traffic.json (merge of profiles/traffic.json + this tag)
set -u
export TMPDIR=/tmp
TAG=${1:-config2_batch16_env}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
[ -f $GRAFT_REPO_ROOT/gpurun_out/traffic.json ] || cp $GRAFT_REPO_ROOT/profiles/traffic.json $GRAFT_REPO_ROOT/gpurun_out/traffic.json
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --reps 1 --no-cpu-baseline --layer-only "$@" > $OUT/$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/parse_pmc.py $OUT $TAG gpurun_out/traffic.json > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
find $OUT -name "*.csv" -size +2M -delete
