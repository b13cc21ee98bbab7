#!/bin/bash
# VALU-issue counters of the hot kernels on a bench.py workload (run on the GPU box):
#   usage: tools/pmc_sq.sh [tag] [bench.py args...]      e.g.  tools/pmc_sq.sh config5_batch4_env --config 5
# Counters-only passes with --kernel-trace (MI355X_MICROARCH.md).  Output: gpurun_out/pmc_sq_<tag>/, summary.txt, and
# gpurun_out/sq.json (merge of profiles/sq.json + this tag) -- what bench.py's `roofline_valu` reads.
set -u
export TMPDIR=/tmp
TAG=${1:-config2_batch16_env}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq_$TAG
mkdir -p $OUT
[ -f $GRAFT_REPO_ROOT/gpurun_out/sq.json ] || cp $GRAFT_REPO_ROOT/profiles/sq.json $GRAFT_REPO_ROOT/gpurun_out/sq.json 2>/dev/null || echo "{}" > $GRAFT_REPO_ROOT/gpurun_out/sq.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --reps 1 --no-cpu-baseline --layer-only "$@" > $OUT/a.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY --output-format csv -d $OUT/b -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --reps 1 --no-cpu-baseline --layer-only "$@" > $OUT/b.log 2>&1
# round 5: where a wave's resident cycles go (quad-cycles summed over waves): issuing by instruction class / waiting
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --output-format csv -d $OUT/c -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --reps 1 --no-cpu-baseline --layer-only "$@" > $OUT/c.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/parse_sq.py $OUT $TAG gpurun_out/sq.json > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
find $OUT -name "*.csv" -size +2M -delete
