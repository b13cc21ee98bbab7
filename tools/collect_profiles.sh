#!/bin/bash
# copy one tools/gpu_final.sh session from gpurun_out/ (scratch) into profiles/ under a tag:  tools/collect_profiles.sh r04l
set -eu
T=$1; G=gpurun_out; P=profiles
last() { tail -1 "$1" > "$2"; }
last $G/bench.txt $P/${T}_bench.json
last $G/bench_warmup5.txt $P/${T}_bench_warmup5.json
last $G/bench_torchrun1.txt $P/${T}_bench_torchrun_world1.json
last $G/bench_config5.txt $P/${T}_bench_config5.json
for f in bench_batch_sweep config3_breakdown host_overhead kbench pytest_gpu smoke trainlight_fused trainlight_unfused; do cp $G/$f.txt $P/${T}_$f.txt; done
cp $G/kernel_stats.csv $P/${T}_kernel_stats.csv
cp $G/kernel_stats_config3.csv $P/${T}_kernel_stats_config3.csv
for d in $G/pmc_*; do
  w=${d#$G/pmc_}; [ -f $d/summary.txt ] || continue
  case $w in sq_*) cp $d/summary.txt $P/${T}_sq_${w#sq_}.txt;; *) cp $d/summary.txt $P/${T}_pmc_traffic_$w.txt;; esac
done
for f in bench_fresh_records; do [ -f $G/$f.txt ] && last $G/$f.txt $P/${T}_$f.json; done
[ -f $G/wavetrace_report.txt ] && cp $G/wavetrace_report.txt $P/${T}_wavetrace_report.txt
cp $G/traffic.json $G/sq.json $P/
ls $P | grep "^${T}_" | wc -l
# round 6: the nested detail of the flat contract line, the N > 1 rehearsal lines, the FETCH_SIZE calibration
for f in bench_detail bench_warmup5_detail bench_fresh_records_detail; do [ -f $G/$f.json ] && cp $G/$f.json $P/${T}_$f.json; done
for n in 2 4 8; do [ -f $G/rehearsal_n$n.json ] && cp $G/rehearsal_n$n.json $P/${T}_rehearsal_gloo_n$n.json; done
for f in fetch_calib_timing fetch_calib_counters; do [ -f $G/$f.txt ] && cp $G/$f.txt $P/${T}_$f.txt; done
ls $P | grep "^${T}_" | wc -l
