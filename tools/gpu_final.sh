#!/bin/bash
# end-of-round evidence (one session, one box): tests, smoke, kbench, bench (plain + torchrun), rocprof kernel stats of the bench loop
# and of the config-3 step, PMC traffic + SQ counters (config 2 and 5), config 5 incl. objective legs, batch sweep, host overhead.
# Everything lands in gpurun_out/; copy what is to be judged into profiles/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
LIB=inverserenderingofindoorscene_amd/libsgrender.so
echo "== pytest gpu"; t0=$SECONDS; timeout 1500 python -m pytest tests -q -m gpu --durations=6 -s > gpurun_out/pytest_gpu.txt 2>&1; echo "pytest wall $((SECONDS-t0)) s"; tail -10 gpurun_out/pytest_gpu.txt | cut -c1-200
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.txt | cut -c1-600
echo "== kbench"; timeout 300 ./tools/kbench $LIB 16 20 > gpurun_out/kbench.txt 2>&1; cat gpurun_out/kbench.txt
# bench.py writes the ONE JSON line to stdout and everything else (RCCL's banner included) to stderr: keep the two apart
echo "== bench"; t0=$SECONDS; timeout 900 python bench.py > gpurun_out/bench.txt 2> gpurun_out/bench.err; echo "bench wall $((SECONDS-t0)) s, stdout lines: $(wc -l < gpurun_out/bench.txt), line bytes: $(wc -c < gpurun_out/bench.txt)" | tee gpurun_out/bench_time.txt; tail -1 gpurun_out/bench.txt | cut -c1-3200; cp bench_detail.json gpurun_out/bench_detail.json
echo "== bench, exactly as the driver runs it"; t0=$SECONDS; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_warmup5.txt 2> gpurun_out/bench_warmup5.err; echo "wall $((SECONDS-t0)) s"; tail -1 gpurun_out/bench_warmup5.txt | cut -c1-900; cp bench_detail.json gpurun_out/bench_warmup5_detail.json
echo "== bench --gpus 2 / 4 / 8 WITHOUT torchrun over gloo on this one GPU (rehearsal of the driver's N > 1 launches: flagged config.rehearsal, no measurement)"; for n in 2 4 8; do SGR_BENCH_BACKEND=gloo SGR_BENCH_DETAIL=gpurun_out/rehearsal_n${n}_detail.json timeout 900 python bench.py --gpus $n --steps 5 --warmup 3 --reps 3 > gpurun_out/rehearsal_n$n.json 2> gpurun_out/rehearsal_n$n.err; echo "N=$n rc=$? $(cut -c1-160 gpurun_out/rehearsal_n$n.json)"; done
echo "== bench torchrun world=1 (RCCL init / barrier / all-reduce path)"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --no-cpu-baseline --layer-only > gpurun_out/bench_torchrun1.txt 2> gpurun_out/bench_torchrun1.err; tail -1 gpurun_out/bench_torchrun1.txt | cut -c1-500
echo "== host overhead"; timeout 300 python tools/host_overhead.py 2>&1 | tail -10 | tee gpurun_out/host_overhead.txt
echo "== rocprof bench loop"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 20 --reps 2 --no-cpu-baseline --no-config5 > $GRAFT_REPO_ROOT/gpurun_out/rocprof.txt 2>&1; cd $GRAFT_REPO_ROOT
for f in $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); do cp $f gpurun_out/kernel_stats.csv; head -12 $f | cut -c1-200; done
find gpurun_out/prof -name "*kernel_trace.csv" -size +1M -delete
echo "== rocprof config-3 step (examples/train_light_synthetic.py)"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof3 -- python $GRAFT_REPO_ROOT/examples/train_light_synthetic.py --batch 16 --steps 43 > $GRAFT_REPO_ROOT/gpurun_out/rocprof3.txt 2>&1; cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/rocprof3.txt
for f in $(find gpurun_out/prof3 -name "*kernel_stats.csv" | head -1); do cp $f gpurun_out/kernel_stats_config3.csv; python tools/config3_breakdown.py $f 43 | tee gpurun_out/config3_breakdown.txt; done
find gpurun_out/prof3 -name "*kernel_trace.csv" -size +1M -delete
echo "== pmc traffic config 2"; bash tools/pmc_traffic.sh config2_batch16_env | grep -E "fwd_pk|sg_bwd_pk"
echo "== pmc sq config 2"; bash tools/pmc_sq.sh config2_batch16_env | grep -E "fwd_pk|sg_bwd_pk"
# round 4: the kernels of the real trainLight step -- the fused objective (premap 1) and with the decoder heads as prologue (premap 3)
echo "== pmc traffic config 2, objective"; bash tools/pmc_traffic.sh config2_batch16_objective --pmc-workload objective | grep -E "fwd_pk|sg_bwd_recon"
echo "== pmc sq config 2, objective"; bash tools/pmc_sq.sh config2_batch16_objective --pmc-workload objective | grep -E "fwd_pk|sg_bwd_recon"
echo "== pmc traffic config 2, objective + heads"; bash tools/pmc_traffic.sh config2_batch16_objective_heads --pmc-workload objective_heads | grep -E "fwd_pk|sg_bwd_recon"
echo "== pmc sq config 2, objective + heads"; bash tools/pmc_sq.sh config2_batch16_objective_heads --pmc-workload objective_heads | grep -E "fwd_pk|sg_bwd_recon"
echo "== pmc traffic config 5"; bash tools/pmc_traffic.sh config5_batch4_env --config 5 | grep -E "fwd_pk|sg_bwd_pk"
echo "== pmc sq config 5"; bash tools/pmc_sq.sh config5_batch4_env --config 5 | grep -E "fwd_pk|sg_bwd_pk"
echo "== pmc traffic config 5, objective"; bash tools/pmc_traffic.sh config5_batch4_objective --config 5 --pmc-workload objective | grep -E "fwd_pk|sg_bwd_recon"
echo "== pmc sq config 5, objective"; bash tools/pmc_sq.sh config5_batch4_objective --config 5 --pmc-workload objective | grep -E "fwd_pk|sg_bwd_recon"
echo "== config 5"; timeout 900 python bench.py --config 5 --no-cpu-baseline > gpurun_out/bench_config5.txt 2> gpurun_out/bench_config5.err; tail -1 gpurun_out/bench_config5.txt | cut -c1-1800
# the counter records are fresh now: the line of record carries them un-stale (config 5's leg included)
echo "== bench with the fresh counter records (driver-style flags)"; cp gpurun_out/traffic.json profiles/traffic.json; cp gpurun_out/sq.json profiles/sq.json; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_fresh_records.txt 2> gpurun_out/bench_fresh_records.err; tail -1 gpurun_out/bench_fresh_records.txt | cut -c1-3200; cp bench_detail.json gpurun_out/bench_fresh_records_detail.json
echo "== FETCH_SIZE calibration on known byte counts (tools/fetch_calib)"; timeout 300 ./tools/fetch_calib 10 | tee gpurun_out/fetch_calib_timing.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/calib_fetch -- $GRAFT_REPO_ROOT/tools/fetch_calib 1 > /dev/null 2>&1; timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/calib_rdreq -- $GRAFT_REPO_ROOT/tools/fetch_calib 1 > /dev/null 2>&1)
python tools/fetch_calib_report.py gpurun_out | tee gpurun_out/fetch_calib_counters.txt
echo "== wavetrace (schedule + shader clock under load)"; if [ -f inverserenderingofindoorscene_amd/variants/libsgrender_trace.so ]; then timeout 300 ./tools/wavetrace inverserenderingofindoorscene_amd/variants/libsgrender_trace.so 16 > /tmp/trace.txt 2>/dev/null; python tools/wavetrace_report.py /tmp/trace.txt > gpurun_out/wavetrace_report.txt 2>&1; grep -E "^[a-z_0-9]+:|shader clock" gpurun_out/wavetrace_report.txt | cut -c1-200; fi
echo "== batch sweep"; for b in 5 8 16 32 64; do st=100; wu=300; if [ $b -ge 32 ]; then st=40; wu=80; fi; timeout 300 python bench.py --batch $b --steps $st --warmup $wu --no-cpu-baseline --layer-only --graph-leg 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; g=c.get('ms_with_loss_graph'); print('batch $b', d['value'], 'Mpix/s', d['ms_per_step'], 'ms/step  fwd', c['fwd_us'], 'us  bwd', c['bwd_us'], 'us  with loss', c['Mpix_with_loss'], 'Mpix/s  with loss, replayed from a HIP graph', None if g is None else round($b * 240 * 320 / (g * 1e-3) / 1e6, 1))"; done | tee gpurun_out/bench_batch_sweep.txt
echo "== trainlight example (fused objective + HIP heads | unfused + torch heads)"
timeout 300 python examples/train_light_synthetic.py --batch 16 --steps 23 2>&1 | tail -1 | tee gpurun_out/trainlight_fused.txt
timeout 300 python examples/train_light_synthetic.py --batch 16 --steps 23 --unfused --torch-heads 2>&1 | tail -1 | tee gpurun_out/trainlight_unfused.txt
