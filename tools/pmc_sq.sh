#!/bin/bash
# SQ issue/stall counters for the hot kernels (kbench as workload) -> gpurun_out/pmc_sq/
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq
mkdir -p $OUT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/a -- $GRAFT_REPO_ROOT/tools/kbench $GRAFT_REPO_ROOT/inverserenderingofindoorscene_amd/libsgrender.so 16 3 > $OUT/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- $GRAFT_REPO_ROOT/tools/kbench $GRAFT_REPO_ROOT/inverserenderingofindoorscene_amd/libsgrender.so 16 3 > $OUT/b.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, os, collections
root=os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out','pmc_sq')
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root,'*','**','*counter_collection.csv'),recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        if 'sgr::' in k: acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
with open(os.path.join(root,'summary.txt'),'w') as out:
    for k,v in acc.items():
        line=k+'\n   '+'  '.join('%s=%.4g'%(c,sum(x)/len(x)) for c,x in sorted(v.items()))
        print(line); out.write(line+'\n')
PY
