#!/bin/bash
# on the GPU box: PMC pass over the expansion kernels (dev aid)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/a -o p -- python3 $GRAFT_REPO_ROOT/scripts/expand_timing.py > $OUT/log.txt 2>&1
python3 - <<PY
import csv, collections, glob
agg=collections.defaultdict(float); disp=collections.defaultdict(set); n=collections.defaultdict(float)
for f in glob.glob('$OUT/a/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'expand' not in r['Kernel_Name']: continue
        k=(r['Kernel_Name'].split('(')[0][:30], r['Grid_Size'], r['Counter_Name'])
        agg[k]+=float(r['Counter_Value']); disp[k].add(r['Dispatch_Id'])
for k,v in sorted(agg.items()): print(k, '%.4g'%(v/len(disp[k])))
PY
