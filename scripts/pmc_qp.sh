#!/bin/bash
# on the GPU box: PMC pass over bench.py (dev aid): per-dispatch averages for the QP kernel
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/a -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extras --steps 5 > /dev/null 2>&1
python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open('$OUT/a/p_counter_collection.csv')))
agg=collections.defaultdict(float); disp=set()
for r in rows:
    if 'qp_quad' in r['Kernel_Name']:
        agg[r['Counter_Name']]+=float(r['Counter_Value']); disp.add(r['Dispatch_Id'])
for k,v in sorted(agg.items()): print('%-26s %.4g'%(k,v/len(disp)))
print('dispatches', len(disp))
PY
