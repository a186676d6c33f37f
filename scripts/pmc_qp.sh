#!/bin/bash
# on the GPU box: PMC passes for scripts/qp_timing.py with the library currently installed (dev aid)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/a -o p -- python3 $GRAFT_REPO_ROOT/scripts/qp_timing.py 20 32768 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -o p -- python3 $GRAFT_REPO_ROOT/scripts/qp_timing.py 20 32768 > /dev/null 2>&1
python3 - <<PY
import csv, collections
for d in ('a','b'):
    rows=list(csv.DictReader(open('$OUT/%s/p_counter_collection.csv'%d)))
    agg=collections.defaultdict(float); disp=set()
    for r in rows:
        if 'qp_' in r['Kernel_Name']:
            agg[r['Counter_Name']]+=float(r['Counter_Value']); disp.add(r['Dispatch_Id'])
    for k,v in sorted(agg.items()): print('%-26s %.4g'%(k,v/len(disp)))
PY
