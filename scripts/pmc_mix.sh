#!/bin/bash
# on the GPU box: instruction-mix PMC pass over bench.py (dev aid): per-dispatch averages per kernel
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_IOPS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extras --steps 3 --warmup 2 > /dev/null 2>&1
done
python3 - <<PY
import csv, collections, glob
agg=collections.defaultdict(float); disp=collections.defaultdict(set)
for f in glob.glob('$OUT/g*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=(r['Kernel_Name'].split('(')[0].replace('void ','')[:40], r['Counter_Name'])
        agg[k]+=float(r['Counter_Value']); disp[k].add(r['Dispatch_Id'])
for (k,c),v in sorted(agg.items()):
    if 'qp_quad' in k or 'interaction' in k: print('%-40s %-28s %.4g'%(k,c,v/len(disp[(k,c)])))
PY
