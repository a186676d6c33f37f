#!/bin/bash
# on the GPU box: rocprofv3 kernel-trace stats of bench.py (dev aid).  usage: kstats.sh <tag> [bench args]
cd /tmp && export TMPDIR=/tmp
tag=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu "$@" > $OUT/bench.json 2> $OUT/bench.err
f=$(find $OUT/stats -name '*kernel_stats.csv' | head -1)
python3 - <<PY
import csv
for r in list(csv.DictReader(open('$f')))[:12]:
    print('%-60s calls %5s avg %10.1f us  %5.1f%%' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
python3 -c "
import json; d=json.load(open('$OUT/bench.json')); print('value %.0f ms/step %.4f kernel_ms %.4f iters %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['mean_ipm_iters']))"
