#!/bin/bash
# GPU box: headline bench (no extras) for each library variant build/libmpcx_<name>.so given on the command line ("tree" = the in-tree library)
for v in "$@"; do
  if [ "$v" = tree ]; then unset MPCX_LIB; else export MPCX_LIB=build/libmpcx_$v.so; fi
  python bench.py --no-extras --no-cpu > gpurun_out/vb_$v.log 2>&1
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
try:
    d = json.loads([x for x in open('gpurun_out/vb_%s.log' % v) if x.startswith('{')][-1])
    print('%-6s ms/step %.4f  qp kernel %.4f ms  mean its %.3f max %d  fail %d' % (v, d['ms_per_step'], d['roofline']['kernel_ms'], d['mean_ipm_iters'], d['max_ipm_iters'], d['qp_failures']))
except Exception as e:
    print(v, 'failed', e)
PY
done
