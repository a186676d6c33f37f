"""dev aid: per-step kernel timeline (start offset, duration, queue) from a rocprofv3 kernel trace of bench.py
   usage on the GPU box:  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extras
                          python scripts/step_timeline.py gpurun_out/tl"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'qp_quad' in r['Kernel_Name']]
a, b = idx[-4], idx[-2]
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b + 1]:
    print('%9.1f %8.1f  q%s %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Queue_Id'], r['Kernel_Name'][:70]))
gaps = [(int(rows[i]['Start_Timestamp']) - int(rows[i - 1]['End_Timestamp'])) / 1e3 for i in idx[4:]]
steps = [(int(rows[idx[k + 1]]['Start_Timestamp']) - int(rows[idx[k]]['Start_Timestamp'])) / 1e3 for k in range(4, len(idx) - 1)]
qp = [(int(rows[i]['End_Timestamp']) - int(rows[i]['Start_Timestamp'])) / 1e3 for i in idx[4:-1]]
print('steps %d: mean period %.1f us, mean QP %.1f us, mean non-QP %.1f us' % (len(steps), sum(steps) / len(steps), sum(qp) / len(qp), (sum(steps) - sum(qp)) / len(steps)))
