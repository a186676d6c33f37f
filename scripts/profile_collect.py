"""Copy the summaries of scripts/profile_round.sh from gpurun_out/prof_round into profiles/ (tracked)."""
import collections, csv, glob, os, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'gpurun_out', 'prof_round')
tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
st = glob.glob(os.path.join(src, 'stats', '**', '*kernel_stats.csv'), recursive=True)
st_small = glob.glob(os.path.join(src, 'stats_small', '**', '*kernel_stats.csv'), recursive=True)
if st_small:
    shutil.copy(st_small[0], os.path.join(root, 'profiles', '%s_kernel_stats_bench_batch256.csv' % tag))
assert st, 'no kernel stats'
shutil.copy(st[0], os.path.join(root, 'profiles', '%s_kernel_stats_bench_default.csv' % tag))
agg = collections.defaultdict(float); disp = collections.defaultdict(set)
for f in glob.glob(os.path.join(src, 'pmc_*', '**', '*counter_collection.csv'), recursive=True):
    part = f[len(src):].split(os.sep)[1]
    scope = 'batch256:' if part.startswith('pmc_small_') else ('frontier:' if part.startswith('pmc_expand_') else '')
    rows = list(csv.DictReader(open(f)))
    big = sorted({int(r['Dispatch_Id']) for r in rows if r['Kernel_Name'].startswith('mpcx::expand_coop_kernel') and int(r['Grid_Size']) >= 9000000})
    free_ids = set(big[:len(big) // 2])       # scripts/expand_timing.py: the free-space frontier (section 8d) first, then round 2's uniform one
    for r in rows:
        name = r['Kernel_Name'].split('(')[0].replace('void ', '')
        sc = scope
        if scope == 'frontier:':
            if not name.startswith('mpcx::expand_coop_kernel') or int(r['Grid_Size']) < 9000000:
                continue                      # only the 2^20-node launches
            if int(r['Dispatch_Id']) not in free_ids:
                sc = 'frontier_uniform:'
        if scope == '' and name.startswith('mpcx::expand'):
            continue
        k = (sc + name, r['Counter_Name'])
        agg[k] += float(r['Counter_Value']); disp[k].add(r['Dispatch_Id'])
with open(os.path.join(root, 'profiles', '%s_pmc_final.csv' % tag), 'w') as f:
    f.write('# %s rocprofv3 --pmc passes over `bench.py --no-cpu --no-extras --steps 3 --warmup 2` (4096 instances x 8 agents, T=20), values PER DISPATCH; rows prefixed batch256: = the same with --batch 256 (2048 QPs: condensed solver), frontier: = expand_coop_kernel on the 2^20-node free-space Prius frontier of SURVEY 8(d), frontier_uniform: = on the uniform frontier of round 2 (scripts/expand_timing.py).\n' % tag)
    f.write('# SQ_* cycle counters are quad-cycles summed over waves. FETCH_SIZE/WRITE_SIZE are in KiB as rocprofv3 reports them; per MI355X_MICROARCH.md '
            'FETCH_SIZE under-reports wide (16 B/lane) reads by 2x and is uncalibrated for the 8 B/lane accesses used here.\n')
    for (k, c), v in sorted(agg.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        f.write('%s,%s,%g\n' % (k, c, v / len(disp[(k, c)])))
print(open(os.path.join(root, 'profiles', '%s_kernel_stats_bench_default.csv' % tag)).read()[:1500])
