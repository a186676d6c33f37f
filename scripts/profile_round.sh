#!/bin/bash
# on the GPU box: the rocprofv3 evidence behind bench.py's numbers (kernel-trace stats, then PMC passes); summaries land
# in gpurun_out/prof_round/ and are copied into profiles/ by scripts/profile_collect.py
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_round
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r1 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu > $OUT/bench_stats.json 2> /dev/null
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$tag -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 3 --warmup 2 > /dev/null 2>&1
done
ls $OUT $OUT/stats | head -30
