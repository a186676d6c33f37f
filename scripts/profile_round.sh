#!/bin/bash
# on the GPU box: the rocprofv3 evidence behind bench.py's numbers (kernel-trace stats, then PMC passes); summaries land
# in gpurun_out/prof_round/ and are copied into profiles/ by scripts/profile_collect.py
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_round
mkdir -p $OUT
rm -rf $OUT/stats $OUT/pmc_*
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r1 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extras > $OUT/bench_stats.json 2> /dev/null
# separate --pmc passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; 8 SQ slots per pass)
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM SQ_INSTS_SMEM" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_IOPS"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$tag -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extras --steps 3 --warmup 2 > /dev/null 2>&1
done
# the condensed solver (small batches: 256 instances x 8 agents = 2048 QPs) incl. its MFMA counters, and the expansion kernel
for grp in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_small_$tag -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extras --batch 256 --steps 3 --warmup 2 > /dev/null 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_small -o r1 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extras --batch 256 > $OUT/bench_small.json 2> /dev/null
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_expand_$tag -o p -- python3 $GRAFT_REPO_ROOT/scripts/expand_timing.py > /dev/null 2>&1
done
ls $OUT $OUT/stats | head -30
