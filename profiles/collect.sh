#!/bin/bash
# Collects the rocprofv3 evidence for the K1 numbers on the GPU box (run through gpurun):
#   profiles/collect.sh <tag> [kt|pmc|all]
# kernel trace + stats of the default bench, and the PMC passes (own runs, counters only) from which
# profiles/<tag>_pmc_summary.txt is derived.  Raw databases stay under gpurun_out/ (scratch).
set -u
TAG=${1:-r01}; WHAT=${2:-all}
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
if [ $WHAT = kt ] || [ $WHAT = all ]; then
  rocprofv3 --kernel-trace --stats -d $OUT -o kt -- python bench.py --steps 10 --warmup 2 --cpu-seconds 0 > $OUT/bench_kt.log 2>&1
  python profiles/summarize_rocpd.py $OUT/kt_results.db $OUT/kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --cpu-seconds 0, 1x MI355X" > /dev/null
fi
if [ $WHAT = pmc ] || [ $WHAT = all ]; then
  i=0
  for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
             "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" \
             "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAIT_ANY" \
             "SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_LDS_BANK_CONFLICT"; do
    i=$((i+1))
    rocprofv3 --pmc $grp -d $OUT -o pmc$i -- python bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $OUT/bench_pmc$i.log 2>&1
    python profiles/summarize_rocpd.py $OUT/pmc${i}_results.db $OUT/pmc$i.txt "pass $i: $grp" > /dev/null
  done
fi
ls $OUT
