#!/bin/bash
# Collects the rocprofv3 evidence on the GPU box (run through gpurun) and turns it into the committed
# summaries:
#   profiles/collect.sh <tag> [k1|k2|k2r|k2b|k3|u|all]   (k2r: K2's row kernel on the cell-major copy, k2b: on the band-limited copy)
# Per workload: one --kernel-trace --stats run (kernel durations) and the PMC passes (own runs, counters
# only -- never combined with a trace).  K1 = the default bench (python bench.py, 256^3 icosphere);
# K2 / K3 / U = profiles/pmc_workloads.py at BASELINE configs[4] sizes.  Raw databases stay under
# /tmp on the box (too big to travel); profiles/summarize_pmc.py writes profiles/<tag>_pmc_summary.txt,
# profiles/<tag>_kernel_stats.txt and the machine-readable profiles/counters.json that bench.py reads
# (keyed by a hash of discregrid_amd/csrc, so that stale counters are never reported).
set -u
TAG=${1:-r02b}; WHAT=${2:-all}
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
python -c "import torch" 2>/dev/null   # (page the image in before the first profiled run)
OUT=/tmp/prof_$TAG; mkdir -p $OUT gpurun_out/prof_$TAG   # raw databases (100+ MB) stay on the box; summaries travel
GROUPS_BASE=("SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
             "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum")
GROUPS_VMEM=("TA_TA_BUSY_sum TD_TD_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum")   # K3 / K2 rows: is the vector memory pipeline the roof?
GROUPS_K1=("SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_LDS_BANK_CONFLICT")
run_set () {  # name, command..., then the counter groups come from the global array GRP
  local name=$1; shift
  timeout 240 rocprofv3 --kernel-trace --stats -d $OUT -o ${name}_kt -- "$@" > $OUT/${name}_kt.log 2>&1
  local i=0
  for grp in "${GRP[@]}"; do
    i=$((i+1))
    timeout 240 rocprofv3 --pmc $grp -d $OUT -o ${name}_pmc$i -- "$@" > $OUT/${name}_pmc$i.log 2>&1
  done
}
if [ $WHAT = k1 ] || [ $WHAT = all ]; then
  GRP=("${GROUPS_BASE[@]}" "${GROUPS_K1[@]}")
  run_set k1 python bench.py --steps 10 --warmup 2 --no-extras
fi
for w in k2 k2r k2b k3 u; do
  if [ $WHAT = $w ] || [ $WHAT = all ]; then
    GRP=("${GROUPS_BASE[@]}")
    if [ $w = k3 ] || [ $w = k2 ] || [ $w = k2r ] || [ $w = k2b ]; then GRP=("${GROUPS_BASE[@]}" "${GROUPS_VMEM[@]}"); fi
    if [ $w = k3 ]; then GRP=("${GROUPS_BASE[@]}" "${GROUPS_VMEM[@]}" "${GROUPS_K1[@]}"); fi   # (+ instruction mix and the LDS pipe: k_density_cells keeps its tables and sums there)
    run_set $w python profiles/pmc_workloads.py $w
  fi
done
python profiles/summarize_pmc.py $OUT $TAG
cp profiles/counters.json profiles/${TAG}_pmc_summary.txt profiles/${TAG}_kernel_stats.txt gpurun_out/prof_$TAG/
cp $OUT/*.log gpurun_out/prof_$TAG/ 2>/dev/null
