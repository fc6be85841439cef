#!/bin/bash
# K3 (128^3) counters of the vector memory path: TA / TCP / TD / SQ VMEM.  Output: gpurun_out/r3/k3_pmc_<tag>.txt
TAG=${1:-base}; RES=${2:-128}
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$(pwd)
O=gpurun_out/r3; mkdir -p $O; D=/tmp/k3pmc_$TAG; rm -rf $D; mkdir -p $D
PMCG=("GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TD_TD_BUSY_sum" \
      "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
      "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
      "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS")
python -c "import torch" 2>/dev/null   # (page the image in before the clock of the first profiled run starts)
i=0
for grp in "${PMCG[@]}"; do
  i=$((i+1))
  (cd /tmp && timeout 90 rocprofv3 --pmc $grp -d $D -o p$i -- python $R/tools/k3_run.py --res $RES --steps 1) > $D/p$i.log 2>&1
done
(cd /tmp && timeout 90 rocprofv3 --kernel-trace --stats -d $D -o kt -- python $R/tools/k3_run.py --res $RES --steps 2) > $D/kt.log 2>&1
python tools/pmc_dump.py $D k_density > $O/k3_pmc_$TAG.txt 2>&1; for f in $D/p*.log; do echo "== $f"; grep -i "error\|unable\|missing" $f | head -3; done >> $O/k3_pmc_$TAG.txt
tail -3 $D/kt.log >> $O/k3_pmc_$TAG.txt; tail -5 $D/p1.log >> $O/k3_pmc_$TAG.txt; find $D -name "*.db" | head >> $O/k3_pmc_$TAG.txt
