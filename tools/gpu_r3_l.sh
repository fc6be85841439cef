#!/bin/bash
# K3 row-block kernel: second sweep at 256^3 + counters of the vector memory path at 128^3
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r3; mkdir -p $O
timeout 300 python tools/k3_run.py --res 256 --steps 2 --sweep "DG_K3_ROWS=1;DG_K3_ROWS=4;DG_K3_ROWS=1,DG_K3_RB0=2,DG_K3_RB1=16,DG_K3_RB2=4;DG_K3_ROWS=1,DG_K3_RB0=2,DG_K3_RB1=16,DG_K3_RB2=8;DG_K3_ROWS=1,DG_K3_RB0=1,DG_K3_RB1=16,DG_K3_RB2=8;DG_K3_ROWS=1,DG_K3_RB0=2,DG_K3_RB1=8,DG_K3_RB2=16;DG_K3_ROWS=1,DG_K3_RB0=2,DG_K3_RB1=4,DG_K3_RB2=8;DG_K3_ROWS=1,DG_K3_RB0=1,DG_K3_RB1=32,DG_K3_RB2=4;DG_K3_ROWS=4,DG_K3_RB0=4,DG_K3_RB1=8,DG_K3_RB2=4;DG_K3_ROWS=4,DG_K3_RB0=2,DG_K3_RB1=8,DG_K3_RB2=8;DG_K3_ROWS=3,DG_K3_RB0=2,DG_K3_RB1=8,DG_K3_RB2=16" > $O/k3_rows_256b.log 2>&1
cat $O/k3_rows_256b.log
bash tools/gpu_k3_pmc.sh rows 128
cat $O/k3_pmc_rows.txt | grep -v "k_density_pairs" | head -60
