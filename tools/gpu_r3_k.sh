#!/bin/bash
# K3 row-block kernel: lane shapes and block shapes against the pair kernel, with the reference digests
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r3; mkdir -p $O
timeout 300 python tools/k3_run.py --res 128 --steps 3 --check --sweep "DG_K3_ROWS=0;DG_K3_ROWS=1;DG_K3_ROWS=2;DG_K3_ROWS=3;DG_K3_ROWS=4;DG_K3_ROWS=5;DG_K3_ROWS=1,DG_K3_RB0=1,DG_K3_RB1=8,DG_K3_RB2=16;DG_K3_ROWS=1,DG_K3_RB0=4,DG_K3_RB1=8,DG_K3_RB2=4;DG_K3_ROWS=1,DG_K3_RB0=2,DG_K3_RB1=4,DG_K3_RB2=16;DG_K3_ROWS=1,DG_K3_RB0=2,DG_K3_RB1=16,DG_K3_RB2=4" > $O/k3_rows_128.log 2>&1
timeout 300 python tools/k3_run.py --res 256 --steps 2 --check --sweep "DG_K3_ROWS=0;DG_K3_ROWS=1;DG_K3_ROWS=2;DG_K3_ROWS=3;DG_K3_ROWS=1,DG_K3_RB0=1,DG_K3_RB1=8,DG_K3_RB2=16;DG_K3_ROWS=1,DG_K3_RB0=4,DG_K3_RB1=8,DG_K3_RB2=4" > $O/k3_rows_256.log 2>&1
cat $O/k3_rows_128.log $O/k3_rows_256.log
