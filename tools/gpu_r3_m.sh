#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r3; mkdir -p $O
timeout 300 python tools/k3_run.py --res 128 --steps 3 --check --sweep "DG_K3_FACE=0;DG_K3_FACE=1" > $O/k3_face_128.log 2>&1
timeout 300 python tools/k3_run.py --res 256 --steps 2 --check --sweep "DG_K3_FACE=0;DG_K3_FACE=1;DG_K3_FACE=0;DG_K3_FACE=1" > $O/k3_face_256.log 2>&1
cat $O/k3_face_128.log $O/k3_face_256.log
