#!/bin/bash
# round 4, session B: k_density_cells with the division-free gamma; counters of the new kernel
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r4b
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_density_map.py -x -q -m gpu > gpurun_out/r4b/density_tests.log 2>&1
tail -3 gpurun_out/r4b/density_tests.log
timeout 300 python tools/k3_run.py --res 128 --steps 2 --check --sweep "DG_K3_CELLS=0;DG_K3_CELLS=1;DG_K3_CELLS=1,DG_K3_WAVES3=0" > gpurun_out/r4b/k3_128.jsonl 2> gpurun_out/r4b/k3_128.err
cat gpurun_out/r4b/k3_128.jsonl
timeout 600 python tools/k3_run.py --res 256 --steps 2 --check --sweep "DG_K3_CELLS=0;DG_K3_CELLS=1;DG_K3_CELLS=1,DG_K3_WAVES3=0;DG_K3_CELLS=1,DG_K3_RB1=8,DG_K3_RB2=8;DG_K3_CELLS=1,DG_K3_RB1=8,DG_K3_RB2=16;DG_K3_CELLS=1,DG_K3_RB1=4,DG_K3_RB2=8" > gpurun_out/r4b/k3_256.jsonl 2> gpurun_out/r4b/k3_256.err
cat gpurun_out/r4b/k3_256.jsonl
timeout 900 bash profiles/collect.sh r04a k3 > gpurun_out/r4b/collect.log 2>&1
tail -12 gpurun_out/r4b/collect.log
