#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r3; mkdir -p $O; rm -f $O/k3_align.log
for w in 2 0; do
  echo "DG_K3_PAIRS=$w" >> $O/k3_align.log
  (DG_K3_PAIRS=$w timeout 200 python tools/k3_run.py --res 128 --steps 3 --check; DG_K3_PAIRS=$w timeout 200 python tools/k3_run.py --res 256 --steps 2) >> $O/k3_align.log 2>&1
done
(timeout 600 python -m pytest tests/test_gpu_density_map.py tests/test_gpu_digests.py tests/test_gpu_parity.py -x -q 2>&1 | tail -4) >> $O/k3_align.log
