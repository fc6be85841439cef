#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r3; mkdir -p $O; rm -f $O/k3_pairs.log
(timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > $O/t2.log
for w in 2 0; do
  echo "DG_K3_PAIRS=$w" >> $O/k3_pairs.log
  (DG_K3_PAIRS=$w timeout 200 python tools/k3_run.py --res 128 --steps 3 --check; DG_K3_PAIRS=$w timeout 200 python tools/k3_run.py --res 256 --steps 2) >> $O/k3_pairs.log 2>&1
done
(timeout 600 python bench.py 2>&1 | tail -1) > $O/b4.log
