#!/bin/bash
# bench sweep: DG_FORCE max_leaf x k1_fast (kernel-only runs)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=gpurun_out/k1_sweep.txt; : > $OUT
for leaf in ${LEAVES:-2 4 6}; do
  for fast in ${FASTS:-0 1}; do
    DG_FORCE="max_leaf=$leaf;k1_fast=$fast" timeout 300 python bench.py --steps 10 --warmup 3 --no-extras > /tmp/b.json 2> /tmp/b.err
    python - <<PY >> $OUT
import json
d=json.load(open("/tmp/b.json"))
print("leaf=$leaf fast=$fast %.1f Mnodes/s %.3f ms" % (d["value"], d["ms_per_step"]))
PY
  done
done
cat $OUT
