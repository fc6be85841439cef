#!/bin/bash
# gpurun helper (round 6): what the STRONG-scaling step (the metric's own 256^3 lattice shared by N ranks) costs beyond its kernels --
# N processes on the ONE GPU of the box, every exchange form the one-GPU rig can run with real library paths (host vector, copy on the
# shared-memory control plane), 1 / 2 / 4 pieces.  On one device the ranks' kernels run one after the other (their sum is the 1-GPU
# kernel time), so step time - 14.2 ms is what launches, barriers and copy enqueues add per step.  One line per run.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/${1:-strong}; mkdir -p $O
python -c "import torch" 2>/dev/null
for n in 2 4; do for form in host copy-shm; do for pieces in 1 2 4; do
  DG_BENCH_SELFTEST_ONE_GPU=1 timeout 300 python bench.py --gpus $n --steps 5 --warmup 2 --scaling strong --exchange $form --pieces $pieces --no-preflight \
      > $O/s_${n}_${form}_${pieces}.json 2> $O/s_${n}_${form}_${pieces}.err
  python - $O/s_${n}_${form}_${pieces}.json $n $form $pieces <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    pr = r["config"]["exchange"]["per_rank"]
    print("N=%s %-9s pieces=%s  step %.3f ms  sum of the ranks' sampling %.3f ms  exchange wait per rank %s" % (
        sys.argv[2], sys.argv[3], sys.argv[4], r["ms_per_step"], sum(sum(x) for x in pr["sample_ms"]), pr["exchange_wait_ms"]))
except Exception as e:
    print("N=%s %s pieces=%s FAILED %s" % (sys.argv[2], sys.argv[3], sys.argv[4], e))
PY
done; done; done | tee $O/strong_budget.txt
