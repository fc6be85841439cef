#!/bin/bash
# A/B of the K1 variants on the GPU box (run through gpurun): parity suite, then bench with the exact
# kernel only (DG_FORCE=k1_fast=0) and with the filtered kernel (default), then a kernel trace of the default.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out
(time timeout 900 python -m pytest tests -m gpu -x -q) > gpurun_out/ab_gputest.log 2>&1; tail -3 gpurun_out/ab_gputest.log
for fast in 0 1; do
  DG_FORCE="k1_fast=$fast" timeout 300 python bench.py --steps 20 --warmup 5 --no-extras > gpurun_out/ab_bench_fast$fast.json 2> gpurun_out/ab_bench_fast$fast.err
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_bench_fast$fast.json"))
print("k1_fast=$fast", d["value"], "Mnodes/s", d["ms_per_step"], "ms")
PY
done
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_ab -o ab -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-extras > /tmp/prof_ab.log 2>&1
cd "$GRAFT_REPO_ROOT"; DB=$(ls /tmp/prof_ab/*/*.db /tmp/prof_ab/*.db 2>/dev/null | head -1)
python profiles/summarize_rocpd.py "$DB" gpurun_out/ab_kernel_stats.txt "K1 A/B: filtered kernel (default)" | head -24
