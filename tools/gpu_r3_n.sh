#!/bin/bash
# evidence of the tree with the row-block K3: counters + kernel stats, the bench line, the whole GPU suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r3; mkdir -p $O
bash profiles/collect.sh r03 all > $O/collect.log 2>&1
cp gpurun_out/prof_r03/counters.json profiles/counters.json
timeout 600 python bench.py > $O/bench_final.json 2> $O/bench_final.err
tail -c 600 $O/bench_final.json
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 | tee $O/pytest_gpu.log
