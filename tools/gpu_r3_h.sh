#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$(pwd)
O=gpurun_out/r3; mkdir -p $O
python - <<'PY'
import sys; sys.path.insert(0,'tests')
import dgtest as T
V,F=T.icosphere(71); T.write_obj('/tmp/ico71.obj',V,F)
PY
(tests/cpp/build/host_api_driver flowtrace /tmp/ico71.obj "256 256 256" 0.1 10000000) > $O/flowtrace.log 2>&1
D=/tmp/mc; rm -rf $D; mkdir -p $D
(cd /tmp && timeout 240 rocprofv3 --memory-copy-trace --kernel-trace --stats -d $D -o flow -- $R/tests/cpp/build/host_api_driver flowtrace /tmp/ico71.obj "256 256 256" 0.1 10000000) > $D/log.txt 2>&1
tail -2 $D/log.txt >> $O/flowtrace.log
python tools/memcopy_summary.py $D > $O/flow_memcopy.txt 2>&1
python tools/pmc_dump.py $D "" | sort -k3,3 | head -40 >> $O/flow_memcopy.txt
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > $O/t4.log
