#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r3; mkdir -p $O
(timeout 1200 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_host_api.py -x -q --durations=8 2>&1 | tail -20) > $O/t3.log
python - <<'PY'
import sys; sys.path.insert(0,'tests')
import dgtest as T
V,F=T.icosphere(71); T.write_obj('/tmp/ico71.obj',V,F)
PY
(tests/cpp/build/unchanged_caller addfunction /tmp/ico71.obj "256 256 256" 5) > $O/addfn_lazy5.log 2>&1
