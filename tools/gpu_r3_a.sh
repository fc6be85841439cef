#!/bin/bash
# round-3 probe A: addFunction timing breakdown, available counters, new bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r3; mkdir -p $O
python - <<'PY'
import sys; sys.path.insert(0,'tests')
import dgtest as T
V,F=T.icosphere(71); T.write_obj('/tmp/ico71.obj',V,F)
PY
(DG_HOST_DEBUG=1 tests/cpp/build/unchanged_caller addfunction /tmp/ico71.obj "256 256 256" 4) > $O/addfn_debug.log 2>&1
(DG_LAZY_HOST=0 tests/cpp/build/unchanged_caller addfunction /tmp/ico71.obj "256 256 256" 4) > $O/addfn_eager.log 2>&1
(cd /tmp && rocprofv3 --list-avail) > $O/avail.txt 2>&1
(timeout 400 python bench.py 2>&1 | tail -2) > $O/b2.log
