#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r3; mkdir -p $O
python - <<'PY'
import sys; sys.path.insert(0,'tests')
import dgtest as T
V,F=T.icosphere(71); T.write_obj('/tmp/ico71.obj',V,F)
PY
(tests/cpp/build/unchanged_caller addfunction /tmp/ico71.obj "256 256 256" 6) > $O/addfn_warm.log 2>&1
(DG_ADDFN_NO_WARM=1 tests/cpp/build/unchanged_caller addfunction /tmp/ico71.obj "256 256 256" 6) > $O/addfn_cold.log 2>&1
(timeout 400 python tests/perf/fuzz_parity.py 240 33 2>&1 | tail -4) > $O/fuzz.log
