#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r3; mkdir -p $O
python - <<'PY'
import sys; sys.path.insert(0,'tests')
import dgtest as T
V,F=T.icosphere(71); T.write_obj('/tmp/ico71.obj',V,F)
PY
(tests/cpp/build/unchanged_caller addfunction /tmp/ico71.obj "256 256 256" 7) > $O/addfn_lazy4.log 2>&1
(DG_LAZY_HOST=0 tests/cpp/build/unchanged_caller addfunction /tmp/ico71.obj "256 256 256" 4) > $O/addfn_eager4.log 2>&1
timeout 500 bash tools/gpu_k3_pmc.sh base 128
