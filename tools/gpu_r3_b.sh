#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r3; mkdir -p $O
python - <<'PY'
import sys; sys.path.insert(0,'tests')
import dgtest as T
V,F=T.icosphere(71); T.write_obj('/tmp/ico71.obj',V,F)
PY
(DG_HOST_DEBUG=1 tests/cpp/build/unchanged_caller addfunction /tmp/ico71.obj "256 256 256" 5) > $O/addfn_debug2.log 2>&1
(DG_LAZY_HOST=0 tests/cpp/build/unchanged_caller addfunction /tmp/ico71.obj "256 256 256" 4) > $O/addfn_eager2.log 2>&1
(DG_HOST_DIRECT_FRACTIONS="1" tests/cpp/build/unchanged_caller addfunction /tmp/ico71.obj "256 256 256" 4) > $O/addfn_one.log 2>&1
bash tools/gpu_k3_pmc.sh base 128
(python tools/k3_run.py --res 128 --steps 3 --check; python tools/k3_run.py --res 256 --steps 2) > $O/k3_base.log 2>&1
(timeout 600 python bench.py 2>&1 | tail -2) > $O/b3.log
