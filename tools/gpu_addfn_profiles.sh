export TMPDIR=/tmp
python - > /tmp/ico71.log 2>&1 <<PY
import sys; sys.path.insert(0, "tests"); import dgtest as T
V, F = T.icosphere(71); T.write_obj("/tmp/ico71.obj", V, F)
PY
for f in "$@"; do echo "== $f"; DG_FORCE="host_debug=1;$f" timeout 120 tests/cpp/build/unchanged_caller addfunction /tmp/ico71.obj "256 256 256" 4 2>&1 | grep "dg_sdf_sample_field\|host copy job: direct" | tail -4 | cut -c1-200; done
