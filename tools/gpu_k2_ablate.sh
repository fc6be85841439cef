#!/bin/bash
# gpurun helper: the staged K2 gather with one phase removed at a time, per-kernel times.  Build the variants first (no GPU needed):
#   python -c "from discregrid_amd.build import build; import os; [build(defines=('-DDG_K2_ABLATE=%d' % b,), out=os.path.abspath('discregrid_amd/variants/libdg_ablate%d.so' % b)) for b in (1, 8, 9, 15)]"
# (libdg_ablate*.so under discregrid_amd/variants/ are git-ignored and travel with the snapshot)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/${1:-ablate}; mkdir -p $O
python -c "import torch" 2>/dev/null
for lib in default discregrid_amd/variants/libdg_ablate*.so; do
  if [ "$lib" = default ]; then unset DG_LIB; tag=default; else export DG_LIB=$PWD/$lib; tag=$(basename $lib .so); fi
  rm -rf /tmp/k2trace_$tag
  timeout 90 rocprofv3 --kernel-trace --stats -d /tmp/k2trace_$tag -o t -- python tests/perf/k2_plain_ab.py --once --variants "k2_tiles=1" > /dev/null 2> $O/err_$tag.txt || tail -3 $O/err_$tag.txt
  echo "== $tag"
  python - "$tag" <<'PY'
import glob, sqlite3, sys
for f in glob.glob("/tmp/k2trace_%s/**/*_results.db" % sys.argv[1], recursive=True):
    c = sqlite3.connect(f)
    for name, dur, grid in c.execute("select name,duration,grid_x from kernels order by start"):
        if "k_interpolate_tiles" in name:
            print("  %-50s grid %9d  %9.1f us" % (name.replace("dg::(anonymous namespace)::", "")[:50], grid, dur / 1e3))
PY
done | tee $O/ablate.txt
