#!/bin/bash
# Same-box A/B of K1 across library builds: the in-tree library and every discregrid_amd/variants/*.so, interleaved, ROUNDS times.
#   bash tools/gpu_k1_libs.sh [rounds] [meshes]      (through gpurun; output gpurun_out/k1_libs.txt)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out
ROUNDS=${1:-3}; MESHES=${2:-ico256,bunny256,dragon256}
python -c "import torch" 2>/dev/null
for r in $(seq 1 $ROUNDS); do
  for lib in default discregrid_amd/variants/*.so; do
    if [ "$lib" = default ]; then unset DG_LIB; else export DG_LIB=$PWD/$lib; fi
    timeout 300 python tools/k1_time.py --meshes $MESHES 2> gpurun_out/k1_libs.err || tail -3 gpurun_out/k1_libs.err
  done
done | tee gpurun_out/k1_libs.txt
# optional third argument "trace": per-kernel durations of every library on the first mesh (rocprofv3 --kernel-trace --stats)
if [ "${3:-}" = trace ]; then
  for lib in default discregrid_amd/variants/*.so; do
    if [ "$lib" = default ]; then unset DG_LIB; tag=default; else export DG_LIB=$PWD/$lib; tag=$(basename $lib .so); fi
    rm -rf /tmp/k1trace_$tag
    timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/k1trace_$tag -o t -- python tools/k1_time.py --meshes ${MESHES%%,*} > /dev/null 2> gpurun_out/k1_libs_trace.err || tail -3 gpurun_out/k1_libs_trace.err
    echo "== $tag"
    python - "$tag" <<'PY'
import glob, sqlite3, sys
for f in glob.glob("/tmp/k1trace_%s/**/*_results.db" % sys.argv[1], recursive=True):
    c = sqlite3.connect(f)
    for name, calls, avg in c.execute("select name,total_calls,average from top_kernels"):
        if "k_sample" in name or "k_heavy" in name:
            print("  %-60s calls %4d  avg %10.1f us" % (name[:60], calls, avg))
PY
  done | tee gpurun_out/k1_libs_trace.txt
fi
