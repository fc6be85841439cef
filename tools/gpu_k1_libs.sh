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
