#!/bin/bash
# Same-box A/B of whole library builds: the in-tree library against every discregrid_amd/variants/*.so
# (DG_LIB selects the library the Python binding loads).  usage (through gpurun): bash tools/gpu_lib_ab.sh [rounds]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out
ROUNDS=${1:-3}
for r in $(seq 1 $ROUNDS); do
  for lib in default discregrid_amd/variants/*.so; do
    if [ "$lib" = default ]; then unset DG_LIB; else export DG_LIB=$PWD/$lib; fi
    for fast in 1 0; do
      DG_FORCE="k1_fast=$fast" timeout 300 python bench.py --steps 20 --warmup 5 --no-extras > gpurun_out/lab.json 2> gpurun_out/lab.err
      python - <<PY
import json
d=json.load(open("gpurun_out/lab.json"))
print("round $r  %-44s k1_fast=$fast  %8.1f Mnodes/s  %7.3f ms" % ("$lib", d["value"], d["ms_per_step"]))
PY
    done
  done
done | tee gpurun_out/lib_ab.txt
