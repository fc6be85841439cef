#!/bin/bash
# End-to-end wall time of the command-line tools on the GPU box (icosphere nu=71, 100 820 triangles).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python - <<'PY'
import sys
sys.path.insert(0,'tests')
import dgtest as T
V,F=T.icosphere(71); T.write_obj('/tmp/ico71.obj',V,F)
PY
B=discregrid_amd/cpp/build
t() { local s=$(date +%s%N); "$@" > /tmp/cli.log 2>&1; local rc=$?; local e=$(date +%s%N); echo "$(( (e - s) / 1000000 )) ms (rc $rc): $*"; tr '\r' '\n' < /tmp/cli.log | grep -E "took|rror" | tail -3; }
t $B/GenerateSDF -r "128 128 128" -o /tmp/ico_128.cdf /tmp/ico71.obj
t $B/GenerateSDF -r "256 256 256" -o /tmp/ico_256.cdf /tmp/ico71.obj
ls -la /tmp/ico_128.cdf /tmp/ico_256.cdf | awk '{print "  bytes", $5, $9}'
t $B/GenerateDensityMap -s 0.1 -r 1000 -o /tmp/ico_128.cdm /tmp/ico_128.cdf
t $B/DiscreteFieldToBitmap -s 2048 -o /tmp/a.bmp /tmp/ico_128.cdf
