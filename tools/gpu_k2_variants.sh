#!/bin/bash
# K2 A/B over library variants built into discregrid_amd/variants/ (DG_LIB selects the library)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out; OUT=gpurun_out/k2_variants.txt; : > $OUT
for v in ${VARIANTS:-rm4 mo4 mo2 mo1 mo8}; do
  DG_LIB=$PWD/discregrid_amd/variants/libdg_$v.so timeout 300 python tests/perf/bench_interpolate.py --cpu-seconds ${CPUSEC:-0} --steps 5 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    d = json.loads(line)
    print('$v %-18s %-22s grad=%d %8.1f Mq/s %.3f ms hbm_alg_frac %.3f %s' % (d['layout'], d['distribution'], d['gradient'], d['value'], d['ms'], d['roofline']['frac'], d.get('tile_major_build_ms', '')))
" >> $OUT
done
cat $OUT
