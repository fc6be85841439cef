#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python tests/perf/bench_interpolate.py --cpu-seconds ${CPUSEC:-0} --steps 5 2>/dev/null | tee gpurun_out/k2_bench.jsonl | python -c "
import sys, json
for line in sys.stdin:
    d = json.loads(line)
    print('%-18s %-22s grad=%d %8.1f Mq/s %.3f ms hbm_alg_frac %.3f %s %s' % (d['layout'], d['distribution'], d['gradient'], d['value'], d['ms'], d['roofline']['frac'], d.get('tile_major_build_ms', ''), d.get('bit_exact_vs_oracle_on_sample', '')))
"
