#!/bin/bash
# gpurun helper (round 6): parity of the staged K2 path, the same-box A/B (K2_VARIANTS="a|b|..." overrides the variants), a kernel trace of
# one call per variant (-> $O/k2_kernel_stats.txt, $O/k2_dispatches.txt)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/${1:-r06b}
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -x -k "staged_tiles or binned_path" > $O/parity.txt 2>&1; tail -5 $O/parity.txt
python tests/perf/k2_plain_ab.py --rounds 3 ${K2_VARIANTS:+--variants "$K2_VARIANTS"} > $O/k2_ab.jsonl 2> $O/k2_ab.err; cat $O/k2_ab.jsonl | cut -c1-600
rocprofv3 --kernel-trace --stats -d $O/trace -o k2 -- python tests/perf/k2_plain_ab.py --once --variants "k2_tiles=0|k2_tiles=1" > $O/trace.log 2>&1
DB=$(find $O/trace -name "*_results.db" | head -1)
python profiles/summarize_rocpd.py "$DB" $O/k2_kernel_stats.txt "k2_plain_ab.py --once: one call per case of k2_tiles=0 and k2_tiles=1" > /dev/null 2>&1 || ls -R $O/trace | head
grep -E "k_tile|k_interpolate|k_bin|rocprim|Memset|fill" $O/k2_kernel_stats.txt | cut -c1-160 | head -30
python - "$DB" > $O/k2_dispatches.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for r in c.execute("select name,duration,grid_x,workgroup_x,vgpr_count,lds_size from kernels order by start"):
    n = r[0].replace("dg::(anonymous namespace)::", "").replace("void ", "")
    if n.startswith("rocprim"):
        n = "rocprim:" + n.split("::")[-1][:40] if "trampoline" not in n else "rocprim:trampoline"
    print("%-50s %10d ns grid=%d wg=%d vgpr=%d lds=%d" % (n[:50], r[1], r[2], r[3], r[4], r[5]))
PY
rm -rf $O/trace
