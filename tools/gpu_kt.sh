#!/bin/bash
# kernel trace of one profiles/pmc_workloads.py workload: per-kernel mean durations
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
W=${1:-k2}
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_kt_$W -o kt -- python $GRAFT_REPO_ROOT/profiles/pmc_workloads.py $W > /tmp/kt.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY | tee gpurun_out/kt_$W.txt
import sqlite3, glob
db = glob.glob("/tmp/prof_kt_$W/**/kt_results.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = list(c.execute("select name,duration from kernels order by start"))
import collections
agg = collections.OrderedDict()
for n, d in rows:
    import re
    m = re.search(r"(k_\w+(<[^>]*>)?)", n) or re.search(r"(onesweep\w*|radix_sort\w*|\w+_kernel\b)", n)
    k = m.group(1) if m else n[:60]
    agg.setdefault(k, []).append(d)
for k, v in agg.items():
    print("%-72s n=%3d mean %.4f ms min %.4f max %.4f" % (k, len(v), sum(v)/len(v)/1e6, min(v)/1e6, max(v)/1e6))
PY
