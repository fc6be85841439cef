#!/usr/bin/env python3
"""Mean counter value per kernel from the rocprofv3 databases under a directory (one --pmc pass each).
    python tools/pmc_dump.py <dir> [kernel-substring]"""
import glob
import os
import sqlite3
import sys

d = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
for db in sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)):
    c = sqlite3.connect(db)
    try:
        rows = list(c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"))
    except sqlite3.Error:
        try:
            rows = [(n, "duration_ms", a * 1e3, k) for n, k, a in c.execute("select name,total_calls,average from top_kernels")]
        except sqlite3.Error:
            continue
    for k, cn, v, n in rows:
        k = str(k)
        if want in k:
            print("%-28s %-44s %-40s %16.6g %4d" % (os.path.basename(db)[:28], k[:44], cn, v, n))
