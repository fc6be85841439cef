#!/usr/bin/env python3
"""Memory copies of a rocprofv3 --memory-copy-trace run, grouped by direction: count, bytes, time.
    python tools/memcopy_summary.py <dir-with-the-.db>"""
import glob
import os
import sqlite3
import sys

for db in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)):
    c = sqlite3.connect(db)
    names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    cand = [n for n in names if "memory_cop" in n.lower() or "memcpy" in n.lower() or n.lower() == "memory_copies"]
    print("#", os.path.basename(db), "tables/views with copies:", cand)
    for t in cand:
        cols = [r[1] for r in c.execute("pragma table_info(%s)" % t)]
        print("#  %s columns: %s" % (t, cols))
        name_col = next((x for x in ("name", "kind", "direction", "copy_kind") if x in cols), None)
        size_col = next((x for x in ("size", "bytes", "copy_bytes") if x in cols), None)
        dur = "duration" if "duration" in cols else ("(end - start)" if "end" in cols and "start" in cols else None)
        if name_col and size_col:
            q = "select %s, count(*), sum(%s)%s from %s group by %s" % (name_col, size_col, (", sum(%s)" % dur) if dur else "", t, name_col)
            for row in c.execute(q):
                gb = (row[2] or 0) / 1e9
                ms = (row[3] or 0) / 1e6 if dur else float("nan")
                print("%-44s copies %6d  bytes %10.4f GB  time %9.2f ms" % (str(row[0])[:44], row[1], gb, ms))
            break
