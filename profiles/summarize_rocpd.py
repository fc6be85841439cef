"""Turns a rocprofv3 rocpd SQLite result (gpurun_out/<dir>/*_results.db) into the small text
summaries committed under profiles/.  Usage: python profiles/summarize_rocpd.py <db> <out.txt> [note]"""
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    c = sqlite3.connect(db)
    lines = []
    if note:
        lines.append("# " + note)
    lines.append("# source: %s (rocprofv3 --kernel-trace --stats)" % db)
    lines.append("%-90s %8s %16s %16s %8s" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
    for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        lines.append("%-90s %8d %16.0f %16.0f %8.3f" % (name[:90], calls, total * 1e3, avg * 1e3, pct))
    lines.append("")
    lines.append("# per-dispatch (first 12): name, duration_ns, grid, workgroup, vgpr, sgpr, lds, scratch")
    q = ("select name,duration,grid_x,workgroup_x,vgpr_count,sgpr_count,lds_size,scratch_size from kernels "
         "order by start limit 12")
    for r in c.execute(q):
        lines.append("%-60s %12d grid=%d wg=%d vgpr=%d sgpr=%d lds=%d scratch=%d" % ((r[0][:60],) + tuple(r[1:])))
    try:
        rows = list(c.execute("select * from counters_collection limit 1"))
        if rows:
            cur = c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                            "group by kernel_name, counter_name")
            lines.append("")
            lines.append("# PMC counters (mean per dispatch): kernel, counter, value, n")
            for r in cur:
                lines.append("%-60s %-28s %20.1f %6d" % (str(r[0])[:60], r[1], r[2], r[3]))
    except sqlite3.Error as e:
        lines.append("# (no counters: %s)" % e)
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
