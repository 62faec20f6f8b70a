"""Kernel statistics from a rocprofv3 rocpd database (the default output format of this ROCm): name, calls, total, average.
Usage: python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--csv out.csv]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(db.execute(
        f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
        f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc"))
    total = sum(r[2] for r in rows) or 1
    lines = ['"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"']
    for r in rows:
        lines.append('"%s",%d,%d,%.1f,%.2f,%d,%d' % (r[0], r[1], r[2], r[3], 100.0 * r[2] / total, r[4], r[5]))
    if len(sys.argv) > 3 and sys.argv[2] == "--csv":
        open(sys.argv[3], "w").write("\n".join(lines) + "\n")
    for ln in lines[:40]:
        print(ln[:200])


if __name__ == "__main__":
    main()
