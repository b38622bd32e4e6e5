#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 `--kernel-trace` run (rocpd SQLite output): calls, average, total.
   python tools/rocpd_stats.py <results.db> [--csv out.csv] [--top N]"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--csv")
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    cur = sqlite3.connect(a.db).cursor()
    rows = cur.execute("select name, count(*), avg(duration), sum(duration), min(duration), max(duration), max(vgpr_count), max(scratch_size) "
                       "from kernels group by name order by sum(duration) desc").fetchall()
    lines = ["Name,Calls,AverageNs,TotalDurationNs,MinNs,MaxNs,VGPRs,ScratchBytes"]
    for r in rows:
        lines.append('"%s",%d,%.1f,%d,%d,%d,%s,%s' % (r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]))
    if a.csv:
        open(a.csv, "w").write("\n".join(lines) + "\n")
    for r in rows[:a.top]:
        print(f"{r[0][:120]:120s} calls {r[1]:5d} avg {r[2] / 1e6:9.3f} ms total {r[3] / 1e6:10.1f} ms vgpr {r[6]} scratch {r[7]}")


if __name__ == "__main__":
    main()
