#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / mean / min / max.

Usage: python tools/rocpd_summary.py <results.db> [--csv out.csv]
Used to produce the summaries committed under profiles/ from gpurun_out/*.db.
"""
import sqlite3
import sys


def main():
    path = sys.argv[1]
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(
        f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
        f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["kernel,calls,total_ns,avg_ns,min_ns,max_ns,pct"]
    for n, c, t, a, mn, mx in rows:
        n = n.replace(",", ";")
        lines.append(f"{n},{c},{t},{a:.1f},{mn},{mx},{100.0 * t / total:.2f}")
    out = "\n".join(lines)
    print(out)
    if "--csv" in sys.argv:
        open(sys.argv[sys.argv.index("--csv") + 1], "w").write(out + "\n")


if __name__ == "__main__":
    main()
