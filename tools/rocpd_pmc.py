#!/usr/bin/env python3
"""Per-kernel PMC averages from a rocprofv3 (rocpd sqlite) counter-collection run.
Usage: python tools/rocpd_pmc.py <results.db> [kernel-substring]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    q = ("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection "
         "where kernel_name like ? group by kernel_name, counter_name order by kernel_name, counter_name")
    try:
        rows = cur.execute(q, (f"%{sub}%",)).fetchall()
    except sqlite3.OperationalError:
        print("columns:", cols)
        raise
    print("kernel,counter,dispatches,avg_per_dispatch,sum")
    for k, c, n, a, s in rows:
        print(f"{k.split('(')[0][:60].replace(',', ';')},{c},{n},{a:.1f},{s:.1f}")


if __name__ == "__main__":
    main()
