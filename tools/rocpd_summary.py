"""Summarise rocprofv3 (rocpd sqlite) outputs: per-kernel time stats and PMC counters."""
import glob
import sqlite3
import sys


def kernel_stats(db):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
        "from kernels group by name order by 3 desc"
    ).fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# {db}")
    print("kernel,calls,total_ns,avg_ns,min_ns,max_ns,percent")
    for r in rows:
        print(f"{r[0]},{r[1]},{r[2]},{r[3]:.0f},{r[4]},{r[5]},{100*r[2]/total:.2f}")


def pmc(db):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    print(f"# {db}")
    try:
        rows = cur.execute(
            "select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection "
            "group by kernel_name, counter_name order by kernel_name, counter_name"
        ).fetchall()
    except sqlite3.OperationalError:
        print("columns:", cols)
        return
    print("kernel,counter,dispatches,avg_per_dispatch,sum")
    for r in rows:
        print(f"{r[0]},{r[1]},{r[2]},{r[3]:.1f},{r[4]:.1f}")


if __name__ == "__main__":
    for db in sys.argv[1:]:
        if "stats" in db:
            kernel_stats(db)
        else:
            pmc(db)
