#!/usr/bin/env python3
"""Turn rocprofv3 rocpd databases (gpurun_out/prof_*/..._results.db) into the small text summaries kept under
profiles/.  usage: summarize_rocprof.py stats <db> | pmc <db>"""
import sqlite3
import sys


def short(name):
    return name.split("(")[0].replace("void ", "")[:70]


def main():
    mode, db = sys.argv[1], sys.argv[2]
    cur = sqlite3.connect(db).cursor()
    if mode == "stats":
        print("kernel,calls,total_us,avg_us,min_us,max_us,percent")
        rows = cur.execute(
            "select name, count(*), sum(duration)/1e3, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3 "
            "from kernels group by name order by sum(duration) desc").fetchall()
        tot = sum(r[2] for r in rows)
        for r in rows:
            print(f"{short(r[0])},{r[1]},{r[2]:.1f},{r[3]:.2f},{r[4]:.2f},{r[5]:.2f},{100 * r[2] / tot:.2f}")
    else:
        print("kernel,counter,dispatches,avg_value,min_value,max_value")
        for r in cur.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) "
                             "from counters_collection group by kernel_name, counter_name order by 1, 2"):
            print(f"{short(r[0])},{r[1]},{r[2]},{r[3]:.3f},{r[4]:.3f},{r[5]:.3f}")


if __name__ == "__main__":
    main()
