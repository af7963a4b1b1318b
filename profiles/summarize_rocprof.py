#!/usr/bin/env python3
"""Turn rocprofv3 rocpd databases (gpurun_out/prof_*/..._results.db) into the small text summaries kept under
profiles/.  usage: summarize_rocprof.py stats <db> | pmc <db>"""
import csv
import sqlite3
import sys


def short(name):
    return name.split("(")[0].replace("void ", "")[:70]


def main():
    mode, db = sys.argv[1], sys.argv[2]
    cur = sqlite3.connect(db).cursor()
    if mode == "stats":
        w = csv.writer(sys.stdout, lineterminator="\n")  # kernel names contain commas: quoted
        w.writerow("kernel,calls,total_us,avg_us,min_us,max_us,percent".split(","))
        rows = cur.execute(
            "select name, count(*), sum(duration)/1e3, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3 "
            "from kernels group by name order by sum(duration) desc").fetchall()
        tot = sum(r[2] for r in rows)
        for r in rows:
            w.writerow([short(r[0]), r[1], f"{r[2]:.1f}", f"{r[3]:.2f}", f"{r[4]:.2f}", f"{r[5]:.2f}", f"{100 * r[2] / tot:.2f}"])
    else:
        w = csv.writer(sys.stdout, lineterminator="\n")
        w.writerow("kernel,counter,dispatches,avg_value,min_value,max_value".split(","))
        for r in cur.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) "
                             "from counters_collection group by kernel_name, counter_name order by 1, 2"):
            w.writerow([short(r[0]), r[1], r[2], f"{r[3]:.3f}", f"{r[4]:.3f}", f"{r[5]:.3f}"])


if __name__ == "__main__":
    main()
