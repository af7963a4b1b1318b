#!/usr/bin/env python3
"""Start / end of the last dispatches whose kernel name matches a pattern, from a rocprofv3 rocpd database: shows whether
back-to-back kernels of one stream overlap, queue behind each other or sit idle.  usage: timeline_rocprof.py <db> <substring> [rows=24]"""
import sqlite3, sys
db, pat = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 24
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, start, end, grid_x, workgroup_x, lds_size, vgpr_count, scratch_size from kernels where name like ? order by start desc limit ?", (f"%{pat}%", n)).fetchall()[::-1]
t0 = rows[0][1]
for r in rows:
    print("%-60s start %9.1f us  end %9.1f us  dur %8.1f  grid %d wg %d lds %s vgpr %s scratch %s" % (r[0].split("(")[0][-60:], (r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], r[4], r[5], r[6], r[7]))
