#!/usr/bin/env python3
"""Per-launch timeline of the last round-4 call in a rocprofv3 kernel-trace database (start, duration, grid, queue)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = list(cur.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.queue_id from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, 60)
n = len(rows); rows = rows[2 * n // 3:]
t0 = rows[0][1]
import re
for r in rows[lo:hi]:
    name = re.sub(r"^_ZN4mrbf2r4\d+", "", r[0])[:34]
    print("%-34s start %8.1f dur %7.1f grid %6dx%-3d q%s" % (name, (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], r[4], r[5]))
