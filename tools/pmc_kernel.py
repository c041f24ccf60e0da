"""Per-kernel sums of the PMC counters in a rocprofv3 database: python tools/pmc_kernel.py <results.db> <kernel-substring>"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
pat = sys.argv[2]
rows = con.execute("select name, counter_name, count(*), sum(counter_value) from pmc_events where name like ? group by name, counter_name", ("%" + pat + "%",)).fetchall()
for r in rows:
    print("%-60s %-28s launches %3d  sum %.4g  per launch %.4g" % (r[0][:60], r[1], r[2], r[3], r[3] / r[2]))
