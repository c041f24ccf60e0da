"""Timeline of one fit cycle from a rocprofv3 kernel-trace database: every dispatch between two consecutive launches of the anchor
kernel (default: the persistent factorisation), with its queue, start offset and duration -- shows what is on the critical path.
usage: python tools/cycle_timeline.py gpurun_out/pc2/t_results.db [anchor-substring] [which-cycle]"""
import sqlite3
import sys

db = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "potrf_mega_kernel"
which = int(sys.argv[3]) if len(sys.argv) > 3 else -2
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
scols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
namecol = "display_name" if "display_name" in scols else "kernel_name"
names = {r[0]: r[1] for r in c.execute("select id, %s from %s" % (namecol, ks))}
rows = list(c.execute("select kernel_id, queue_id, start, end, grid_size_x, workgroup_size_x from %s order by start" % kd))
idx = [i for i, r in enumerate(rows) if anchor in names.get(r[0], "")]
i0, i1 = idx[which - 1], idx[which]
t0 = rows[i0][3]
print("cycle between the ends of two %s launches: %.1f us" % (anchor, (rows[i1][3] - t0) / 1e3))
queues = sorted({r[1] for r in rows[i0 + 1:i1 + 1]})
prev_end = {q: t0 for q in queues}
for r in rows[i0 + 1:i1 + 1]:
    nm = names.get(r[0], "?").split("(")[0].replace("mrbf::", "").replace("void ", "")[:46]
    q = queues.index(r[1])
    print("q%d %s start %8.1f  dur %7.1f  gap %6.1f  grid %d" % (q, nm.ljust(46), (r[2] - t0) / 1e3, (r[3] - r[2]) / 1e3, (r[2] - prev_end[r[1]]) / 1e3, r[4] // max(r[5], 1)))
    prev_end[r[1]] = r[3]
