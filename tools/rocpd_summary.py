"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into a per-kernel stats table,
equivalent to `--stats` CSV output.  Usage: python tools/rocpd_summary.py results.db > out.txt"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % kd)]
kcols = [r[1] for r in cur.execute("pragma table_info(%s)" % ks)]
namecol = "kernel_name" if "kernel_name" in kcols else ("display_name" if "display_name" in kcols else kcols[-1])
rows = cur.execute("select s.%s, d.end - d.start from %s d join %s s on d.kernel_id = s.id" % (namecol, kd, ks)).fetchall()
agg = {}
for name, dur in rows:
    name = re.sub(r"\(.*", "", name)
    a = agg.setdefault(name, [0, 0, 1 << 62, 0])
    a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
tot = sum(a[1] for a in agg.values())
print("%-64s %8s %14s %12s %10s %10s %7s" % ("Name", "Calls", "TotalNs", "AvgNs", "MinNs", "MaxNs", "Pct"))
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-64s %8d %14d %12.0f %10d %10d %6.2f%%" % (name[:64], a[0], a[1], a[1] / a[0], a[2], a[3], 100.0 * a[1] / tot))
print("TOTAL kernel ns: %d over %d dispatches" % (tot, len(rows)))
