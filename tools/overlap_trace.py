"""Which kernels ran WHILE a given kernel was running (rocprofv3 --kernel-trace rocpd database)?
usage: python tools/overlap_trace.py results.db lstm_ring_kernel"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
kcols = [r[1] for r in cur.execute("pragma table_info(%s)" % ks)]
namecol = "kernel_name" if "kernel_name" in kcols else "display_name"
rows = cur.execute("select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (namecol, kd, ks)).fetchall()
rows = [(re.sub(r"\(.*", "", n), a, b) for n, a, b in rows]
hits = [r for r in rows if sys.argv[2] in r[0]]
print("%d launches of %s" % (len(hits), sys.argv[2]))
for n, a, b in hits[-2:]:
    inside = [(m, x, y) for m, x, y in rows if x < b and y > a and m != n]
    busy = sum(min(y, b) - max(x, a) for _, x, y in inside)
    print("%s: %.3f ms; %d other kernels overlap it for %.3f ms in total" % (n[:40], (b - a) / 1e6, len(inside), busy / 1e6))
    prev = [(m, x, y) for m, x, y in rows if y <= a][-3:]
    nxt = [(m, x, y) for m, x, y in rows if x >= b][:3]
    for m, x, y in prev: print("   before: %-50s ends %.3f ms before" % (m[:50], (a - y) / 1e6))
    for m, x, y in inside[:6]: print("   during: %-50s %.3f ms" % (m[:50], (y - x) / 1e6))
    for m, x, y in nxt: print("   after : %-50s starts %.3f ms after" % (m[:50], (x - b) / 1e6))
