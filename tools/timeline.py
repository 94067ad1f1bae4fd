"""GPU timeline summary from a rocprofv3 rocpd database: busy fraction (union of kernel intervals) over the last T seconds
of kernel activity, per-kernel time inside that window, and the memory-copy volume/time when a copy trace is present.
    python tools/timeline.py results.db [window_seconds]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
rows = db.execute("select start, end, name from kernels order by start").fetchall()
t_end = max(r[1] for r in rows)
t0 = t_end - int(win * 1e9)
rows = [r for r in rows if r[1] > t0]
busy, cur_s, cur_e = 0, None, None
per = {}
for s, e, nm in rows:
    s = max(s, t0)
    per[nm] = per.get(nm, 0) + (e - s)
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("window %.3f s: GPU busy (union of kernels) %.3f s = %.1f %%; sum of kernel durations %.3f s" % (win, busy / 1e9, 100.0 * busy / (win * 1e9), sum(per.values()) / 1e9))
for nm, v in sorted(per.items(), key=lambda kv: -kv[1])[:14]:
    print("  %-80s %8.3f s" % (nm[:80], v / 1e9))
try:
    cols = [c[1] for c in db.execute("pragma table_info('memory_copies')")]
    mc = db.execute("select start, end, size, name from memory_copies where end > ?", (t0,)).fetchall()
    tot = {}
    for s, e, sz, nm in mc:
        k = nm
        a = tot.setdefault(k, [0, 0, 0])
        a[0] += 1; a[1] += sz; a[2] += e - max(s, t0)
    for k, a in tot.items():
        print("  copies %-40s n=%6d  %.3f GB  %.3f s" % (k, a[0], a[1] / 1e9, a[2] / 1e9))
except Exception as ex:
    print("no copy trace:", ex)
