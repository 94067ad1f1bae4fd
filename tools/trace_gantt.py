"""Per-lane timeline of one mm_gpu_map_batch call from an MM2AMD_TRACE file: the calls are separated by the moments no lane has a
stage open; prints call K (default: the middle one) as  lane | stage | start ms | duration ms, and per-stage sums per lane.

    python tools/trace_gantt.py trace.tsv [K]

Measurement scaffolding."""
import sys
from collections import defaultdict

recs = [l.rstrip("\n").split("\t") for l in open(sys.argv[1])]
recs = sorted((float(c), float(d), int(a), b) for a, b, c, d in recs)
calls, cur, cur_end = [], [], None
for s, e, lane, st in recs:
    if cur_end is not None and s > cur_end + 0.004:
        calls.append(cur)
        cur = []
    cur.append((s, e, lane, st))
    cur_end = e if cur_end is None or e > cur_end else cur_end
calls.append(cur)
calls = [c for c in calls if len(c) > 20]
print("%d calls: %s" % (len(calls), " ".join("%.0f" % ((max(e for _, e, _, _ in c) - c[0][0]) * 1e3) for c in calls)))
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(calls) // 2
c = calls[k]
t0 = c[0][0]
print("call %d: %.1f ms" % (k, (max(e for _, e, _, _ in c) - t0) * 1e3))
tot = defaultdict(float)
for s, e, lane, st in c:
    tot[(lane, st)] += e - s
    print("%d  %-24s %8.1f %8.1f" % (lane, st, (s - t0) * 1e3, (e - s) * 1e3))
print("per lane and stage (ms):")
for (lane, st), v in sorted(tot.items()):
    print("  %d  %-24s %8.1f" % (lane, st, v * 1e3))
