"""Summarise an MM2AMD_TRACE file: per-stage totals and the union of gpu:* intervals (how much of the wall had GPU work queued)."""
import sys
from collections import defaultdict
recs = [l.rstrip("\n").split("\t") for l in open(sys.argv[1])]
recs = [(int(a), b, float(c), float(d)) for a, b, c, d in recs]
tmax = max(r[3] for r in recs)
win = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
recs = [r for r in recs if r[3] > tmax - win]
tmin = min(r[2] for r in recs)
tot = defaultdict(float)
for l, s, a, b in recs:
    tot[s] += b - a
print("window %.3f s" % (tmax - tmin))
for s, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print("  %-28s %.3f s" % (s, v))
iv = sorted((a, b) for l, s, a, b in recs if s.startswith("gpu:"))
busy, cs, ce = 0.0, None, None
for a, b in iv:
    if ce is None or a > ce:
        if ce is not None: busy += ce - cs
        cs, ce = a, b
    else: ce = max(ce, b)
busy += ce - cs
print("union of gpu:* stages: %.3f s" % busy)
