"""Text Gantt chart of an MM2AMD_TRACE file: one row per lane, one character per --res milliseconds over the last --win seconds.
  S sketch+collect (GPU)   E expand..backtrack (GPU)   a D2H of chains   v chains->vectors (host)   P host:pre   p host:plan   o ksw-order (host)
  K DP kernels (GPU)       g D2H of CIGARs             u ksw-unperm      c host:consume              f host:finish   . nothing open
and a last row: how many lanes have a GPU stage open (0 = the GPU has nothing queued by the mapper).
    python tools/trace_ascii.py trace.tsv [--win 1.3] [--res 4]
Measurement scaffolding."""
import sys

SYM = {"gpu:sketch+collect": "S", "gpu:expand..backtrack": "E", "gpu:expand+sort, d2h:anchors": "E", "d2h:chains": "a", "host:chains->vectors": "v", "host:pre": "P",
       "host:plan": "p", "host:ksw-order": "o", "gpu:ksw": "K", "d2h:cigar": "g", "host:ksw-unperm": "u", "host:consume": "c", "host:finish": "f"}
GPU = set("SEK")


def main():
    a = sys.argv
    win = float(a[a.index("--win") + 1]) if "--win" in a else 1.3
    res = float(a[a.index("--res") + 1]) if "--res" in a else 4.0
    recs = [l.rstrip("\n").split("\t") for l in open(a[1])]
    recs = [(int(x[0]), x[1], float(x[2]), float(x[3])) for x in recs]
    t1 = max(r[3] for r in recs)
    t0 = t1 - win
    n = int(win * 1e3 / res)
    lanes = sorted({r[0] for r in recs})
    rows = {l: ["."] * n for l in lanes}
    for l, st, b, e in recs:
        if e <= t0:
            continue
        s = SYM.get(st, "?")
        i0, i1 = max(0, int((b - t0) * 1e3 / res)), min(n - 1, int((e - t0) * 1e3 / res))
        for i in range(i0, i1 + 1):
            rows[l][i] = s
    for l in lanes:
        print("lane %d |%s|" % (l, "".join(rows[l])))
    cnt = [sum(1 for l in lanes if rows[l][i] in GPU) for i in range(n)]
    print("GPU st |%s|" % "".join(str(min(c, 9)) for c in cnt))
    print("columns with no lane in a GPU stage: %.1f %%; with one: %.1f %%" % (100.0 * sum(1 for c in cnt if c == 0) / n, 100.0 * sum(1 for c in cnt if c == 1) / n))
    tot = {}
    for l, st, b, e in recs:
        if e > t0:
            tot[st] = tot.get(st, 0.0) + e - max(b, t0)
    print("stage seconds over lanes in the window:", {k: round(v, 3) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])})


if __name__ == "__main__":
    main()
