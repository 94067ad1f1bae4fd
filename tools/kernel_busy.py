"""Union of the kernel intervals of a rocprofv3 --kernel-trace run: how much of the wall clock between the first and the last kernel
of the busiest stretch the GPU was executing at least one kernel, and how many ran concurrently on average.

    python tools/kernel_busy.py <dir with *_kernel_trace.csv> [--last-fraction 0.6]

Measurement scaffolding."""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    frac = float(sys.argv[sys.argv.index("--last-fraction") + 1]) if "--last-fraction" in sys.argv else 0.6
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    iv = []
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                iv.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row["Kernel_Name"]))
    iv.sort()
    if not iv:
        print("no kernel trace found under", d)
        return
    t0, t1 = iv[0][0], max(e for _, e, _ in iv)
    cut = t1 - (t1 - t0) * frac  # the timed steps are the tail of the run (index build and warm-up come first)
    iv = [x for x in iv if x[0] >= cut]
    t0 = iv[0][0]
    busy, cur_s, cur_e, total = 0, iv[0][0], iv[0][1], 0
    per = {}
    for s, e, k in iv:
        total += e - s
        k = k.split("(")[0][:60]
        per[k] = per.get(k, 0) + (e - s)
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    wall = t1 - t0
    print("window %.1f ms: GPU busy (>= 1 kernel) %.1f ms = %.3f of the wall; sum of kernel durations %.1f ms (%.2f concurrent on average while busy)"
          % (wall / 1e6, busy / 1e6, busy / wall, total / 1e6, total / max(busy, 1)))
    for k, v in sorted(per.items(), key=lambda x: -x[1])[:14]:
        print("  %-60s %9.1f ms  %.3f of wall" % (k, v / 1e6, v / wall))


if __name__ == "__main__":
    main()
