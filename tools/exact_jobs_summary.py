"""Which DP jobs stay in the lane-exact kernel?  From an MM2AMD_DUMP_JOBS file (round, qlen, tlen, flag & 0x1fff, w per job): the jobs that are
neither gap fills nor extensions the register-resident kernels take (ksw_host.cpp: fast_eligible / ext_eligible), grouped by why, with
their DP cells (band-limited) and anti-diagonal counts.   python tools/exact_jobs_summary.py jobs.tsv
Measurement scaffolding."""
import sys
from collections import defaultdict
APPROX_MAX, EXTZ_ONLY, RIGHT, REV_CIGAR = 0x08, 0x40, 0x02, 0x80  # ksw2.h flags as KswJob carries them (KSW_EZ_*)
tot = defaultdict(lambda: [0, 0.0, 0.0])
n_all, cells_all = 0, 0.0
for line in open(sys.argv[1]):
    f = line.split()
    if len(f) != 5:
        continue  # (several lanes append to the file: the odd torn line)
    try:
        rnd, q, t, flag, w = int(f[0]), int(f[1]), int(f[2]), int(f[3], 16), int(f[4])
    except ValueError:
        continue
    n_all += 1
    cells_all += q * t
    nobind = w < 0 or (w + 1 >= q and w + 1 >= t)
    if flag == APPROX_MAX and nobind and q <= 1024 and t <= 3072:
        continue  # gap-fill kernels
    ext = flag in (EXTZ_ONLY, EXTZ_ONLY | RIGHT | REV_CIGAR)
    if ext and nobind and q <= 512 and t <= 512:
        continue  # ksw_ext_kernel
    width = min(q, t, w + 2 if w >= 0 else 1 << 30)
    cells = (q + t) * width
    if ext:
        why = "extension, band binds (q+t > w)" if not nobind else "extension, q or t > 512"
    elif flag == APPROX_MAX:
        why = "gap fill, band binds" if not nobind else "gap fill, q > 1024 or t > 3072"
    elif flag == 0:
        why = "second pass (exact score, flag 0)"
    else:
        why = "other flag 0x%x" % flag
    key = (why, "round %d" % min(rnd, 2), "width<=%d" % (64 if width <= 64 else 192 if width <= 192 else 448 if width <= 448 else 960 if width <= 960 else 99999))
    a = tot[key]
    a[0] += 1
    a[1] += cells
    a[2] += q + t
print("%d jobs, %.3g cells in all; lane-exact:" % (n_all, cells_all))
for k, a in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("  %-42s %-8s %-12s %8d jobs  %10.3g cells  mean rows %6.0f" % (k[0], k[1], k[2], a[0], a[1], a[2] / a[0]))
