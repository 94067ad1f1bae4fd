"""Summarise a rocprofv3 rocpd database (--kernel-trace --stats) as a per-kernel table: calls, total/avg/min/max ns, %.
    python tools/rocpd_summary.py gpurun_out/prof_full/bench_results.db > profiles/r01_bench_kernel_stats.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(lds_size), max(scratch_size) "
                  "from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print("%-70s %8s %14s %12s %12s %12s %6s %5s %7s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "%", "vgpr", "lds", "scratch"))
for r in rows:
    print("%-70s %8d %14d %12d %12d %12d %6.2f %5s %7s %7s" % (r[0][:70], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot, r[6], r[7], r[8]))
