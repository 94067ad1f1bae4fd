"""Time the lane-exact kernel on ONT-like extension jobs (flags 0x40 / 0xC2, band 751) and on gap fills just beyond the
register-resident kernel's width, through the C ABI; prints the HIP-event kernel times from the library's profiler."""
import sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from seqsim import random_pair
from reflib import ts_mat
import minimap2_amd as mm

rng = np.random.default_rng(1)
mat = ts_mat(2, 4)


def run(name, jobs):
    mm.ksw_extd2_batch(jobs[:500], mat, 4, 2, 24, 1)
    mm.profile_enable(True)
    mm.ksw_extd2_batch(jobs, mat, 4, 2, 24, 1)
    prof = mm.profile_get()
    mm.profile_enable(False)
    cells = sum(len(q) * min(len(t), 2 * w + 1 if w >= 0 else len(t)) for q, t, w, *_ in jobs)
    ms = sum(v["ms"] for v in prof.values())
    print("%s: %d jobs, %.3g cells, kernels %.2f ms %s -> %.1f GCUPS, %.1f us/job/5120 waves" % (name, len(jobs), cells, ms, {k: round(v["ms"], 2) for k, v in prof.items()}, cells / ms / 1e6, ms * 1e3 / (len(jobs) / 5120.0)))


n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
base = [random_pair(rng, int(np.clip(rng.normal(70, 40), 10, 300)), 0.12) for _ in range(500)]
run("ext 0x40 small", [(base[i % 500][0], base[i % 500][1], 751, 400, 10, 0x40) for i in range(n)])
run("ext 0xC2 small", [(base[i % 500][0][::-1].copy(), base[i % 500][1][::-1].copy(), 751, 400, 10, 0xC2) for i in range(n)])
run("gapfill 0x08 small (fast kernel)", [(base[i % 500][0], base[i % 500][1], 30001, 400, -1, 0x08) for i in range(n)])
big = [random_pair(rng, int(rng.integers(520, 700)), 0.12) for _ in range(200)]
run("gapfill 0x08 t>512 (exact)", [(big[i % 200][0], big[i % 200][1], 30001, 400, -1, 0x08) for i in range(n // 10)])
