"""Time the batched extd2 kernel on ONT-like gap-fill jobs (median 234x235, flag 0x08, w 30001) through the C ABI."""
import sys, time
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from seqsim import random_pair
from reflib import ts_mat
import minimap2_amd as mm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = np.random.default_rng(1)
base = [random_pair(rng, int(np.clip(rng.normal(235, 40), 50, 480)), 0.12) for _ in range(500)]
jobs = [(base[i % 500][0], base[i % 500][1], 30001, 400, -1, 0x08) for i in range(n)]
cells = sum(len(q) * len(t) for q, t, *_ in jobs)
mat = ts_mat(2, 4)
mm.ksw_extd2_batch(jobs[:2000], mat, 4, 2, 24, 1)
for rep in range(3):
    t0 = time.time()
    mm.ksw_extd2_batch(jobs, mat, 4, 2, 24, 1)
    dt = time.time() - t0
    print("jobs %d cells %.3g wall %.3fs -> %.2f GCUPS (incl. packing/H2D/D2H)" % (n, cells, dt, cells / dt / 1e9), flush=True)
