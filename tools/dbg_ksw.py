import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import minimap2_amd as mm, reflib
from seqsim import random_pair
rng = np.random.default_rng(1)
mat = reflib.ts_mat(2, 4)
def run(jobs):
    got = mm.ksw_extd2_batch(jobs, mat, 4, 2, 24, 1)
    for j, g in zip(jobs, got):
        w = reflib.ora_extd2(j[0], j[1], mat, 4, 2, 24, 1, j[2], j[3], j[4], j[5])
        print(len(j[0]), len(j[1]), "OK" if g == w else ("DIFF score %d vs %d cigar_eq=%s" % (g[8], w[8], g[10] == w[10])))
t = rng.integers(0, 4, 8, dtype=np.uint8)
print("single identical 8x8"); run([(t.copy(), t, 30001, 400, -1, 0x08)])
print("pair identical"); run([(t.copy(), t, 30001, 400, -1, 0x08)] * 2)
q, t2 = random_pair(rng, 40, 0.1)
print("single 40"); run([(q, t2, 30001, 400, -1, 0x08)])
print("pair 8 + 40"); run([(t.copy(), t, 30001, 400, -1, 0x08), (q, t2, 30001, 400, -1, 0x08)])
t3 = rng.integers(0, 4, 100, dtype=np.uint8)
print("single 100 identical"); run([(t3.copy(), t3, 30001, 400, -1, 0x08)])
