"""The 32-lane Z-drop walk of the gap-fill kernels (gf_zdrop_scan, minimap2_amd/csrc/ksw_gapfill_dev.hpp) as a specification:
the same decomposition -- prefix sums over the CIGAR operations, 32 consecutive segments of steps, per-segment sums and highest
prefixes, a scan that hands every segment its starting score and running maximum (the later of equals), a second walk applying
update_max_zdrop, the largest drop of the lowest lane -- restated in Python and compared with the reference's serial walk
(mm_test_zdrop / update_max_zdrop, align.c:46-84) on alignments built to produce ties: equal maxima, equal drops, long gaps, runs
shorter and longer than a segment, fewer steps than lanes.  The device code itself is checked against the oracle on the GPU
(tests/test_gpu_ksw.py); this pins the algorithm it implements."""
import numpy as np

LANES = 32
INT_MIN = -(1 << 31)


def serial(ops, tseq, qseq, mat, gq, ge):
    """update_max_zdrop over the alignment, start to end (align.c:61-84 without the inversion test)"""
    score, mx, mx_i, mx_j = 0, INT_MIN, -1, -1
    zd = [0, -1, -1, -1, -1]
    ci = cj = 0

    def track(sc, pi, pj):
        nonlocal mx, mx_i, mx_j
        if sc < mx:
            li, lj = pi - mx_i, pj - mx_j
            zz = mx - sc - abs(li - lj) * ge
            if zz > zd[0]:
                zd[:] = [zz, mx_i, pi, mx_j, pj]
        else:
            mx, mx_i, mx_j = sc, pi, pj

    for op, ln in ops:
        if op == 0:
            for l in range(ln):
                score += mat[tseq[ci + l]][qseq[cj + l]]
                track(score, ci + l, cj + l)
            ci += ln
            cj += ln
        else:
            score -= gq + ge * ln
            if op == 1:
                cj += ln
            else:
                ci += ln
            track(score, ci, cj)
    return zd


def parallel(ops, tseq, qseq, mat, gq, ge):
    n_ops = len(ops)
    # (1) first step and position of every operation
    pA, pB, S, ti, tj = [], [], 0, 0, 0
    for op, ln in ops:
        pA.append(S)
        pB.append((ti, tj))
        S += ln if op == 0 else 1
        ti += ln if op != 1 else 0
        tj += ln if op != 2 else 0
    seg = (S + LANES - 1) // LANES

    def walk(lane, body):
        lo = min(lane * seg, S)
        hi = min(lo + seg, S)
        if lo >= hi:
            return
        f = max(k for k in range(n_ops) if pA[k] <= lo)  # the binary search of the kernel
        op, ln = ops[f]
        first, cnt = pA[f], (ops[f][1] if ops[f][0] == 0 else 1)
        ci, cj = pB[f]
        for s in range(lo, hi):
            if s >= first + cnt:
                if op != 1:
                    ci += ln
                if op != 2:
                    cj += ln
                first += cnt
                f += 1
                op, ln = ops[f]
                cnt = ln if op == 0 else 1
            if op == 0:
                o = s - first
                body(mat[tseq[ci + o]][qseq[cj + o]], ci + o, cj + o)
            else:
                body(-(gq + ge * ln), ci + ln if op == 2 else ci, cj + ln if op == 1 else cj)

    # (2) per segment: sum, highest prefix (the later of equals)
    sums, rel = [0] * LANES, [(INT_MIN, -1, -1)] * LANES
    for lane in range(LANES):
        acc = [0, INT_MIN, -1, -1]

        def body(d, pi, pj, acc=acc):
            acc[0] += d
            if acc[0] >= acc[1]:
                acc[1], acc[2], acc[3] = acc[0], pi, pj
        walk(lane, body)
        sums[lane], rel[lane] = acc[0], (acc[1], acc[2], acc[3])
    start = [sum(sums[:lane]) for lane in range(LANES)]
    incl = []
    for lane in range(LANES):
        m = (INT_MIN, -1, -1) if rel[lane][0] == INT_MIN else (start[lane] + rel[lane][0], rel[lane][1], rel[lane][2])
        if lane and incl[-1][0] > m[0]:  # inclusive running maximum over the lanes: the later of equals
            m = incl[-1]
        incl.append(m)
    # (3) update_max_zdrop over every segment, starting from the maximum over the lanes before it
    best = []
    for lane in range(LANES):
        st = [start[lane]] + list(incl[lane - 1] if lane else (INT_MIN, -1, -1))
        z = [0, -1, -1, -1, -1]

        def body(d, pi, pj, st=st, z=z):
            st[0] += d
            if st[0] < st[1]:
                zz = st[1] - st[0] - abs((pi - st[2]) - (pj - st[3])) * ge
                if zz > z[0]:
                    z[:] = [zz, st[2], pi, st[3], pj]
            else:
                st[1], st[2], st[3] = st[0], pi, pj
        walk(lane, body)
        best.append(z)
    top = max(z[0] for z in best)
    return next(z for z in best if z[0] == top)  # the lowest lane holding the largest drop


def random_alignment(rng, n_ops, max_len, alphabet):
    ops, last = [], -1
    for _ in range(n_ops):
        op = int(rng.choice([0, 0, 0, 1, 2]))
        if op == last:
            op = (op + 1) % 3
        ops.append((op, int(rng.integers(1, max_len + 1))))
        last = op
    if ops[0][0] != 0:
        ops.insert(0, (0, 1))
    tl = sum(l for o, l in ops if o != 1)
    ql = sum(l for o, l in ops if o != 2)
    return ops, rng.integers(0, alphabet, tl), rng.integers(0, alphabet, ql)


def test_parallel_walk_equals_serial_walk():
    rng = np.random.default_rng(5)
    n = 0
    for case in range(3000):
        n_ops = int(rng.choice([1, 2, 3, 5, 8, 20, 60, 150]))
        max_len = int(rng.choice([1, 2, 3, 8, 40, 200]))
        alphabet = int(rng.choice([1, 2, 4]))  # one letter: every base matches, plateaus of equal maxima need gaps of cost 0 ... see scores
        a, b, gq, ge = [(1, 1, 0, 1), (2, 4, 4, 2), (1, 1, 1, 1), (1, 0, 0, 0), (2, 2, 2, 1)][case % 5]
        mat = [[a if i == j else -b for j in range(4)] for i in range(4)]
        ops, t, q = random_alignment(rng, n_ops, max_len, alphabet)
        assert parallel(ops, t, q, mat, gq, ge) == serial(ops, t, q, mat, gq, ge), (case, ops)
        n += 1
    assert n == 3000
