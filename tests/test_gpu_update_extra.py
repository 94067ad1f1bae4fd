"""region_finish_kernel through the kernel-level C ABI (mm2amd_update_extra_batch) against the UNMODIFIED reference's mm_append_cigar +
mm_fix_cigar + mm_update_extra (align.c:320-334, :105-181, :254-303; oracle/ref_align_shim.c compiles align.c where it lies and calls its
static functions) on CIGARs no DP would emit -- the paths real alignments reach once in thousands of regions and the kernel must still get
exactly right: empty operations, I/D clusters (5I6D7I), leading gaps, indels in homopolymers and short tandem repeats that left-align through
whole matches (one shift feeding the next), windows whose first operation joins the last one before it, single-operation and empty windows."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import reflib  # noqa: E402

pytestmark = pytest.mark.gpu
EMU = os.environ.get("MM2AMD_EMU") == "1"


def random_region(rng, n_ops, alphabet, p_zero, p_cluster):
    """(query codes, target codes, pieces): a consistent region -- the operations cover exactly the two stretches"""
    ops = []
    while len(ops) < n_ops:
        u = rng.random()
        if u < p_cluster:  # an I/D cluster, possibly with an empty match inside
            for _ in range(int(rng.integers(2, 5))):
                ops.append((int(rng.integers(1, 3)), int(rng.integers(1, 7))))
                if rng.random() < 0.3:
                    ops.append((0, 0))
        elif u < p_cluster + p_zero:
            ops.append((int(rng.integers(0, 3)), 0))
        elif u < 0.55:
            ops.append((0, int(rng.integers(1, 30))))
        elif u < 0.75:
            ops.append((1, int(rng.integers(1, 6))))
        elif u < 0.95:
            ops.append((2, int(rng.integers(1, 6))))
        elif u < 0.97:
            ops.append((3, int(rng.integers(20, 200))))
        else:
            ops.append((0, int(rng.integers(1, 4))))
    if rng.random() < 0.5:
        ops.insert(0, (0, int(rng.integers(1, 9))))
    ops.append((0, int(rng.integers(1, 20))))  # (something the shifts cannot remove)
    tlen = sum(l for o, l in ops if o in (0, 2, 3))
    # the target: runs of one base and short tandem repeats between random stretches, so that indels CAN move
    t = np.empty(tlen + 1, dtype=np.uint8)
    p = 0
    while p < tlen:
        kind = rng.random()
        L = int(rng.integers(1, 25))
        if kind < 0.4:
            t[p:p + L] = rng.integers(0, alphabet)
        elif kind < 0.7:
            unit = rng.integers(0, alphabet, int(rng.integers(2, 4)))
            t[p:p + L] = np.tile(unit, L)[:len(t[p:p + L])]
        else:
            t[p:p + L] = rng.integers(0, alphabet, len(t[p:p + L]))
        p += L
    t = t[:tlen]
    if rng.random() < 0.3 and tlen > 0:
        t[rng.integers(0, tlen, max(1, tlen // 50))] = 4
    q, tp = [], 0
    for o, l in ops:
        if o == 0:
            seg = t[tp:tp + l].copy()
            mut = rng.random(l) < 0.1
            seg[mut] = rng.integers(0, alphabet, int(mut.sum()))
            q.append(seg)
            tp += l
        elif o == 1:  # inserted bases: mostly a continuation of what stands before them
            src = t[max(0, tp - l):tp] if rng.random() < 0.7 and tp >= l else rng.integers(0, alphabet, l).astype(np.uint8)
            q.append(np.asarray(src, dtype=np.uint8)[:l] if len(src) >= l else rng.integers(0, alphabet, l).astype(np.uint8))
        else:
            tp += l
    q = np.concatenate(q) if q else np.zeros(0, dtype=np.uint8)
    if rng.random() < 0.2 and len(q) > 0:
        q[rng.integers(0, len(q), max(1, len(q) // 60))] = 4
    # windows: cut the operation list, sometimes THROUGH an operation (its two halves join again), sometimes leaving an empty window
    words = []
    for o, l in ops:
        if l > 1 and rng.random() < 0.15:
            a = int(rng.integers(1, l))
            words.append(((l - a) << 4 | o, True))   # the cut falls after this half
            words[-1] = (a << 4 | o, True)
            words.append(((l - a) << 4 | o, False))
        else:
            words.append((l << 4 | o, rng.random() < 0.08))
    pieces, cur = [], []
    for w, cut in words:
        cur.append(w)
        if cut:
            pieces.append(cur)
            cur = []
            if rng.random() < 0.1:
                pieces.append([])
    pieces.append(cur)
    return bytes(q), bytes(t), pieces


@pytest.mark.parametrize("log_gap", [1, 0])
def test_update_extra_batch_equals_the_reference(log_gap):
    import minimap2_amd as mm
    if not os.path.exists(reflib.REFALIGN_SO):
        pytest.skip("oracle/_ref/librefalign.so not built (dev container: make -C oracle)")
    rng = np.random.default_rng(17 + log_gap)
    mat = reflib.ts_mat(2, 4)
    jobs = []
    n_jobs = 60 if EMU else 1500
    for i in range(n_jobs):
        n_ops = int(rng.choice([1, 2, 3, 8, 40, 150, 400 if EMU else 1500]))
        jobs.append(random_region(rng, n_ops, int(rng.choice([1, 2, 4])), float(rng.choice([0.0, 0.03, 0.1])), float(rng.choice([0.0, 0.02, 0.1]))))
    # corner cases: no window, one empty window, a single operation, a gap only in front of a match, a CIGAR that is one joined match
    jobs += [(b"", b"", []), (b"", b"", [[]]), (b"\0\1\2", b"\0\1\2", [[3 << 4]]), (b"\0\0\0\1", b"\0\1", [[2 << 4 | 1], [2 << 4]]),
             (b"\0\1", b"\3\3\0\1", [[2 << 4 | 2, 2 << 4]]), (b"\0\1\2\3", b"\0\1\2\3", [[1 << 4], [1 << 4], [], [2 << 4]])]
    got = mm.update_extra_batch(jobs, mat, 4, 2, log_gap)
    n_shrunk = n_lead = 0
    for i, (qs, ts, pieces) in enumerate(jobs):
        want = reflib.ref_update_extra(qs, ts, pieces, mat, 4, 2, log_gap)
        assert got[i] == want, "region %d (%d operations in %d windows): %r != %r" % (i, sum(len(p) for p in pieces), len(pieces), got[i], want)
        n_shrunk += len(want[0]) < sum(len(p) for p in pieces) - len(pieces)
        n_lead += want[5] > 0 or want[6] > 0
    assert n_shrunk > n_jobs // 10 and n_lead > 0  # the paths this test is for were taken
