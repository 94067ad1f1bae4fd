"""oracle sketch / exact radix sort / chaining restatements against the compiled reference."""
import numpy as np
import pytest

import reflib


def _rand_seq(rng, n, n_frac=0.0, lowcomplex=False):
    if lowcomplex:
        unit = rng.integers(0, 4, int(rng.integers(1, 6)))
        s = np.resize(unit, n)
        mut = rng.random(n) < 0.02
        s = np.where(mut, rng.integers(0, 4, n), s)
    else:
        s = rng.integers(0, 4, n)
    out = np.frombuffer(b"ACGT", dtype=np.uint8)[s].copy()
    if n_frac > 0:
        out[rng.random(n) < n_frac] = ord("N")
    if n > 10:
        out[rng.integers(0, n, 3)] = np.frombuffer(b"acg", dtype=np.uint8)  # lower case is accepted too
    return out.tobytes()


@pytest.mark.parametrize("w,k,hpc", [(10, 15, 0), (19, 19, 0), (5, 15, 0), (10, 19, 1), (1, 7, 0), (50, 28, 0), (11, 21, 0)])
def test_sketch_matches_reference(w, k, hpc):
    rng = np.random.default_rng(w * 100 + k)
    for it in range(60):
        n = int(rng.choice([1, 5, k, k + w - 1, k + w, 100, 1000, 12000]))
        seq = _rand_seq(rng, n, float(rng.choice([0, 0, 0.001, 0.05])), lowcomplex=bool(it % 5 == 0))
        r = reflib.ref_sketch(seq, w, k, rid=it, is_hpc=hpc)
        o = reflib.ora_sketch(seq, w, k, rid=it, is_hpc=hpc)
        assert r.shape == o.shape and (r == o).all(), (n, w, k, hpc)


def test_sketch_palindromes_and_homopolymers():
    for seq in (b"ACGT" * 500, b"A" * 3000, b"AT" * 2000, b"ACGTACGTTTTTAAAAACGTACGT" * 100, b"N" * 100 + b"ACGTTGCA" * 300):
        for (w, k) in ((10, 15), (5, 4), (3, 6)):
            assert (reflib.ref_sketch(seq, w, k) == reflib.ora_sketch(seq, w, k)).all()


def test_exact_unstable_sort_with_ties():
    rng = np.random.default_rng(3)
    for it in range(300):
        n = int(rng.choice([0, 1, 2, 63, 64, 65, 66, 200, 1000, 5000]))
        bits = int(rng.choice([1, 3, 8, 12, 16, 33, 64]))
        x = rng.integers(0, 2 ** min(bits, 63), n, dtype=np.uint64)
        if bits == 64:
            x = x << np.uint64(1) | rng.integers(0, 2, n, dtype=np.uint64)
        a = np.stack([x, np.arange(n, dtype=np.uint64)], axis=1)
        assert (reflib.ref_sort128(a) == reflib.ora_sort128(a)).all(), (n, bits)
        assert (reflib.ref_sort64(x) == reflib.ora_sort64(x)).all()


def _anchors(rng, n_true, n_noise, qlen=10000, err=0.1, n_rid=3):
    """Anchor set shaped like collect_seed_hits' output: a colinear run plus random hits, sorted with the reference's sort."""
    xs, ys = [], []
    rid = int(rng.integers(0, n_rid)); strand = int(rng.integers(0, 2)); r0 = int(rng.integers(0, 10 ** 6))
    q = 20
    drift = 0
    for i in range(n_true):
        q += int(rng.integers(1, 40))
        if rng.random() < err:
            drift += int(rng.integers(-8, 9))
        if rng.random() < 0.01:
            drift += int(rng.integers(-2000, 2000))
        xs.append(strand << 63 | rid << 32 | max(0, r0 + q + drift)); ys.append(15 << 32 | q)
    for i in range(n_noise):
        xs.append(int(rng.integers(0, 2)) << 63 | int(rng.integers(0, n_rid)) << 32 | int(rng.integers(0, 10 ** 6 + qlen)))
        ys.append(15 << 32 | int(rng.integers(14, qlen)))
    a = np.array([xs, ys], dtype=np.uint64).T.copy().reshape(-1, 2)
    return reflib.ref_sort128(a)


@pytest.mark.parametrize("cdna", [0, 1])
def test_chaining_matches_reference(cdna):
    rng = np.random.default_rng(17 + cdna)
    for it in range(80):
        a = _anchors(rng, int(rng.integers(0, 400)), int(rng.integers(0, 800)))
        args = (5000, 5000, int(rng.choice([100, 500])), int(rng.choice([5, 25])), int(rng.choice([50, 5000])), 3, 40,
                float(np.float32(0.8 * 0.01 * 15)), 0.0, cdna, 1)
        ru, ra = reflib.ref_lchain_dp(a, *args)
        ou, oa = reflib.ora_lchain_dp(a, *args)
        assert ru.shape == ou.shape and (ru == ou).all(), it
        assert ra.shape == oa.shape and (ra == oa).all(), it


def test_chaining_dense_repeats():
    # many anchors within max_dist: exercises the max_skip early exit, the max_iter clamp and the max_ii shortcut
    rng = np.random.default_rng(5)
    for it in range(20):
        n = 3000
        q = np.sort(rng.integers(15, 9000, n))
        r = q + rng.integers(-30, 31, n) + 10 ** 5
        a = np.stack([(np.uint64(1) << np.uint64(32)) | r.astype(np.uint64), (np.uint64(15) << np.uint64(32)) | q.astype(np.uint64)], axis=1)
        a = reflib.ref_sort128(a)
        args = (5000, 5000, 500, 25, int(rng.choice([200, 5000])), 3, 40, 0.12, 0.0, 0, 1)
        ru, ra = reflib.ref_lchain_dp(a, *args)
        ou, oa = reflib.ora_lchain_dp(a, *args)
        assert (ru == ou).all() and (ra == oa).all()
