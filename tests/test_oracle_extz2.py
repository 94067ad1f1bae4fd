"""oracle/ksw_extz2.c (single-affine, lane-exact) against the compiled reference's ksw_extz2_sse: every result field and the
CIGAR, over the flag sets mm_align1 uses, unbanded and with binding bands, for several (match, mismatch, gap) settings."""
import os

import numpy as np
import pytest

import reflib
from reflib import ref_extz2, ora_extz2, ts_mat
from seqsim import random_pair

pytestmark = pytest.mark.skipif(not os.path.exists(reflib.REF_SO), reason="needs oracle/_ref (dev container)")

SCORING = {"a2b4q4e2": (2, 4, 4, 2), "a1b4q6e2": (1, 4, 6, 2), "a1b9q16e2": (1, 9, 16, 2), "a2b8q12e2": (2, 8, 12, 2), "a1b2q2e1": (1, 2, 2, 1)}
FLAGS = [0x08, 0x00, 0x40, 0xC2, 0x0C, 0x41, 0x18]


def _check(rng, qlen, w, zdrop, eb, flag, sc, err=0.12, n_frac=0.0, indel=0, transition=0):
    a, b, go, ge = SCORING[sc]
    q, t = random_pair(rng, qlen, err, n_frac, indel)
    mat = ts_mat(a, b, 1, transition)
    assert ref_extz2(q, t, mat, go, ge, w, zdrop, eb, flag) == ora_extz2(q, t, mat, go, ge, w, zdrop, eb, flag), (len(q), len(t), w, zdrop, eb, hex(flag), sc)


@pytest.mark.parametrize("flag", FLAGS)
def test_unbanded(flag):
    rng = np.random.default_rng(flag + 3)
    for it in range(150):
        _check(rng, int(rng.integers(1, 400)), 30001, int(rng.choice([-1, 100, 400])), int(rng.choice([-1, 10])), flag, str(rng.choice(list(SCORING))),
               err=float(rng.choice([0.0, 0.05, 0.12, 0.3])), n_frac=float(rng.choice([0, 0, 0.02])), transition=3 if flag & 4 else 0)


@pytest.mark.parametrize("flag", [0x40, 0xC2, 0x00, 0x08])
def test_band_binding(flag):
    rng = np.random.default_rng(200 + flag)
    for it in range(150):
        _check(rng, int(rng.integers(20, 900)), int(rng.integers(1, 120)), int(rng.choice([-1, 100, 400])), int(rng.choice([-1, 10])), flag,
               str(rng.choice(list(SCORING))), err=float(rng.choice([0.02, 0.12])), indel=int(rng.choice([0, 0, 30, -30, 150, -150])))


def test_multiple_of_16_lengths_and_long_band():
    rng = np.random.default_rng(9)
    mat = ts_mat(2, 4)
    for tl in (16, 32, 48, 64, 256):
        for it in range(10):
            t = rng.integers(0, 4, tl, dtype=np.uint8)
            q = rng.integers(0, 4, int(rng.integers(1, 2 * tl)), dtype=np.uint8)
            for flag in (0x08, 0x40, 0xC2, 0):
                for w in (5, 751, 30001):
                    assert ref_extz2(q, t, mat, 4, 2, w, 400, 10, flag) == ora_extz2(q, t, mat, 4, 2, w, 400, 10, flag)
    for it in range(4):
        _check(rng, int(rng.integers(1500, 3000)), 751, 400, 10, [0x40, 0xC2][it & 1], "a2b4q4e2", indel=int(rng.choice([0, 700, -700])))
