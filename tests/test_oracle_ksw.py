"""The C restatement (oracle/ksw_extd2.c) against the compiled reference ksw_extd2_sse: all result fields and the CIGAR
must be identical, over every flag set mm_align1 uses (align.c:791,840,844,883), with and without a binding band."""
import numpy as np
import pytest

from reflib import ref_extd2, ora_extd2, ts_mat
from seqsim import random_pair

FLAGS = [0x08, 0x00, 0x40, 0xC2, 0x48 | 0x04, 0x01 | 0x40]  # gap-fill approx, exact, right ext, left ext, generic, score-only
PRESETS = {  # (a, b, q, e, q2, e2)
    "ont": (2, 4, 4, 2, 24, 1),
    "hifi": (1, 4, 6, 2, 26, 1),
    "swap": (2, 4, 24, 1, 4, 2),
    "asm5": (1, 19, 39, 3, 81, 1),
}


def _check(rng, qlen, w, zdrop, end_bonus, flag, preset, err=0.12, n_frac=0.0, indel=0, transition=0):
    a, b, go, ge, go2, ge2 = PRESETS[preset]
    q, t = random_pair(rng, qlen, err, n_frac, indel)
    mat = ts_mat(a, b, 1, transition)
    r = ref_extd2(q, t, mat, go, ge, go2, ge2, w, zdrop, end_bonus, flag)
    o = ora_extd2(q, t, mat, go, ge, go2, ge2, w, zdrop, end_bonus, flag)
    assert r == o, (qlen, len(q), len(t), w, zdrop, end_bonus, hex(flag), preset)


@pytest.mark.parametrize("flag", FLAGS)
def test_unbanded_small(flag):
    rng = np.random.default_rng(flag + 1)
    for it in range(150):
        qlen = int(rng.integers(1, 400))
        _check(rng, qlen, 30001, int(rng.choice([-1, 200, 400])), int(rng.choice([-1, 10])), flag,
               str(rng.choice(list(PRESETS))), err=float(rng.choice([0.0, 0.05, 0.12, 0.3])),
               n_frac=float(rng.choice([0, 0, 0.02])), transition=3 if flag & 4 else 0)


@pytest.mark.parametrize("flag", [0x40, 0xC2, 0x00, 0x08])
def test_band_binding(flag):
    rng = np.random.default_rng(100 + flag)
    for it in range(120):
        qlen = int(rng.integers(20, 900))
        w = int(rng.integers(1, 120))
        indel = int(rng.choice([0, 0, 30, -30, 150, -150]))
        _check(rng, qlen, w, int(rng.choice([-1, 100, 400])), int(rng.choice([-1, 10])), flag, "ont",
               err=float(rng.choice([0.02, 0.12])), indel=indel)


def test_multiple_of_16_lengths():
    # score-chunk overshoot past the s[] array happens when tlen is a multiple of 16 (see oracle/ksw_extd2.c header)
    rng = np.random.default_rng(7)
    for tl in (16, 32, 48, 64, 256):
        for it in range(20):
            t = rng.integers(0, 4, tl, dtype=np.uint8)
            q = rng.integers(0, 4, int(rng.integers(1, 2 * tl)), dtype=np.uint8)
            mat = ts_mat(2, 4)
            for flag in (0x08, 0x40, 0xC2, 0):
                for w in (5, 751, 30001):
                    assert ref_extd2(q, t, mat, 4, 2, 24, 1, w, 400, 10, flag) == ora_extd2(q, t, mat, 4, 2, 24, 1, w, 400, 10, flag)


def test_long_extension_band_751():
    rng = np.random.default_rng(11)
    for it in range(6):
        qlen = int(rng.integers(1500, 3000))
        _check(rng, qlen, 751, 400, 10, [0x40, 0xC2][it & 1], "ont", indel=int(rng.choice([0, 700, -700, 900])))
