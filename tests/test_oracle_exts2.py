"""oracle/ksw_exts2.c (splice-aware, lane-exact) against the compiled reference's ksw_exts2_sse: every result field and the
CIGAR (including N operations), for the flag sets mm_align1 uses with -x splice (forward/reverse transcript strand, left/right
extension, gap fill), the old and the miniprot-style splice model, with and without junction annotation bytes."""
import os

import numpy as np
import pytest

import reflib
from reflib import ref_exts2, ora_exts2, ts_mat
from seqsim import spliced_pair

pytestmark = pytest.mark.skipif(not os.path.exists(reflib.REF_SO), reason="needs oracle/_ref (dev container)")

SPL = [(1, 2, 2, 1, 32, 9), (1, 4, 6, 1, 24, 5)]  # (a, b, q, e, q2, noncan): -x splice, -x splice:hq
FOR, REV, FLANK, CMPLX, SCORE = 0x100, 0x200, 0x400, 0x800, 0x1000


@pytest.mark.parametrize("strand", [FOR, REV])
@pytest.mark.parametrize("base", [0x08, 0x40, 0xC2, 0x00, 0x01 | 0x40])
def test_spliced_alignments(strand, base):
    rng = np.random.default_rng(strand + base)
    for it in range(60):
        a, b, go, ge, go2, noncan = SPL[it & 1]
        mat = ts_mat(a, b, 1, 0)
        q, t = spliced_pair(rng, int(rng.integers(1, 5)), float(rng.choice([0.0, 0.03, 0.1])))
        if base & 0x80:
            q, t = q[::-1].copy(), t[::-1].copy()
        flag = base | strand | FLANK | (CMPLX if it % 3 else 0)
        zdrop, eb = int(rng.choice([-1, 100, 200])), int(rng.choice([-1, 5]))
        assert ref_exts2(q, t, mat, go, ge, go2, noncan, zdrop, eb, 9, 5, flag) == ora_exts2(q, t, mat, go, ge, go2, noncan, zdrop, eb, 9, 5, flag), (it, len(q), len(t), hex(flag))


def test_junction_annotation_bytes_and_generic_scores():
    rng = np.random.default_rng(5)
    for it in range(60):
        a, b, go, ge, go2, noncan = SPL[it & 1]
        q, t = spliced_pair(rng, 3, 0.05)
        strand = [FOR, REV][it & 1]
        flag = [0x08, 0x40, 0xC2][it % 3] | strand | FLANK | CMPLX
        if flag & 0x80:
            q, t = q[::-1].copy(), t[::-1].copy()
        junc = rng.integers(0, 16, len(t), dtype=np.uint8)
        junc[rng.random(len(t)) < 0.9] = 0
        mat = ts_mat(a, b, 1, 0)
        assert ref_exts2(q, t, mat, go, ge, go2, noncan, 200, -1, 9, 5, flag, junc) == ora_exts2(q, t, mat, go, ge, go2, noncan, 200, -1, 9, 5, flag, junc)
        sj = rng.integers(0, 256, len(t), dtype=np.uint8)
        sj[rng.random(len(t)) < 0.8] = 0xff
        assert ref_exts2(q, t, mat, go, ge, go2, noncan, 200, -1, 9, 5, flag | SCORE, sj) == ora_exts2(q, t, mat, go, ge, go2, noncan, 200, -1, 9, 5, flag | SCORE, sj)
        mat2 = ts_mat(a, b, 1, 3)
        assert ref_exts2(q, t, mat2, go, ge, go2, noncan, 200, -1, 9, 5, flag | 0x04) == ora_exts2(q, t, mat2, go, ge, go2, noncan, 200, -1, 9, 5, flag | 0x04)


def test_lengths_multiple_of_16_and_unrelated_sequences():
    rng = np.random.default_rng(6)
    mat = ts_mat(1, 2, 1, 0)
    for tl in (16, 32, 48, 64, 256):
        for it in range(8):
            t = rng.integers(0, 4, tl, dtype=np.uint8)
            q = rng.integers(0, 4, int(rng.integers(1, 2 * tl)), dtype=np.uint8)
            for flag in (0x08, 0x40, 0xC2, 0):
                f = flag | FOR | FLANK | CMPLX
                assert ref_exts2(q, t, mat, 2, 1, 32, 9, 200, 5, 9, 5, f) == ora_exts2(q, t, mat, 2, 1, 32, 9, 200, 5, 9, 5, f)
