"""The interior rows of the splice gap-fill kernel (splice_lean_rows, minimap2_amd/csrc/ksw_splice.hip) compute a cell with one list of gfx950 instructions
(splice_cell); under the wave emulator the kernel runs with the C++ twin next to it.  This case reads the instruction list out of the source, interprets it on both
16-bit halves (tests/test_gf_cell_asm.py's interpreter) and compares every output with the recurrence of ksw2_exts2_sse.c:249-348 (left-aligned gaps, the variant
gap fills use) written in plain integers, for both splice models' site costs; it also checks the spacing rule (no instruction reads the result of the packed
instruction right before it).  Runs without a GPU; tests/test_gpu_ksw.py::test_splice_gap_fill_kernel is the hardware's word on the same instructions."""
import os

import numpy as np
import pytest

from test_gf_cell_asm import _blocks, _function, _pk, _run

SRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "minimap2_amd", "csrc", "ksw_splice.hip")


def _plain(tb, qb, x, v, x2, up, yp, dn, ac, sc):
    """one valid cell of ksw_exts2_sse (ksw2_exts2_sse.c:249-348): z = max(s, a, b, a2 + acceptor), the FIRST candidate that reaches it names the state"""
    a_, b_, q, e, q2, noncan = sc
    z = -e if (tb | qb) & 4 else (a_ if tb == qb else -b_)  # (the splice preset's matrix has a zero N score: sc_N = -e)
    a, b, a2 = x + v, yp + up, x2 + v
    a2a = a2 + ac
    d = 0
    for k, c in enumerate((a, b, a2a)):
        if c > z:
            d, z = k + 1, c
    un, vn = z - v, z - up
    a, b, a2 = a - (z - q), b - (z - q), a2 - (z - q2)
    if a > 0:
        d |= 0x08
    if b > 0:
        d |= 0x10
    m2 = max(a2, dn)
    if m2 - dn > 0:
        d |= 0x20
    return un, vn, max(a, 0) - (q + e), max(b, 0) - (q + e), m2 - q2, d


@pytest.mark.parametrize("sc", [(1, 2, 2, 1, 32, 9), (1, 2, 2, 1, 24, 5), (2, 4, 4, 2, 40, 12)])
def test_splice_cell_assembly_against_the_recurrence(sc):
    a_, b_, q, e, q2, noncan = sc
    qe = q + e
    blocks = _blocks(_function("splice_cell", SRC))
    assert len(blocks) == 2
    pk2 = lambda v: _pk(v, v)
    rng = np.random.default_rng(17 + q2)
    n_pk = None
    for it in range(6000):
        cells = []
        for h in range(2):
            tb, qb = int(rng.integers(0, 5)), int(rng.integers(0, 5))
            if it % 3 == 0:
                qb = tb
            x, yp = int(rng.integers(-qe, -e + 1)), int(rng.integers(-qe, -e + 1))          # x, y in [-(q+e), -e]
            x2 = int(rng.integers(-q2 - noncan, 1))                                          # x2 in [-q2 - donor cost, 0] (no extension cost)
            up, v = int(rng.integers(-q2 - noncan, a_ + q2 + noncan + 1)), int(rng.integers(-q2 - noncan, a_ + q2 + noncan + 1))
            dn, ac = -int(rng.choice([0, noncan // 2, noncan, 3, 5, 7, 10])), -int(rng.choice([0, noncan // 2, noncan, 3, 5, 7, 10]))
            if it % 5 == 0:
                v = up + yp - x  # a tie between the two short-gap candidates
            if it % 7 == 0:
                ac = x + v - x2 - v  # a tie between the short gap and the intron
            cells.append((tb, qb, x, v, x2, up, yp, dn, ac))
        A, B = cells
        env = dict(tv=_pk(A[0], B[0]), qv=_pk(A[1], B[1]), x=_pk(A[2], B[2]), v=_pk(A[3], B[3]), x2=_pk(A[4], B[4]), up=_pk(A[5], B[5]), yp=_pk(A[6], B[6]),
                   dn=_pk(A[7], B[7]), ac=_pk(A[8], B[8]), P_MCH=pk2(a_))
        env.update({"K.misd": pk2(-b_ - a_), "K.scn": pk2(-e), "K.q": pk2(q), "K.q2": pk2(q2), "K.qe": pk2(qe)})
        cnt = _run(blocks, env)
        n_pk = cnt if n_pk is None else n_pk
        assert cnt == n_pk
        for h, cell in enumerate((A, B)):
            want = _plain(*cell, sc)
            half = lambda name: (env[name] >> (16 * h)) & 0xffff
            s16 = lambda t: t - 0x10000 if t & 0x8000 else t
            got = (s16(half("un")), s16(half("vn")), s16(half("a")), s16(half("b")), s16(half("a2")), half("ea"))
            assert got == want, (it, h, cell, got, want)
    assert n_pk == (41, 2)  # 41 packed + 2 32-bit VALU instructions per register set and row
