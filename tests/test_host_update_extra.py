"""The HOST's twin of mm_append_cigar + mm_fix_cigar + mm_update_extra (align.cpp: what finishes a region where region_finish_kernel does not run) against
the UNMODIFIED reference's own static routines (oracle/ref_align_shim.c compiles align.c where it lies) on adversarial CIGARs: tests/test_gpu_update_extra.py's
generator -- operations cut by window borders, empty windows, gaps that slide into each other, clusters of insertions and deletions, leading gaps, matches that
are used up.  Through a test hook of the check library (tests/cpucheck/backend_check.cpp: check_update_extra_host); no product ABI involved."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import reflib  # noqa: E402
import test_gpu_update_extra as T  # noqa: E402  (the generator; its own cases are GPU-marked)

CHECK_SO = os.path.join(HERE, "_build", "libmm2amd_check.so")


def _host_update_extra(lib, jobs, mat, q, e, log_gap, eqx=False):
    import minimap2_amd as mm
    n = len(jobs)
    arr = (mm.FinJob * max(n, 1))()
    keep, tot = [], 0
    for i, (qs, ts, pieces) in enumerate(jobs):
        qb, tb = bytes(qs), bytes(ts)
        parr = [(C.c_uint32 * max(len(p_), 1))(*p_) for p_ in pieces]
        pp = (C.POINTER(C.c_uint32) * max(len(pieces), 1))(*[C.cast(a, C.POINTER(C.c_uint32)) for a in parr])
        pl = (C.c_int32 * max(len(pieces), 1))(*[len(p_) for p_ in pieces])
        keep.append((qb, tb, parr, pp, pl))
        arr[i].query, arr[i].target = C.cast(C.c_char_p(qb), C.c_void_p), C.cast(C.c_char_p(tb), C.c_void_p)
        arr[i].qlen, arr[i].tlen, arr[i].n_pieces, arr[i].piece, arr[i].piece_len = len(qb), len(tb), len(pieces), pp, pl
        tot += sum(len(p_) for p_ in pieces) + (len(qb) + 8 if eqx else 0)
    res = (mm.FinRes * max(n, 1))()
    pool = (C.c_uint32 * max(tot, 1))()
    lib.check_update_extra_host.restype = C.c_int
    lib.check_update_extra_host.argtypes = [C.c_int, C.c_void_p, C.c_char_p, C.c_int8, C.c_int8, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    assert lib.check_update_extra_host(n, arr, bytes(mat), q, e, 1 if log_gap else 0, 1 if eqx else 0, res, pool, max(tot, 1)) == 0
    out = []
    for i in range(n):
        r = res[i]
        out.append(None if r.n_cigar < 0 else (tuple(pool[r.cigar_off:r.cigar_off + r.n_cigar]), r.blen, r.mlen, r.n_ambi, r.dp_max, r.qshift, r.tshift, r.is_spliced))
    return out


@pytest.mark.parametrize("log_gap", [1, 0])
def test_host_update_extra_equals_the_reference(log_gap):
    if not os.path.exists(reflib.REFALIGN_SO) or not os.path.exists(CHECK_SO):
        pytest.skip("needs oracle/_ref/librefalign.so and tests/_build/libmm2amd_check.so (dev container)")
    lib = C.CDLL(CHECK_SO)
    rng = np.random.default_rng(29 + log_gap)
    mat = reflib.ts_mat(2, 4)
    jobs = [T.random_region(rng, int(rng.choice([1, 2, 3, 8, 40, 150, 600])), int(rng.choice([1, 2, 4])), float(rng.choice([0.0, 0.03, 0.1])), float(rng.choice([0.0, 0.02, 0.1])))
            for _ in range(700)]
    jobs += [(b"\0\1\2", b"\0\1\2", [[3 << 4]]), (b"\0\0\0\1", b"\0\1", [[2 << 4 | 1], [2 << 4]]), (b"\0\1", b"\3\3\0\1", [[2 << 4 | 2, 2 << 4]]),
             (b"\0\1\2\3", b"\0\1\2\3", [[1 << 4], [1 << 4], [], [2 << 4]])]
    got = _host_update_extra(lib, jobs, mat, 4, 2, log_gap)
    n_shrunk = n_lead = 0
    for i, (qs, ts, pieces) in enumerate(jobs):
        want = reflib.ref_update_extra(qs, ts, pieces, mat, 4, 2, log_gap)
        assert got[i] == want, "region %d (%d operations in %d windows): %r != %r" % (i, sum(len(p) for p in pieces), len(pieces), got[i], want)
        n_shrunk += len(want[0]) < sum(len(p) for p in pieces) - len(pieces)
        n_lead += want[5] > 0 or want[6] > 0
    assert n_shrunk > 70 and n_lead > 0  # the paths this test is for were taken


def test_host_eqx_split_equals_the_reference():
    """MM_F_EQX on top (mm_update_cigar_eqx, align.c:183-252): matches cut into = and X stretches, or relabelled in place when no match splits"""
    if not os.path.exists(reflib.REFALIGN_SO) or not os.path.exists(CHECK_SO):
        pytest.skip("needs oracle/_ref/librefalign.so and tests/_build/libmm2amd_check.so (dev container)")
    lib = C.CDLL(CHECK_SO)
    rng = np.random.default_rng(31)
    mat = reflib.ts_mat(2, 4)
    jobs = [T.random_region(rng, int(rng.choice([1, 2, 3, 8, 40, 150])), int(rng.choice([1, 2, 4])), float(rng.choice([0.0, 0.03, 0.1])), float(rng.choice([0.0, 0.02, 0.1])))
            for _ in range(500)]
    jobs += [(b"\0\1\2", b"\0\1\2", [[3 << 4]]), (b"\0\1\2", b"\3\3\3", [[3 << 4]]), (b"\0\1\2\3", b"\0\1\1\3", [[4 << 4]])]
    got = _host_update_extra(lib, jobs, mat, 4, 2, 1, eqx=True)
    n_split = n_inplace = 0
    for i, (qs, ts, pieces) in enumerate(jobs):
        want = reflib.ref_update_extra(qs, ts, pieces, mat, 4, 2, 1, eqx=True)
        assert got[i] == want, "region %d: %r != %r" % (i, got[i], want)
        ops = [c & 0xf for c in want[0]]
        n_split += 8 in ops
        n_inplace += 7 in ops and 8 not in ops
    assert n_split > 100 and n_inplace > 20
