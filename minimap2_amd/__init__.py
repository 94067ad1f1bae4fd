"""minimap2_amd -- MI355X-native seed-chain-extend engine behind minimap2's API.

This package is a thin ctypes mirror of the C ABI in include/mm2amd.h; all compute happens in
libmm2amd.so (hand-written HIP for gfx950).  There is no CPU fallback: if the library is not built or no
GPU is visible, calls raise."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmm2amd.so")


class Mm2AmdError(RuntimeError):
    pass


class KswJob(C.Structure):  # mm2amd_ksw_job_t
    _fields_ = [("query", C.c_void_p), ("target", C.c_void_p), ("qlen", C.c_int32), ("tlen", C.c_int32), ("w", C.c_int32),
                ("zdrop", C.c_int32), ("end_bonus", C.c_int32), ("flag", C.c_int32)]


class KswRes(C.Structure):  # mm2amd_ksw_res_t
    _fields_ = [(n, C.c_int32) for n in ("max", "zdropped", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q", "score",
                                         "n_cigar", "reach_end")] + [("cigar_off", C.c_uint32)]


_lib = None


def lib():
    """Load libmm2amd.so (built in-tree by minimap2_amd.build); raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Mm2AmdError("libmm2amd.so is not built (run `python -m minimap2_amd.build`); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.mm2amd_last_error.restype = C.c_char_p
        L.mm2amd_ksw_extd2_batch.restype = C.c_int
        L.mm2amd_ksw_extd2_batch.argtypes = [C.c_int, C.POINTER(KswJob), C.c_int8, C.c_char_p, C.c_int8, C.c_int8, C.c_int8,
                                             C.c_int8, C.POINTER(KswRes), C.POINTER(C.c_uint32), C.c_size_t]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise Mm2AmdError("mm2amd error %d: %s" % (rc, lib().mm2amd_last_error().decode()))


def ksw_extd2_batch(jobs, mat, gapo, gape, gapo2, gape2):
    """jobs: list of (query_bytes, target_bytes, w, zdrop, end_bonus, flag) with nt4 codes 0..4.
    Returns a list of (max, zdropped, max_q, max_t, mqe, mqe_t, mte, mte_q, score, reach_end, cigar_tuple),
    the same tuple layout tests/reflib.py produces for the reference's ksw_extd2_sse."""
    n = len(jobs)
    arr = (KswJob * n)()
    keep = []
    tot = 0
    for i, (q, t, w, zdrop, end_bonus, flag) in enumerate(jobs):
        qb, tb = bytes(q), bytes(t)
        keep.append((qb, tb))
        arr[i].query = C.cast(C.c_char_p(qb), C.c_void_p)
        arr[i].target = C.cast(C.c_char_p(tb), C.c_void_p)
        arr[i].qlen, arr[i].tlen, arr[i].w, arr[i].zdrop, arr[i].end_bonus, arr[i].flag = len(qb), len(tb), w, zdrop, end_bonus, flag
        tot += len(qb) + len(tb)
    res = (KswRes * n)()
    pool = (C.c_uint32 * max(tot, 1))()
    _check(lib().mm2amd_ksw_extd2_batch(n, arr, 5, bytes(mat), gapo, gape, gapo2, gape2, res, pool, max(tot, 1)))
    out = []
    for r in res:
        out.append((r.max, r.zdropped, r.max_q, r.max_t, r.mqe, r.mqe_t, r.mte, r.mte_q, r.score, r.reach_end,
                    tuple(pool[r.cigar_off:r.cigar_off + r.n_cigar])))
    return out
