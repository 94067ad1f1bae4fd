"""ctypes bindings used by the tests only: the compiled reference (oracle/_ref) and our C restatement
(oracle/liboracle.so).  TEST INFRASTRUCTURE -- never imported by minimap2_amd/."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libminimap2_ref.so")
REF_BIN = os.path.join(ORACLE_DIR, "_ref", "minimap2_ref")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")


def ensure_built():
    """(Re)build what can be built here: liboracle.so always, _ref only when /root/reference exists."""
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)


class KswExtz(C.Structure):  # ksw_extz_t, ksw2.h:34-43
    _fields_ = [("max_zd", C.c_uint32), ("max_q", C.c_int), ("max_t", C.c_int), ("mqe", C.c_int), ("mqe_t", C.c_int),
                ("mte", C.c_int), ("mte_q", C.c_int), ("score", C.c_int), ("m_cigar", C.c_int), ("n_cigar", C.c_int),
                ("reach_end", C.c_int), ("cigar", C.POINTER(C.c_uint32))]

    @property
    def max(self):
        return self.max_zd & 0x7FFFFFFF

    @property
    def zdropped(self):
        return self.max_zd >> 31


class OraEz(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("max", "zdropped", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q", "score",
                                         "n_cigar", "reach_end", "cigar_overflow")]


_ref = None
_ora = None


def ref():
    global _ref
    if _ref is None:
        if not os.path.exists(REF_SO):
            ensure_built()
        _ref = C.CDLL(REF_SO)
        _ref.ksw_extd2_sse.restype = None
        _ref.ksw_extd2_sse.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int8, C.c_char_p,
                                       C.c_int8, C.c_int8, C.c_int8, C.c_int8, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.POINTER(KswExtz)]
        _ref.free.argtypes = [C.c_void_p]
    return _ref


def ora():
    global _ora
    if _ora is None:
        ensure_built()
        _ora = C.CDLL(ORACLE_SO)
        _ora.ora_ksw_extd2.restype = None
        _ora.ora_ksw_extd2.argtypes = [C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int8, C.c_char_p,
                                       C.c_int8, C.c_int8, C.c_int8, C.c_int8, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.POINTER(OraEz), C.POINTER(C.c_uint32), C.c_int]
    return _ora


def ez_tuple_ref(ez):
    cig = [ez.cigar[i] for i in range(ez.n_cigar)]
    return (ez.max, ez.zdropped, ez.max_q, ez.max_t, ez.mqe, ez.mqe_t, ez.mte, ez.mte_q, ez.score, ez.reach_end, tuple(cig))


def ref_extd2(q, t, mat, go, ge, go2, ge2, w, zdrop, end_bonus, flag):
    """Run the reference ksw_extd2_sse (dispatches to the SSE4.1 build here) on byte strings of 0..4 codes."""
    ez = KswExtz()
    ref().ksw_extd2_sse(None, len(q), bytes(q), len(t), bytes(t), 5, bytes(mat), go, ge, go2, ge2, w, zdrop, end_bonus, flag,
                        C.byref(ez))
    out = ez_tuple_ref(ez)
    if ez.cigar:
        ref().free(ez.cigar)
    return out


def ora_extd2(q, t, mat, go, ge, go2, ge2, w, zdrop, end_bonus, flag):
    ez = OraEz()
    cap = len(q) + len(t) + 8
    buf = (C.c_uint32 * cap)()
    ora().ora_ksw_extd2(len(q), bytes(q), len(t), bytes(t), 5, bytes(mat), go, ge, go2, ge2, w, zdrop, end_bonus, flag,
                        C.byref(ez), buf, cap)
    assert not ez.cigar_overflow
    return (ez.max, ez.zdropped, ez.max_q, ez.max_t, ez.mqe, ez.mqe_t, ez.mte, ez.mte_q, ez.score, ez.reach_end,
            tuple(buf[i] for i in range(ez.n_cigar)))


def ts_mat(a, b, sc_ambi=1, transition=0):
    """ksw_gen_ts_mat, align.c:10-36 -> 25 signed bytes"""
    import numpy as np
    a, b, sc_ambi = abs(a), -abs(b), -abs(sc_ambi)
    m = np.full((5, 5), b, dtype=np.int8)
    for i in range(4):
        m[i, i] = a
    m[:, 4] = sc_ambi
    m[4, :] = sc_ambi
    if transition != 0 and -abs(transition) != b:
        tr = -abs(transition)
        m[0, 2] = m[1, 3] = m[2, 0] = m[3, 1] = tr
    return m.tobytes()
