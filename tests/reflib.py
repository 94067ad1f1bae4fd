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


def ref_extz2(q, t, mat, go, ge, w, zdrop, end_bonus, flag):
    """the reference's ksw_extz2_sse (single-affine)"""
    R = ref()
    R.ksw_extz2_sse.restype = None
    R.ksw_extz2_sse.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int8, C.c_char_p, C.c_int8, C.c_int8, C.c_int, C.c_int,
                                C.c_int, C.c_int, C.POINTER(KswExtz)]
    ez = KswExtz()
    R.ksw_extz2_sse(None, len(q), bytes(q), len(t), bytes(t), 5, bytes(mat), go, ge, w, zdrop, end_bonus, flag, C.byref(ez))
    out = ez_tuple_ref(ez)
    if ez.cigar:
        R.free(ez.cigar)
    return out


def ora_extz2(q, t, mat, go, ge, w, zdrop, end_bonus, flag):
    O = ora()
    O.ora_ksw_extz2.restype = None
    O.ora_ksw_extz2.argtypes = [C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int8, C.c_char_p, C.c_int8, C.c_int8, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.POINTER(OraEz), C.POINTER(C.c_uint32), C.c_int]
    ez = OraEz()
    cap = len(q) + len(t) + 8
    buf = (C.c_uint32 * cap)()
    O.ora_ksw_extz2(len(q), bytes(q), len(t), bytes(t), 5, bytes(mat), go, ge, w, zdrop, end_bonus, flag, C.byref(ez), buf, cap)
    assert not ez.cigar_overflow
    return (ez.max, ez.zdropped, ez.max_q, ez.max_t, ez.mqe, ez.mqe_t, ez.mte, ez.mte_q, ez.score, ez.reach_end,
            tuple(buf[i] for i in range(ez.n_cigar)))


EXTS_ARGS = [C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int8, C.c_char_p, C.c_int8, C.c_int8, C.c_int8, C.c_int8, C.c_int, C.c_int, C.c_int8, C.c_int8,
             C.c_int, C.c_char_p]


def ref_exts2(q, t, mat, go, ge, go2, noncan, zdrop, end_bonus, junc_bonus, junc_pen, flag, junc=None):
    """the reference's ksw_exts2_sse (splice-aware)"""
    R = ref()
    R.ksw_exts2_sse.restype = None
    R.ksw_exts2_sse.argtypes = [C.c_void_p] + EXTS_ARGS + [C.POINTER(KswExtz)]
    ez = KswExtz()
    R.ksw_exts2_sse(None, len(q), bytes(q), len(t), bytes(t), 5, bytes(mat), go, ge, go2, noncan, zdrop, end_bonus, junc_bonus, junc_pen, flag,
                    None if junc is None else bytes(junc), C.byref(ez))
    out = ez_tuple_ref(ez)
    if ez.cigar:
        R.free(ez.cigar)
    return out


def ora_exts2(q, t, mat, go, ge, go2, noncan, zdrop, end_bonus, junc_bonus, junc_pen, flag, junc=None):
    O = ora()
    O.ora_ksw_exts2.restype = None
    O.ora_ksw_exts2.argtypes = EXTS_ARGS + [C.POINTER(OraEz), C.POINTER(C.c_uint32), C.c_int]
    ez = OraEz()
    cap = len(q) + len(t) + 8
    buf = (C.c_uint32 * cap)()
    O.ora_ksw_exts2(len(q), bytes(q), len(t), bytes(t), 5, bytes(mat), go, ge, go2, noncan, zdrop, end_bonus, junc_bonus, junc_pen, flag,
                    None if junc is None else bytes(junc), C.byref(ez), buf, cap)
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


# ---------------------------------------------------------------------------------------------------------
# sketch / sort / chaining bindings
# ---------------------------------------------------------------------------------------------------------
import numpy as np


class MM128V(C.Structure):  # mm128_v, minimap.h:78
    _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("a", C.c_void_p)]


_libc = C.CDLL(None)
_libc.malloc.restype = C.c_void_p
_libc.malloc.argtypes = [C.c_size_t]
_libc.free.argtypes = [C.c_void_p]


def ref_sketch(seq, w, k, rid=0, is_hpc=0):
    """mm_sketch of an ASCII sequence -> uint64 array of shape (n, 2)"""
    R = ref()
    R.mm_sketch.restype = None
    R.mm_sketch.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.POINTER(MM128V)]
    v = MM128V(0, 0, None)
    R.mm_sketch(None, seq, len(seq), w, k, rid, is_hpc, C.byref(v))
    out = np.ctypeslib.as_array(C.cast(v.a, C.POINTER(C.c_uint64)), shape=(v.n, 2)).copy() if v.n else np.zeros((0, 2), np.uint64)
    if v.a:
        _libc.free(v.a)
    return out


def ora_sketch(seq, w, k, rid=0, is_hpc=0):
    O = ora()
    O.ora_sketch.restype = C.c_int64
    O.ora_sketch.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_void_p, C.c_int64]
    cap = len(seq) + 1
    buf = np.zeros((cap, 2), np.uint64)
    n = O.ora_sketch(seq, len(seq), w, k, rid, is_hpc, buf.ctypes.data, cap)
    assert n <= cap
    return buf[:n].copy()


def _sorter(libf, name):
    f = getattr(libf(), name)
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_void_p]

    def run(arr):
        a = np.ascontiguousarray(arr, dtype=np.uint64).copy()
        f(a.ctypes.data, a.ctypes.data + a.nbytes)
        return a
    return run


def ref_sort128(arr): return _sorter(ref, "radix_sort_128x")(arr)
def ora_sort128(arr): return _sorter(ora, "ora_radix_sort_128x")(arr)
def ref_sort64(arr): return _sorter(ref, "radix_sort_64")(arr)
def ora_sort64(arr): return _sorter(ora, "ora_radix_sort_64")(arr)


CHAIN_ARGS = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int64]


def ref_lchain_dp(a, max_dist_x, max_dist_y, bw, max_skip, max_iter, min_cnt, min_sc, pen_gap, pen_skip, is_cdna, n_seg):
    """mg_lchain_dp -> (u array, compacted anchors (n,2))"""
    R = ref()
    R.mg_lchain_dp.restype = C.c_void_p
    R.mg_lchain_dp.argtypes = CHAIN_ARGS + [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.c_void_p]
    a = np.ascontiguousarray(a, dtype=np.uint64)
    n = a.shape[0]
    if n == 0:
        return np.zeros(0, np.uint64), np.zeros((0, 2), np.uint64)
    mem = _libc.malloc(a.nbytes)
    C.memmove(mem, a.ctypes.data, a.nbytes)
    n_u = C.c_int(0)
    u = C.c_void_p()
    b = R.mg_lchain_dp(max_dist_x, max_dist_y, bw, max_skip, max_iter, min_cnt, min_sc, pen_gap, pen_skip, is_cdna, n_seg, n, mem,
                       C.byref(n_u), C.byref(u), None)
    if n_u.value == 0:
        return np.zeros(0, np.uint64), np.zeros((0, 2), np.uint64)
    uu = np.ctypeslib.as_array(C.cast(u, C.POINTER(C.c_uint64)), shape=(n_u.value,)).copy()
    na = int((uu & np.uint64(0xffffffff)).sum())
    bb = np.ctypeslib.as_array(C.cast(b, C.POINTER(C.c_uint64)), shape=(na, 2)).copy()
    _libc.free(u); _libc.free(b)
    return uu, bb


def ora_lchain_dp(a, max_dist_x, max_dist_y, bw, max_skip, max_iter, min_cnt, min_sc, pen_gap, pen_skip, is_cdna, n_seg):
    O = ora()
    O.ora_lchain_dp.restype = C.c_int
    O.ora_lchain_dp.argtypes = CHAIN_ARGS + [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
    a = np.ascontiguousarray(a, dtype=np.uint64).copy()
    n = a.shape[0]
    u = np.zeros(max(n, 1), np.uint64)
    na = C.c_int64(0)
    n_u = O.ora_lchain_dp(max_dist_x, max_dist_y, bw, max_skip, max_iter, min_cnt, min_sc, pen_gap, pen_skip, is_cdna, n_seg, n,
                          a.ctypes.data, u.ctypes.data, C.byref(na))
    return u[:n_u].copy(), a[:na.value].copy()


# ---------------------------------------------------------------------------------------------------------
# whole-read mapping with the compiled reference: mm_idx_str + mm_map (minimap.h:324, :350)
# ---------------------------------------------------------------------------------------------------------
class RefMapper(object):
    def __init__(self, refs, preset, names=None, cigar=True, extra_flag=0):
        import minimap2_amd as mm  # struct mirrors only
        self.mm = mm
        R = self.R = C.CDLL(REF_SO)
        self.io, self.mo = mm.IdxOpt(), mm.MapOpt()
        R.mm_set_opt(None, C.byref(self.io), C.byref(self.mo))
        if preset is not None:
            assert R.mm_set_opt(preset.encode(), C.byref(self.io), C.byref(self.mo)) == 0
        if cigar:
            self.mo.flag |= mm.F_CIGAR
        self.mo.flag |= extra_flag
        R.mm_idx_str.restype = C.c_void_p
        R.mm_idx_str.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)]
        n = len(refs)
        nm = [(x.encode() if isinstance(x, str) else x) for x in (names or ["ref%d" % i for i in range(n)])]
        self._keep = (list(refs), nm)
        self.mi = R.mm_idx_str(self.io.w, self.io.k, self.io.flag & 1, self.io.bucket_bits, n, (C.c_char_p * n)(*refs), (C.c_char_p * n)(*nm))
        R.mm_mapopt_update.argtypes = [C.c_void_p, C.c_void_p]
        R.mm_mapopt_update(C.byref(self.mo), self.mi)
        R.mm_tbuf_init.restype = C.c_void_p
        R.mm_tbuf_destroy.argtypes = [C.c_void_p]
        R.mm_map.restype = C.POINTER(mm.Reg1)
        R.mm_map.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_char_p]
        R.mm_idx_destroy.argtypes = [C.c_void_p]
        self.tbuf = R.mm_tbuf_init()

    def map(self, name, seq):
        """-> list of Alignment.key()-shaped tuples"""
        mm = self.mm
        n = C.c_int(0)
        nb = name.encode() if isinstance(name, str) else name
        regs = self.R.mm_map(self.mi, len(seq), seq, C.byref(n), self.tbuf, C.byref(self.mo), nb)
        out = [a.key() for a in mm._regs_to_alignments(n.value, regs, None, None)]
        for j in range(n.value):
            if regs[j].p:
                _libc.free(C.cast(regs[j].p, C.c_void_p))
        if regs:
            _libc.free(C.cast(regs, C.c_void_p))
        return out

    def close(self):
        if self.mi:
            self.R.mm_tbuf_destroy(self.tbuf)
            self.R.mm_idx_destroy(self.mi)
            self.mi = None


def ref_map_reads(refs, reads, preset, names=None):
    m = RefMapper(refs, preset, names)
    out = [m.map(nm, s) for nm, s in reads]
    m.close()
    return out


# ---------------------------------------------------------------------------------------------------------
# oracle/_ref/librefdrv.so: adopt a flat minimizer table as a reference mm_idx_t, and a threaded mm_map loop
# ---------------------------------------------------------------------------------------------------------
REFDRV_SO = os.path.join(ORACLE_DIR, "_ref", "librefdrv.so")


class RefDriver(object):
    """The reference's mm_map over many reads on n_threads host threads, against an mm_idx_t adopted from flat tables
    (keys/val_off/pos as exported by mm2amd_idx_export or computed by any other means)."""

    def __init__(self, w, k, flag, names, lens, S, keys, val_off, pos, n_threads):
        import minimap2_amd as mm
        self.mm = mm
        D = self.D = C.CDLL(REFDRV_SO)
        D.refdrv_idx_from_flat.restype = C.c_void_p
        D.refdrv_idx_from_flat.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.POINTER(C.c_char_p), C.c_void_p, C.c_void_p,
                                           C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        D.refdrv_map.restype = C.c_double
        D.refdrv_map.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.c_int,
                                 C.POINTER(C.c_int), C.POINTER(C.c_void_p)]
        D.mm_idx_destroy.argtypes = [C.c_void_p]
        D.mm_mapopt_update.argtypes = [C.c_void_p, C.c_void_p]
        n = len(names)
        nm = [(x.encode() if isinstance(x, str) else x) for x in names]
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        self.n_threads = n_threads
        self.mi = D.refdrv_idx_from_flat(w, k, flag, 14, n, (C.c_char_p * n)(*nm), lens.ctypes.data, S.ctypes.data, len(keys), keys.ctypes.data,
                                         val_off.ctypes.data, pos.ctypes.data, n_threads)

    def map_opt(self, preset, cigar=True, extra_flag=0):
        mm = self.mm
        io, mo = mm.IdxOpt(), mm.MapOpt()
        self.D.mm_set_opt(None, C.byref(io), C.byref(mo))
        if preset is not None:
            assert self.D.mm_set_opt(preset.encode(), C.byref(io), C.byref(mo)) == 0
        if cigar:
            mo.flag |= mm.F_CIGAR
        mo.flag |= extra_flag
        self.D.mm_mapopt_update(C.byref(mo), self.mi)
        return mo

    def map(self, mo, reads, n_threads=None):
        """reads: list of (name, seq) -> (wall seconds of the mm_map loop, n_reg, reg) ; free with mm2amd_free_regs-like free_regs().
        A list of (name, seq1, seq2) is mapped as read pairs (mm_map_frag with two segments); n_reg / reg then hold 2 entries per pair."""
        n = len(reads)
        if n and len(reads[0]) == 3:
            D = self.D
            D.refdrv_map_pairs.restype = C.c_double
            D.refdrv_map_pairs.argtypes = D.refdrv_map.argtypes
            names = (C.c_char_p * n)(*[(r[0].encode() if isinstance(r[0], str) else r[0]) for r in reads])
            flat = [s for r in reads for s in r[1:]]
            seqs = (C.c_char_p * (2 * n))(*flat)
            lens = (C.c_int * (2 * n))(*[len(s) for s in flat])
            n_reg, reg = (C.c_int * (2 * n))(), (C.c_void_p * (2 * n))()
            t = D.refdrv_map_pairs(self.mi, C.byref(mo), n, seqs, lens, names, n_threads or self.n_threads, n_reg, reg)
            return t, n_reg, reg
        names = (C.c_char_p * n)(*[(r[0].encode() if isinstance(r[0], str) else r[0]) for r in reads])
        seqs = (C.c_char_p * n)(*[r[1] for r in reads])
        lens = (C.c_int * n)(*[len(r[1]) for r in reads])
        n_reg, reg = (C.c_int * n)(), (C.c_void_p * n)()
        t = self.D.refdrv_map(self.mi, C.byref(mo), n, seqs, lens, names, n_threads or self.n_threads, n_reg, reg)
        return t, n_reg, reg

    def close(self):
        if self.mi:
            self.D.mm_idx_destroy(self.mi)
            self.mi = None


def export_index(al):
    """numpy copies of an Aligner's device-built tables: (S, keys, val_off, pos)"""
    import minimap2_amd as mm
    st = al.index_stat()
    keys = np.zeros(st["n_distinct"], np.uint64)
    val_off = np.zeros(st["n_distinct"] + 1, np.uint32)
    pos = np.zeros(st["n_minimizers"], np.uint64)
    S = np.zeros((st["sum_len"] + 7) // 8, np.uint32)
    rc = mm.lib().mm2amd_idx_export(al._idx, None, keys.ctypes.data, val_off.ctypes.data, pos.ctypes.data, S.ctypes.data)
    assert rc == 0, mm.lib().mm2amd_last_error()
    return S, keys, val_off, pos


REFALIGN_SO = os.path.join(os.path.dirname(REF_SO), "librefalign.so")  # oracle/ref_align_shim.c: the reference's static mm_update_extra behind one entry point


def ref_update_extra(qs, ts, pieces, mat, q, e, log_gap, eqx=False):
    """the UNMODIFIED reference's mm_append_cigar + mm_fix_cigar + mm_update_extra (align.c:320-334, :105-181, :254-303) on a region given as its windows' CIGARs;
    returns (cigar_tuple, blen, mlen, n_ambi, dp_max, qshift, tshift, is_spliced), the layout of minimap2_amd.update_extra_batch"""
    L = C.CDLL(REFALIGN_SO)
    qb, tb = bytes(qs), bytes(ts)
    parr = [(C.c_uint32 * max(len(p), 1))(*p) for p in pieces]
    pp = (C.POINTER(C.c_uint32) * max(len(pieces), 1))(*[C.cast(a, C.POINTER(C.c_uint32)) for a in parr])
    pl = (C.c_int32 * max(len(pieces), 1))(*[len(p) for p in pieces])
    tot = sum(len(p) for p in pieces)
    out = (C.c_uint32 * max(tot, 1))()
    res = (C.c_int32 * 8)()
    if eqx:  # MM_F_EQX: mm_update_cigar_eqx on top (the match stretches can outnumber the operations that came in)
        cap = tot + len(qb) + 8
        out = (C.c_uint32 * cap)()
        L.refshim_update_extra_eqx.restype = C.c_int
        n = L.refshim_update_extra_eqx(len(pieces), pp, pl, len(qb), qb, len(tb), tb, bytes(mat), q, e, 1 if log_gap else 0, out, cap, res)
        assert n >= 0
        return (tuple(out[:n]), res[0], res[1], res[2], res[3], res[4], res[5], res[6])
    L.refshim_update_extra.restype = C.c_int
    n = L.refshim_update_extra(len(pieces), pp, pl, len(qb), qb, len(tb), tb, bytes(mat), q, e, 1 if log_gap else 0, out, res)
    return (tuple(out[:n]), res[0], res[1], res[2], res[3], res[4], res[5], res[6])
