"""minimap2_amd -- MI355X-native seed-chain-extend engine behind minimap2's API.

This package is a thin ctypes mirror of the C ABI in include/mm2amd.h, shaped like the reference's own Python
binding (python/mappy.pyx: Aligner(seq=..., preset=...).map(...)).  All compute happens in libmm2amd.so
(hand-written HIP for gfx950).  There is no CPU fallback: if the library is not built or no GPU is visible,
calls raise Mm2AmdError."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmm2amd.so")


class Mm2AmdError(RuntimeError):
    pass


# ---------------------------------------------------------------------------------------------------------
# struct mirrors (layouts: minimap2_amd/csrc/abi_ref.hpp, checked against the reference headers in tests/)
# ---------------------------------------------------------------------------------------------------------
class KswJob(C.Structure):  # mm2amd_ksw_job_t
    _fields_ = [("query", C.c_void_p), ("target", C.c_void_p), ("qlen", C.c_int32), ("tlen", C.c_int32), ("w", C.c_int32),
                ("zdrop", C.c_int32), ("end_bonus", C.c_int32), ("flag", C.c_int32)]


class FinJob(C.Structure):  # mm2amd_fin_job_t
    _fields_ = [("query", C.c_void_p), ("target", C.c_void_p), ("qlen", C.c_int32), ("tlen", C.c_int32), ("n_pieces", C.c_int32),
                ("piece", C.POINTER(C.POINTER(C.c_uint32))), ("piece_len", C.POINTER(C.c_int32))]


class FinRes(C.Structure):  # mm2amd_fin_res_t
    _fields_ = [("n_cigar", C.c_int32), ("blen", C.c_int32), ("mlen", C.c_int32), ("n_ambi", C.c_int32), ("dp_max", C.c_int32), ("qshift", C.c_int32),
                ("tshift", C.c_int32), ("is_spliced", C.c_int32), ("cigar_off", C.c_uint32)]


class KswRes(C.Structure):  # mm2amd_ksw_res_t
    _fields_ = [(n, C.c_int32) for n in ("max", "zdropped", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q", "score",
                                         "n_cigar", "reach_end")] + [("cigar_off", C.c_uint32)]


class IdxOpt(C.Structure):  # mm_idxopt_t, minimap.h:130-134
    _fields_ = [("k", C.c_short), ("w", C.c_short), ("flag", C.c_short), ("bucket_bits", C.c_short),
                ("mini_batch_size", C.c_int64), ("batch_size", C.c_uint64)]


class MapOpt(C.Structure):  # mm_mapopt_t, minimap.h:136-192
    _fields_ = [("flag", C.c_int64), ("seed", C.c_int), ("sdust_thres", C.c_int), ("max_qlen", C.c_int), ("bw", C.c_int),
                ("bw_long", C.c_int), ("max_gap", C.c_int), ("max_gap_ref", C.c_int), ("max_frag_len", C.c_int),
                ("max_chain_skip", C.c_int), ("max_chain_iter", C.c_int), ("min_cnt", C.c_int), ("min_chain_score", C.c_int),
                ("chain_gap_scale", C.c_float), ("chain_skip_scale", C.c_float), ("rmq_size_cap", C.c_int),
                ("rmq_inner_dist", C.c_int), ("rmq_rescue_size", C.c_int), ("rmq_rescue_ratio", C.c_float),
                ("mask_level", C.c_float), ("mask_len", C.c_int), ("pri_ratio", C.c_float), ("best_n", C.c_int),
                ("alt_drop", C.c_float), ("a", C.c_int), ("b", C.c_int), ("q", C.c_int), ("e", C.c_int), ("q2", C.c_int),
                ("e2", C.c_int), ("transition", C.c_int), ("sc_ambi", C.c_int), ("noncan", C.c_int), ("junc_bonus", C.c_int),
                ("junc_pen", C.c_int), ("zdrop", C.c_int), ("zdrop_inv", C.c_int), ("end_bonus", C.c_int),
                ("min_dp_max", C.c_int), ("min_ksw_len", C.c_int), ("anchor_ext_len", C.c_int), ("anchor_ext_shift", C.c_int),
                ("max_clip_ratio", C.c_float), ("rank_min_len", C.c_int), ("rank_frac", C.c_float), ("pe_ori", C.c_int),
                ("pe_bonus", C.c_int), ("jump_min_match", C.c_int32), ("mid_occ_frac", C.c_float), ("q_occ_frac", C.c_float),
                ("min_mid_occ", C.c_int32), ("max_mid_occ", C.c_int32), ("mid_occ", C.c_int32), ("max_occ", C.c_int32),
                ("max_max_occ", C.c_int32), ("occ_dist", C.c_int32), ("mini_batch_size", C.c_int64), ("max_sw_mat", C.c_int64),
                ("cap_kalloc", C.c_int64), ("split_prefix", C.c_char_p)]


class Extra(C.Structure):  # mm_extra_t header, minimap.h:103-110 (cigar[] follows)
    _fields_ = [("capacity", C.c_uint32), ("dp_score", C.c_int32), ("dp_max", C.c_int32), ("dp_max2", C.c_int32),
                ("dp_max0", C.c_int32), ("n_ambi_strand", C.c_uint32), ("n_cigar", C.c_uint32)]


class Reg1(C.Structure):  # mm_reg1_t, minimap.h:112-127
    _fields_ = [("id", C.c_int32), ("cnt", C.c_int32), ("rid", C.c_int32), ("score", C.c_int32), ("qs", C.c_int32),
                ("qe", C.c_int32), ("rs", C.c_int32), ("re", C.c_int32), ("parent", C.c_int32), ("subsc", C.c_int32),
                ("as_", C.c_int32), ("mlen", C.c_int32), ("blen", C.c_int32), ("n_sub", C.c_int32), ("score0", C.c_int32),
                ("bits", C.c_uint32), ("hash", C.c_uint32), ("div", C.c_float), ("p", C.POINTER(Extra))]

    mapq = property(lambda s: s.bits & 0xff)
    split = property(lambda s: s.bits >> 8 & 3)
    rev = property(lambda s: s.bits >> 10 & 1)
    inv = property(lambda s: s.bits >> 11 & 1)
    sam_pri = property(lambda s: s.bits >> 12 & 1)


class Bseq1(C.Structure):  # mm_bseq1_t, bseq.h:14-17
    _fields_ = [("l_seq", C.c_int), ("rid", C.c_int), ("name", C.c_char_p), ("seq", C.c_char_p), ("qual", C.c_char_p),
                ("comment", C.c_char_p)]


class KernelStat(C.Structure):  # mm2amd_kernel_stat_t
    _fields_ = [("name", C.c_char * 48), ("ms", C.c_double), ("alg_bytes", C.c_double), ("launches", C.c_int64), ("units", C.c_double)]


assert C.sizeof(MapOpt) == 264 and C.sizeof(Reg1) == 80 and C.sizeof(Extra) == 28 and C.sizeof(Bseq1) == 40 and C.sizeof(IdxOpt) == 24

F_CIGAR, F_OUT_SAM = 0x004, 0x008  # MM_F_CIGAR, MM_F_OUT_SAM (minimap.h:12-13)

_lib = None


def _bind(L):
    vp, ip = C.c_void_p, C.POINTER(C.c_int)
    L.mm2amd_last_error.restype = C.c_char_p
    L.mm2amd_backend_name.restype = C.c_char_p
    L.mm_gpu_init.argtypes = [vp, vp, C.c_int]
    L.mm_gpu_map_batch.argtypes = [C.c_int, ip, ip, vp, ip, C.POINTER(vp), ip, ip]
    L.mm_gpu_batch_stage.argtypes = [C.c_int, ip, ip, vp]
    L.mm_gpu_map_staged.argtypes = [ip, C.POINTER(vp), ip, ip]
    L.mm_gpu_batch_stage_queued.argtypes = [C.c_int, ip, ip, vp]
    L.mm_gpu_batch_discard.restype = None
    L.mm2amd_free_regs.argtypes = [C.c_int, ip, C.POINTER(vp)]
    L.mm2amd_free_regs.restype = None
    L.mm2amd_last_stats.argtypes = [C.POINTER(C.c_double), C.c_int]
    L.mm2amd_pack_regs.argtypes = [C.c_int, ip, C.POINTER(vp), vp, C.c_int64]
    L.mm2amd_pack_regs.restype = C.c_int64
    L.mm2amd_unpack_regs.argtypes = [vp, C.c_int64, C.c_int, ip, C.POINTER(vp)]
    L.mm_gpu_init_multi.argtypes = [vp, vp, C.c_int, C.c_int, ip]
    L.mm_gpu_context_generation.restype = C.c_uint64
    L.mm_gpu_destroy_if.argtypes = [C.c_uint64]
    if hasattr(L, "mm2amd_idx_str"):  # the product library (the CPU check library used by tests has no device index)
        L.mm2amd_ksw_extd2_batch.restype = C.c_int
        L.mm2amd_ksw_extd2_batch.argtypes = [C.c_int, C.POINTER(KswJob), C.c_int8, C.c_char_p, C.c_int8, C.c_int8, C.c_int8,
                                             C.c_int8, C.POINTER(KswRes), C.POINTER(C.c_uint32), C.c_size_t]
        L.mm2amd_ksw_extz2_batch.restype = C.c_int
        L.mm2amd_ksw_extz2_batch.argtypes = [C.c_int, C.POINTER(KswJob), C.c_int8, C.c_char_p, C.c_int8, C.c_int8, C.POINTER(KswRes),
                                             C.POINTER(C.c_uint32), C.c_size_t]
        L.mm2amd_ksw_exts2_batch.restype = C.c_int
        L.mm2amd_ksw_exts2_batch.argtypes = [C.c_int, C.POINTER(KswJob), C.c_int8, C.c_char_p, C.c_int8, C.c_int8, C.c_int8, C.c_int8,
                                             C.POINTER(KswRes), C.POINTER(C.c_uint32), C.c_size_t]
        L.mm2amd_update_extra_batch.restype = C.c_int
        L.mm2amd_update_extra_batch.argtypes = [C.c_int, C.POINTER(FinJob), C.c_char_p, C.c_int8, C.c_int8, C.c_int, C.POINTER(FinRes), C.POINTER(C.c_uint32), C.c_size_t]
        L.mm2amd_sort_pairs_u64.argtypes = [vp, vp, C.c_uint64, C.c_int]
        L.mm2amd_exclusive_sum_u32.argtypes = [vp, vp, C.c_uint64]
        L.mm2amd_idx_str.restype = vp
        L.mm2amd_idx_str.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)]
        L.mm2amd_idx_destroy.argtypes = [vp]
        L.mm2amd_idx_destroy.restype = None
        L.mm2amd_idx_stat.argtypes = [vp, ip, ip, ip, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                      C.POINTER(C.c_uint64)]
        L.mm2amd_idx_cal_max_occ.argtypes = [vp, C.c_float]
        L.mm2amd_idx_cal_max_occ.restype = C.c_int32
        L.mm2amd_mapopt_update.argtypes = [vp, vp]
        L.mm2amd_idx_table_shape.argtypes = [vp, ip, ip]
        L.mm2amd_idx_export.argtypes = [vp, vp, vp, vp, vp, vp]
        L.mm_gpu_init_index.argtypes = [vp, vp, C.c_int]
        L.mm_gpu_init_index_multi.argtypes = [vp, vp, C.c_int, C.c_int, ip]
        L.mm2amd_set_opt.argtypes = [C.c_char_p, vp, vp]
        L.mm2amd_check_opt.argtypes = [vp, vp]
        L.mm2amd_idxopt_init.argtypes = [vp]
        L.mm2amd_mapopt_init.argtypes = [vp]
        L.mm_gpu_format_batch.argtypes = [C.c_int, ip, ip, vp, ip, C.POINTER(vp), ip, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.mm_gpu_format_batch_view.argtypes = [C.c_int, ip, ip, vp, ip, C.POINTER(vp), ip, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.mm2amd_profile_enable.argtypes = [C.c_int]
        L.mm2amd_profile_enable.restype = None
        L.mm2amd_profile_get.argtypes = [C.POINTER(KernelStat), C.c_int]
    return L


def host_cpus():
    """CPUs this process may use: hardware threads capped by the container's CPU quota"""
    return int(lib().mm2amd_host_cpus())


def _libc_free(p):
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free(p)


def lib(path=None):
    """Load libmm2amd.so (built in-tree by minimap2_amd.build); raises if it is missing."""
    global _lib
    if path is not None:
        return _bind(C.CDLL(path))
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Mm2AmdError("libmm2amd.so is not built (run `python -m minimap2_amd.build`); there is no CPU fallback")
        _lib = _bind(C.CDLL(LIB_PATH))
    return _lib


def _check(rc, L=None):
    if rc != 0:
        raise Mm2AmdError("mm2amd error %d: %s" % (rc, (L or lib()).mm2amd_last_error().decode()))


def update_extra_batch(jobs, mat, q, e, log_gap):
    """jobs: list of (query_codes, target_codes, [piece, ...]) -- nt4 codes 0..4 of the aligned stretches and the windows' CIGARs (sequences of
    len << 4 | op) in alignment order.  Returns a list of (cigar_tuple, blen, mlen, n_ambi, dp_max, qshift, tshift, is_spliced) -- what the
    reference's mm_update_extra (align.c:254-303) leaves in the hit record -- or None for a region whose operations do not cover its stretches."""
    n = len(jobs)
    arr = (FinJob * max(n, 1))()
    keep, tot = [], 0
    for i, (qs, ts, pieces) in enumerate(jobs):
        qb, tb = bytes(qs), bytes(ts)
        parr = [(C.c_uint32 * max(len(p_), 1))(*p_) for p_ in pieces]
        pp = (C.POINTER(C.c_uint32) * max(len(pieces), 1))(*[C.cast(a, C.POINTER(C.c_uint32)) for a in parr])
        pl = (C.c_int32 * max(len(pieces), 1))(*[len(p_) for p_ in pieces])
        keep.append((qb, tb, parr, pp, pl))
        arr[i].query, arr[i].target = C.cast(C.c_char_p(qb), C.c_void_p), C.cast(C.c_char_p(tb), C.c_void_p)
        arr[i].qlen, arr[i].tlen, arr[i].n_pieces, arr[i].piece, arr[i].piece_len = len(qb), len(tb), len(pieces), pp, pl
        tot += sum(len(p_) for p_ in pieces)
    res = (FinRes * max(n, 1))()
    pool = (C.c_uint32 * max(tot, 1))()
    _check(lib().mm2amd_update_extra_batch(n, arr, bytes(mat), q, e, 1 if log_gap else 0, res, pool, max(tot, 1)))
    out = []
    for i in range(n):
        r = res[i]
        out.append(None if r.n_cigar < 0 else (tuple(pool[r.cigar_off:r.cigar_off + r.n_cigar]), r.blen, r.mlen, r.n_ambi, r.dp_max, r.qshift, r.tshift, r.is_spliced))
    return out


def sort_pairs_u64(keys, vals, bits=64):
    """(keys, vals) -- numpy uint64 arrays of equal length -- sorted by key bits [0, bits), stably, on the device (the index build's sort,
    device_sort.hip).  Returns new arrays."""
    import numpy as np
    k, v = np.ascontiguousarray(keys, dtype=np.uint64).copy(), np.ascontiguousarray(vals, dtype=np.uint64).copy()
    assert k.shape == v.shape and k.ndim == 1
    _check(lib().mm2amd_sort_pairs_u64(k.ctypes.data, v.ctypes.data, k.size, bits))
    return k, v


def exclusive_sum_u32(a):
    """numpy uint32 array of n entries -> n + 1 running sums mod 2^32, on the device (device_sort.hip)"""
    import numpy as np
    x = np.ascontiguousarray(a, dtype=np.uint32)
    out = np.empty(x.size + 1, dtype=np.uint32)
    _check(lib().mm2amd_exclusive_sum_u32(x.ctypes.data, out.ctypes.data, x.size))
    return out


def ksw_extz2_batch(jobs, mat, gapo, gape):
    """single-affine twin of ksw_extd2_batch (ksw_extz2_sse)"""
    return ksw_extd2_batch(jobs, mat, gapo, gape, None, None)


def ksw_exts2_batch(jobs, mat, gapo, gape, gapo2, noncan):
    """splice-aware twin of ksw_extd2_batch (ksw_exts2_sse with junc == NULL); a job's w is ignored"""
    return ksw_extd2_batch(jobs, mat, gapo, gape, gapo2, None, noncan=noncan)


def ksw_extd2_batch(jobs, mat, gapo, gape, gapo2, gape2, noncan=None):
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
    if noncan is not None:
        _check(lib().mm2amd_ksw_exts2_batch(n, arr, 5, bytes(mat), gapo, gape, gapo2, noncan, res, pool, max(tot, 1)))
    elif gapo2 is None:
        _check(lib().mm2amd_ksw_extz2_batch(n, arr, 5, bytes(mat), gapo, gape, res, pool, max(tot, 1)))
    else:
        _check(lib().mm2amd_ksw_extd2_batch(n, arr, 5, bytes(mat), gapo, gape, gapo2, gape2, res, pool, max(tot, 1)))
    out = []
    for r in res:
        out.append((r.max, r.zdropped, r.max_q, r.max_t, r.mqe, r.mqe_t, r.mte, r.mte_q, r.score, r.reach_end,
                    tuple(pool[r.cigar_off:r.cigar_off + r.n_cigar])))
    return out


# ---------------------------------------------------------------------------------------------------------
# mappy-shaped front end
# ---------------------------------------------------------------------------------------------------------
class Alignment(object):
    """One hit; field names follow mappy.Alignment (python/mappy.pyx:30-90)."""
    __slots__ = ("ctg", "ctg_len", "r_st", "r_en", "q_st", "q_en", "strand", "mapq", "is_primary", "mlen", "blen", "NM",
                 "cigar", "score", "dp_score", "rid", "sam_pri", "div", "parent", "id")

    @property
    def cigar_str(self):
        return "".join("%d%s" % (c >> 4, "MIDNSHP=XB"[c & 0xf]) for c in self.cigar)

    def key(self):
        """Everything the SAM/PAF writer reads from mm_reg1_t for this hit, as a comparable tuple."""
        return (self.rid, self.r_st, self.r_en, self.q_st, self.q_en, self.strand, self.mapq, self.is_primary, self.sam_pri,
                self.mlen, self.blen, self.score, self.dp_score, self.parent, self.id, tuple(self.cigar))


def _regs_to_alignments(n, regs, names, lens):
    out = []
    for j in range(n):
        r = regs[j]
        a = Alignment()
        a.rid, a.ctg, a.ctg_len = r.rid, names[r.rid] if names else None, lens[r.rid] if lens else None
        a.r_st, a.r_en, a.q_st, a.q_en = r.rs, r.re, r.qs, r.qe
        a.strand = -1 if r.rev else 1
        a.mapq, a.is_primary, a.sam_pri = r.mapq, int(r.id == r.parent), r.sam_pri
        a.mlen, a.blen, a.score, a.div, a.parent, a.id = r.mlen, r.blen, r.score, r.div, r.parent, r.id
        if r.p:
            ex = r.p.contents
            a.dp_score = ex.dp_score
            cig = C.cast(C.addressof(ex) + C.sizeof(Extra), C.POINTER(C.c_uint32 * ex.n_cigar)).contents if ex.n_cigar else ()
            a.cigar = list(cig)
            a.NM = r.blen - r.mlen + (ex.n_ambi_strand & 0x3fffffff)
        else:
            a.dp_score, a.cigar, a.NM = 0, [], 0
        out.append(a)
    return out


class Batch(object):
    """A mini-batch of reads as the reference's reader hands it to the mapping step: an array of mm_bseq1_t records over host buffers
    (bseq.h:14-17) plus the fragment table (seg_off, n_seg; map.c:560-575).  reads: list of (name, sequence), of sequences, or of
    (name, sequence1, sequence2) for read pairs.  Building one is the reader's work, not part of the hand-over."""

    def __init__(self, reads=None):
        self.n = 0
        self.items, self.arr, self.seg_off, self.n_seg = [], None, None, None
        if reads is None:
            return
        self.n = len(reads)
        seg_off, n_seg = [], []
        for i, r in enumerate(reads):
            r = r if isinstance(r, tuple) else ("read%d" % i, r)
            nb = r[0].encode() if isinstance(r[0], str) else r[0]
            seg_off.append(len(self.items)), n_seg.append(len(r) - 1)
            for q in r[1:]:
                self.items.append((nb, q.encode() if isinstance(q, str) else bytes(q)))
        self.arr = (Bseq1 * max(1, len(self.items)))()
        for k, (nb, sb) in enumerate(self.items):
            self.arr[k].l_seq, self.arr[k].rid, self.arr[k].name, self.arr[k].seq = len(sb), k, nb, sb
        self.seg_off, self.n_seg = (C.c_int * max(1, self.n))(*seg_off), (C.c_int * max(1, self.n))(*n_seg)
        self.bases = sum(len(sb) for _, sb in self.items)

    def rotated(self, k):
        """the same reads starting at fragment k (fragments k.., then 0..k-1): new record and fragment tables over the SAME sequence
        buffers, made with two memmoves -- what a reader producing that order would have handed over"""
        if self.n == 0 or k % self.n == 0:
            return self
        k %= self.n
        b = Batch()
        b.n, b.items, b.bases = self.n, self.items, self.bases  # (items keeps the byte strings alive)
        m = len(self.items)
        first = self.seg_off[k]  # records of fragments k.. start here
        b.arr = (Bseq1 * max(1, m))()
        sz = C.sizeof(Bseq1)
        C.memmove(b.arr, C.byref(self.arr, first * sz), (m - first) * sz)
        C.memmove(C.byref(b.arr, (m - first) * sz), self.arr, first * sz)
        import numpy as np
        ns = np.frombuffer(self.n_seg, dtype=np.int32, count=self.n)
        ns = np.concatenate([ns[k:], ns[:k]])
        so = np.concatenate([[0], np.cumsum(ns)[:-1]]).astype(np.int32)
        b.n_seg, b.seg_off = (C.c_int * self.n).from_buffer_copy(ns.tobytes()), (C.c_int * self.n).from_buffer_copy(so.tobytes())
        return b


class Aligner(object):
    """Index built on the GPU from in-memory sequences + batched mapping; mirrors mappy.Aligner(seq=..., preset=...).

    seq: a sequence string/bytes or a list of them (the reference); names: optional list of contig names.
    n_gpus / device_ids: map every batch on several GPUs of this process (mm_gpu_init_index_multi: index replicated, reads sharded
    by bases; an ordinal may repeat).  Only one Aligner can be the active mapper of the process at a time (the drop-in boundary is
    a process-wide context, like the reference's pipeline): creating a second one makes the first inactive -- its map calls raise,
    and closing or collecting it leaves the new context alone."""

    def __init__(self, seq, preset=None, names=None, k=None, w=None, n_threads=0, cigar=True, sam=False, n_gpus=0, device_ids=None):
        L = lib()
        self._generation, self._idx, self._staged = 0, None, None  # close() must work on a half-built object
        seqs = [seq] if isinstance(seq, (bytes, str)) else list(seq)
        self._seqs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
        n = len(self._seqs)
        self.names = [x if isinstance(x, str) else x.decode() for x in names] if names else ["ref%d" % i for i in range(n)]
        self.lens = [len(s) for s in self._seqs]
        self.idx_opt, self.map_opt = IdxOpt(), MapOpt()
        L.mm2amd_set_opt(None, C.byref(self.idx_opt), C.byref(self.map_opt))
        if preset is not None and L.mm2amd_set_opt(preset.encode(), C.byref(self.idx_opt), C.byref(self.map_opt)) != 0:
            raise Mm2AmdError("unknown preset %r" % preset)
        if k:
            self.idx_opt.k = k
        if w:
            self.idx_opt.w = w
        if cigar:
            self.map_opt.flag |= F_CIGAR
        if sam:
            self.map_opt.flag |= F_OUT_SAM | F_CIGAR
        _check(L.mm2amd_check_opt(C.byref(self.idx_opt), C.byref(self.map_opt)))
        sarr = (C.c_char_p * n)(*self._seqs)
        self._name_bytes = [x.encode() for x in self.names]
        narr = (C.c_char_p * n)(*self._name_bytes)
        self._idx = L.mm2amd_idx_str(self.idx_opt.w, self.idx_opt.k, self.idx_opt.flag & 1, self.idx_opt.bucket_bits, n, sarr, narr)
        if not self._idx:
            raise Mm2AmdError("index construction failed: %s" % L.mm2amd_last_error().decode())
        _check(L.mm2amd_mapopt_update(C.byref(self.map_opt), self._idx))
        ids = (C.c_int * len(device_ids))(*device_ids) if device_ids else None
        _check(L.mm_gpu_init_index_multi(self._idx, C.byref(self.map_opt), n_threads, len(device_ids) if device_ids else n_gpus, ids))
        self._generation = L.mm_gpu_context_generation()
        self._staged = None

    def _active(self):
        if lib().mm_gpu_context_generation() != self._generation:
            raise Mm2AmdError("this Aligner is no longer the process's active mapper (a later Aligner or mm_gpu_init replaced its context)")

    def close(self):
        if getattr(self, "_idx", None):
            lib().mm_gpu_destroy_if(self._generation)  # only the context this object installed
            lib().mm2amd_idx_destroy(self._idx)
            self._idx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def index_stat(self):
        k, w, flag, n_seq = C.c_int(), C.c_int(), C.c_int(), C.c_uint32()
        sl, nd, nm = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(lib().mm2amd_idx_stat(self._idx, k, w, flag, n_seq, sl, nd, nm))
        return {"k": k.value, "w": w.value, "flag": flag.value, "n_seq": n_seq.value, "sum_len": sl.value,
                "n_distinct": nd.value, "n_minimizers": nm.value}

    # -- batch interface: stage() + run() == map_batch() ------------------------------------------------
    def stage(self, reads):
        """reads: list of (name, sequence), of sequences (bytes/str), or of (name, sequence1, sequence2) for read pairs (mapped as
        two-segment fragments, mm_map_frag with n_segs == 2), or a Batch.  Copies them to the GPU."""
        self._active()
        b = reads if isinstance(reads, Batch) else Batch(reads)
        _check(lib().mm_gpu_batch_stage(b.n, b.seg_off, b.n_seg, b.arr))
        self._staged = (b.n, b.arr, b.items, b.seg_off, b.n_seg)

    def pipeline(self, batches, text=True, on_mapped=None, on_text=None, trace=None):
        """The reference's three-step pipeline (map.c:541-643) over an iterable of Batch objects, every step on a thread of its own so
        that the steps of successive batches overlap: hand-over (mm_gpu_batch_stage_queued: pack + H2D beside the mapping of the batch
        before), mapping (mm_gpu_map_staged), output stage (mm_gpu_format_batch_view: SAM / PAF text of the batch in a reused buffer).
        on_mapped(batch, n_reg, reg, rep_len) runs on the output thread before formatting (e.g. the multi-GPU hit gather);
        on_text(batch, address, length) receives the text (valid until the next batch's).  Hit records are freed after on_text.
        trace: a list that receives (step, batch number, start, end) wall-clock intervals of the three steps.
        Returns the number of text bytes produced."""
        import queue
        import threading
        import time
        self._active()
        L = lib()
        q_staged, q_mapped = queue.Queue(maxsize=2), queue.Queue(maxsize=2)
        errors, total = [], [0]

        def stager():
            try:
                for k, b in enumerate(batches):
                    if errors:
                        break
                    t0 = time.time()
                    _check(L.mm_gpu_batch_stage_queued(b.n, b.seg_off, b.n_seg, b.arr))
                    if trace is not None:
                        trace.append(("stage", k, t0, time.time()))
                    q_staged.put(b)
            except Exception as e:  # noqa: BLE001
                errors.append(e)
            q_staged.put(None)

        def mapper():
            try:
                k = 0
                while True:
                    b = q_staged.get()
                    if b is None:
                        break
                    m = max(1, len(b.items))
                    n_reg, reg, rep_len, frag_gap = (C.c_int * m)(), (C.c_void_p * m)(), (C.c_int * m)(), (C.c_int * m)()
                    t0 = time.time()
                    _check(L.mm_gpu_map_staged(n_reg, reg, rep_len, frag_gap))
                    if trace is not None:
                        trace.append(("map", k, t0, time.time()))
                    k += 1
                    q_mapped.put((b, n_reg, reg, rep_len))
            except Exception as e:  # noqa: BLE001
                errors.append(e)
                while True:  # let the stager finish: its staged batches are dropped
                    L.mm_gpu_batch_discard()
                    try:
                        if q_staged.get(timeout=0.05) is None:
                            break
                    except queue.Empty:
                        pass
            q_mapped.put(None)

        def output():
            k = -1
            while True:
                it = q_mapped.get()
                if it is None:
                    break
                b, n_reg, reg, rep_len = it
                k += 1
                t0 = time.time()
                try:
                    if not errors:
                        if on_mapped:
                            on_mapped(b, n_reg, reg, rep_len)
                        if text:
                            out, out_len = C.c_void_p(), C.c_size_t()
                            _check(L.mm_gpu_format_batch_view(b.n, b.seg_off, b.n_seg, b.arr, n_reg, reg, rep_len, C.byref(out), C.byref(out_len)))
                            total[0] += out_len.value
                            if on_text:
                                on_text(b, out.value, out_len.value)
                except Exception as e:  # noqa: BLE001
                    errors.append(e)
                t1 = time.time()
                L.mm2amd_free_regs(len(n_reg), n_reg, reg)
                if trace is not None:
                    trace.append(("output", k, t0, t1)), trace.append(("free", k, t1, time.time()))

        def named(f, nm):  # the OS-level thread name (/proc/<pid>/task/*/comm): bench.py attributes CPU seconds by it
            def g():
                try:
                    C.CDLL(None).prctl(15, nm, 0, 0, 0)  # PR_SET_NAME
                except Exception:  # noqa: BLE001
                    pass
                f()
            return g
        th = [threading.Thread(target=named(f, nm)) for f, nm in ((stager, b"mm2-stager"), (mapper, b"mm2-mapper"), (output, b"mm2-output"))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        self._staged = None
        if errors:
            raise errors[0]
        return total[0]

    def run(self, raw=False):
        """Maps the staged batch.  Returns one list of alignments per read, or a pair of lists for a read pair.  raw=True returns
        (n_reg, reg, rep_len) ctypes arrays (one entry per read, pairs adjacent) that must be passed to free_raw()."""
        if self._staged is None:
            raise Mm2AmdError("run() without stage()")
        self._active()
        n, _, items, seg_off, n_seg = self._staged
        m = max(1, len(items))
        n_reg, reg, rep_len, frag_gap = (C.c_int * m)(), (C.c_void_p * m)(), (C.c_int * m)(), (C.c_int * m)()
        _check(lib().mm_gpu_map_staged(n_reg, reg, rep_len, frag_gap))
        if raw:
            return n_reg, reg, rep_len
        out = []
        for i in range(n):
            per = []
            for k in range(seg_off[i], seg_off[i] + n_seg[i]):
                regs = C.cast(reg[k], C.POINTER(Reg1)) if n_reg[k] else None
                per.append(_regs_to_alignments(n_reg[k], regs, self.names, self.lens))
            out.append(per[0] if n_seg[i] == 1 else tuple(per))
        lib().mm2amd_free_regs(len(items), n_reg, reg)
        return out

    def free_raw(self, n_reg, reg):
        lib().mm2amd_free_regs(len(n_reg), n_reg, reg)

    def format_raw(self, n_reg, reg, rep_len=None):
        """SAM (or PAF, by the aligner's options) records of the staged batch's raw results, as bytes: what the reference's
        output step writes for these reads (mm_gpu_format_batch)."""
        n, arr, _, seg_off, n_seg = self._staged
        out, out_len = C.c_void_p(), C.c_size_t()
        _check(lib().mm_gpu_format_batch(n, seg_off, n_seg, arr, n_reg, reg, rep_len, C.byref(out), C.byref(out_len)))
        try:
            return C.string_at(out, out_len.value)
        finally:
            _libc_free(out)

    def map_batch(self, reads):
        self.stage(reads)
        return self.run()

    def map(self, seq, seq2=None, name="query"):
        """Single-read (or, with seq2, single-pair) convenience wrapper (mappy.Aligner.map); a batch of one."""
        if seq2 is not None:
            return self.map_pairs([(name, seq, seq2)])[0]
        return self.map_batch([(name, seq)])[0]

    def map_pairs(self, pairs, text=False):
        """Paired-end reads: pairs = list of (name, seq1, seq2) (use preset "sr").  Returns a list of (alignments of read 1,
        alignments of read 2); with text=True the SAM/PAF records of the batch instead (mm_gpu_format_batch, mate fields included)."""
        self.stage(pairs)
        if not text:
            return self.run()
        n_reg, reg, rep_len = self.run(raw=True)
        try:
            return self.format_raw(n_reg, reg, rep_len)
        finally:
            self.free_raw(n_reg, reg)

    def last_stats(self):
        v = (C.c_double * 40)()
        k = lib().mm2amd_last_stats(v, 40)
        names = ["t_seed_chain", "t_host_pre", "t_plan", "t_ksw", "t_consume", "t_finish", "n_jobs", "n_rounds", "dp_cells", "dev_allocs", "pin_allocs",
                 "alloc_ns", "cpu_seed_chain", "cpu_host_pre", "cpu_plan", "cpu_ksw", "cpu_consume", "cpu_finish", "n_long_join_dev", "n_long_join_host",
                 "drv_cpu_seed_chain", "drv_cpu_host_pre", "drv_cpu_plan", "drv_cpu_ksw", "drv_cpu_consume", "drv_cpu_finish", "n_early_sub",
                 "n_region_reads_dev", "n_region_reads_host", "n_band128", "n_band256", "n_band_widened", "n_band_rectangle",
                 "arena_dev_bytes", "arena_pin_bytes", "arena_dev_used", "arena_pin_used", "n_band512", "n_band_rectangle_big"]
        return dict(zip(names, list(v)[:k]))


def band_counters():
    """The banded gap-fill kernel's windows since the process started (ksw_band.hip): tried in a band of 128 / 256 diagonals, sent on to the wider band,
    computed again as full rectangles.  (Zeros in a library without the kernels.)"""
    L = lib()
    if not hasattr(L, "mm2amd_alloc_counter"):
        return {"band128": 0, "band256": 0, "widened": 0, "rectangle": 0, "band512": 0, "rectangle_big": 0}
    L.mm2amd_alloc_counter.restype = C.c_longlong
    L.mm2amd_alloc_counter.argtypes = [C.c_int]
    return {"band128": L.mm2amd_alloc_counter(3), "band256": L.mm2amd_alloc_counter(4), "widened": L.mm2amd_alloc_counter(5), "rectangle": L.mm2amd_alloc_counter(6),
            "band512": L.mm2amd_alloc_counter(11), "rectangle_big": L.mm2amd_alloc_counter(12)}


def profile_enable(on=True):
    lib().mm2amd_profile_enable(1 if on else 0)


def profile_get():
    arr = (KernelStat * 64)()
    n = lib().mm2amd_profile_get(arr, 64)
    return {arr[i].name.decode(): {"ms": arr[i].ms, "alg_bytes": arr[i].alg_bytes, "launches": arr[i].launches, "units": arr[i].units} for i in range(n)}
