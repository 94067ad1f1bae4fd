"""The reference driver for read pairs (oracle/refdrv.c: refdrv_map_pairs, what bench.py's cpu_baseline uses for --preset sr) and
the drop-in entry point must hand back the same hit records for a batch of pairs: mm_map_frag on two segments with the pe_ori
flips on one side, mm_gpu_map_batch with n_seg == 2 (host pipeline + oracle-backed check backend) on the other."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import reflib  # noqa: E402
import synth  # noqa: E402

CHECK_SO = os.path.join(HERE, "_build", "libmm2amd_check.so")


@pytest.mark.skipif(not (os.path.exists(reflib.REFDRV_SO) and os.path.exists(CHECK_SO)), reason="needs oracle/_ref and tests/_build (dev container)")
def test_pair_driver_equals_map_batch(tmp_path):
    import minimap2_amd as mm
    from minimap2_amd import shard
    ref, f1, f2, _ = synth.make_pairs(str(tmp_path), n_pairs=150)

    def fasta(path):
        names, seqs = [], []
        for line in open(path, "rb"):
            (names if line.startswith(b">") else seqs).append(line.strip().lstrip(b">"))
        return names, seqs

    rn, rs = fasta(ref)
    n1, s1 = fasta(f1)
    _, s2 = fasta(f2)
    pairs = [(nm[:-2], a, b) for nm, a, b in zip(n1, s1, s2)]
    D = C.CDLL(reflib.REFDRV_SO)
    io, mo = mm.IdxOpt(), mm.MapOpt()
    D.mm_set_opt(None, C.byref(io), C.byref(mo))
    assert D.mm_set_opt(b"sr", C.byref(io), C.byref(mo)) == 0
    mo.flag |= mm.F_CIGAR | mm.F_OUT_SAM
    D.mm_idx_str.restype = C.c_void_p
    D.mm_idx_str.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)]
    mi = D.mm_idx_str(io.w, io.k, 0, 14, len(rs), (C.c_char_p * len(rs))(*rs), (C.c_char_p * len(rs))(*rn))
    D.mm_mapopt_update.argtypes = [C.c_void_p, C.c_void_p]
    D.mm_mapopt_update(C.byref(mo), mi)
    n = len(pairs)
    D.refdrv_map_pairs.restype = C.c_double
    D.refdrv_map_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.c_int,
                                   C.POINTER(C.c_int), C.POINTER(C.c_void_p)]
    flat = [s for p in pairs for s in p[1:]]
    nr, rg = (C.c_int * (2 * n))(), (C.c_void_p * (2 * n))()
    D.refdrv_map_pairs(mi, C.byref(mo), n, (C.c_char_p * (2 * n))(*flat), (C.c_int * (2 * n))(*[len(s) for s in flat]),
                       (C.c_char_p * n)(*[p[0] for p in pairs]), 4, nr, rg)
    K = mm.lib(CHECK_SO)
    assert K.mm_gpu_init(mi, C.byref(mo), 4) == 0, K.mm2amd_last_error()
    arr = (mm.Bseq1 * (2 * n))()
    for k, s in enumerate(flat):
        arr[k].l_seq, arr[k].rid, arr[k].name, arr[k].seq = len(s), k, pairs[k // 2][0], s
    n_reg, reg = (C.c_int * (2 * n))(), (C.c_void_p * (2 * n))()
    rep, gap = (C.c_int * (2 * n))(), (C.c_int * (2 * n))()
    rc = K.mm_gpu_map_batch(n, (C.c_int * n)(*range(0, 2 * n, 2)), (C.c_int * n)(*([2] * n)), arr, n_reg, reg, rep, gap)
    assert rc == 0, K.mm2amd_last_error()
    want = shard.pack_hits(K, nr, rg).numpy().tobytes()
    got = shard.pack_hits(K, n_reg, reg).numpy().tobytes()
    assert sum(nr) > n and list(nr) == list(n_reg)
    assert want == got
    K.mm2amd_free_regs(2 * n, n_reg, reg)
    K.mm2amd_free_regs(2 * n, nr, rg)
    K.mm_gpu_destroy()
    D.mm_idx_destroy.argtypes = [C.c_void_p]
    D.mm_idx_destroy(mi)
