"""Device-side index construction and the mappy-shaped Python front end, on the GPU, against the compiled reference:
  * the flat tables built by index_build.hip must describe exactly the (minimizer -> ascending positions) map of the
    reference's mm_idx_t built by mm_idx_str from the same sequences (including N runs, tandem repeats, tiny contigs);
  * Aligner.map_batch must return the hits mm_map returns, field by field and CIGAR by CIGAR."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import reflib  # noqa: E402
import synth  # noqa: E402

pytestmark = pytest.mark.gpu


def _tricky_reference(rng, total):
    """uniform sequence seasoned with N runs (some right at 2048-base chunk borders), dinucleotide/homopolymer repeats and tiny contigs"""
    c0 = synth.ACGT[rng.integers(0, 4, total, dtype=np.uint8)].copy()
    for _ in range(40):
        p = int(rng.integers(0, total - 3000))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            c0[p:p + int(rng.integers(1, 60))] = ord("N")
        elif kind == 1:
            c0[p:p + 400] = np.frombuffer((b"AC" * 200), dtype=np.uint8)
        elif kind == 2:
            c0[p:p + 300] = ord("A")
        else:
            q = (p // 2048) * 2048 + int(rng.integers(-30, 30))
            if 0 < q < total - 100:
                c0[q:q + int(rng.integers(1, 25))] = ord("N")
                c0[q + 30:q + 230] = np.frombuffer((b"GAT" * 67)[:200], dtype=np.uint8)
    seqs = [c0.tobytes(), b"ACGTTGCA", b"N" * 50, synth.ACGT[rng.integers(0, 4, 5000, dtype=np.uint8)].tobytes(), b"ACGTACGTAGCTAGCTAGCTAGCATGCATGCATCGATCGATCGACTAGCTAGCTAGCTAC"]
    return seqs


def _ref_index_pairs(seqs, w, k, hpc=0):
    """(hash, pos) pairs of the reference's index, via its public mm_idx_get over every minimizer mm_sketch reports"""
    R = C.CDLL(reflib.REF_SO)
    R.mm_idx_str.restype = C.c_void_p
    R.mm_idx_str.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)]
    n = len(seqs)
    mi = R.mm_idx_str(w, k, hpc, 14, n, (C.c_char_p * n)(*seqs), None)
    R.mm_idx_get.restype = C.POINTER(C.c_uint64)
    R.mm_idx_get.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_int)]
    out = {}
    for rid, s in enumerate(seqs):
        mz = reflib.ref_sketch(s, w, k, rid, hpc)
        for h in np.unique(mz[:, 0] >> np.uint64(8)):
            cnt = C.c_int(0)
            p = R.mm_idx_get(mi, int(h), C.byref(cnt))
            out[int(h)] = [p[i] for i in range(cnt.value)]
    R.mm_idx_destroy.argtypes = [C.c_void_p]
    R.mm_idx_destroy(mi)
    return out


@pytest.mark.parametrize("w,k,hpc", [(10, 15, 0), (19, 19, 0), (5, 15, 0), (40, 21, 0), (10, 19, 1), (10, 15, 1), (40, 21, 1)])
def test_device_index_equals_reference_index(w, k, hpc):
    import minimap2_amd as mm
    rng = np.random.default_rng(100 + w)
    seqs = _tricky_reference(rng, 300000)
    L = mm.lib()
    n = len(seqs)
    idx = L.mm2amd_idx_str(w, k, hpc, 14, n, (C.c_char_p * n)(*seqs), None)
    assert idx, L.mm2amd_last_error()
    nd, nm, sl = C.c_uint64(), C.c_uint64(), C.c_uint64()
    assert L.mm2amd_idx_stat(idx, None, None, None, None, sl, nd, nm) == 0
    keys = np.zeros(nd.value, np.uint64)
    val_off = np.zeros(nd.value + 1, np.uint32)
    pos = np.zeros(nm.value, np.uint64)
    S = np.zeros((sl.value + 7) // 8, np.uint32)
    assert L.mm2amd_idx_export(idx, None, keys.ctypes.data, val_off.ctypes.data, pos.ctypes.data, S.ctypes.data) == 0
    want = _ref_index_pairs(seqs, w, k, hpc)
    assert len(keys) == len(want)
    assert np.all(keys[1:] > keys[:-1])
    for i, h in enumerate(keys.tolist()):
        assert pos[val_off[i]:val_off[i + 1]].tolist() == want[h], (w, k, hex(h))
    # packed sequence == the reference's 4-bit layout (mmpriv.h:34-35)
    cat = np.frombuffer(b"".join(seqs), dtype=np.uint8)
    lut = np.full(256, 4, np.uint8)
    for ch, v in zip(b"ACGTacgt", [0, 1, 2, 3, 0, 1, 2, 3]):
        lut[ch] = v
    codes = lut[cat]
    got = (S[np.arange(len(codes)) >> 3] >> ((np.arange(len(codes)) & 7) << 2).astype(np.uint32)) & 0xf
    assert np.array_equal(got.astype(np.uint8), codes)
    # mm_idx_cal_max_occ from the histogram
    R = C.CDLL(reflib.REF_SO)
    L.mm2amd_idx_destroy(idx)


@pytest.mark.parametrize("kind,preset,n_reads,seed", [("ont", "map-ont", 60, 31), ("hifi", "map-hifi", 30, 32), ("hifi", "lr:hq", 20, 33), ("hifi", "map-pb", 20, 34)])
def test_aligner_equals_mm_map(kind, preset, n_reads, seed):
    import minimap2_amd as mm
    rng = np.random.default_rng(seed)
    contigs = synth.gen_reference(rng, 3000000, 3)
    mean, sd, err = synth.PROFILES[kind]
    reads = synth.gen_reads(rng, contigs, n_reads, mean, sd, err)
    refs = [synth.ACGT[c].tobytes() for c in contigs]
    rds = [("read%d" % i, synth.ACGT[r].tobytes()) for i, r in enumerate(reads)]
    rds += [("empty", b""), ("short", refs[0][100:130]), ("withN", refs[1][5000:5600] + b"NNNNNNNNNNNN" + refs[1][5612:8000])]
    # tandem duplications inside the read: the same minimizers at two query positions hit the same reference positions, i.e.
    # anchors with equal sort keys -- the case where the reference's unstable sort order is observable
    for t in range(6):
        p0 = 20000 + 7000 * t
        rds.append(("dup%d" % t, refs[2][p0:p0 + 3000] + refs[2][p0 + 2000:p0 + 3000] * (1 + t % 3) + refs[2][p0 + 3000:p0 + 6000]))
        rds.append(("sdup%d" % t, refs[1][p0:p0 + 2500] + refs[1][p0 + 2350:p0 + 2500] + refs[1][p0 + 2500:p0 + 5000]))  # few duplicated keys
    # structural differences inside the read: the gap-fill alignments across them trip the Z-drop test (re-alignment, region
    # split, inversion rescue: align.c:843-868, :916-971)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    for t in range(10):
        p0 = 100000 + 9000 * t
        r0 = refs[0]
        junk = synth.ACGT[rng.integers(0, 4, 150 + 60 * t, dtype=np.uint8)].tobytes()
        rds.append(("ins%d" % t, r0[p0:p0 + 3000] + junk + r0[p0 + 3000:p0 + 6000]))
        rds.append(("del%d" % t, r0[p0:p0 + 3000] + r0[p0 + 3200 + 80 * t:p0 + 6500]))
        rds.append(("inv%d" % t, r0[p0:p0 + 3000] + r0[p0 + 3000:p0 + 3400 + 100 * t].translate(comp)[::-1] + r0[p0 + 3400 + 100 * t:p0 + 7000]))
    names = ["chr%d" % (i + 1) for i in range(3)]
    al = mm.Aligner(refs, preset=preset, names=names, n_threads=8)
    st = al.index_stat()
    got = al.map_batch(rds)
    al.close()
    ref = reflib.RefMapper(refs, preset, names)
    assert st["n_seq"] == 3 and st["sum_len"] == sum(len(r) for r in refs)
    assert al.map_opt.mid_occ == ref.mo.mid_occ
    want = [ref.map(nm, s) for nm, s in rds]
    ref.close()
    assert sum(1 for h in got if h) >= n_reads - 1
    for i in range(len(rds)):
        assert [a.key() for a in got[i]] == want[i], (rds[i][0], len(rds[i][1]))


def test_profile_counters_and_substeps():
    import minimap2_amd as mm
    rng = np.random.default_rng(41)
    contigs = synth.gen_reference(rng, 2000000, 2)
    reads = synth.gen_reads(rng, contigs, 40, 8000, 1000, 0.12)
    refs = [synth.ACGT[c].tobytes() for c in contigs]
    rds = [("read%d" % i, synth.ACGT[r].tobytes()) for i, r in enumerate(reads)]
    al = mm.Aligner(refs, preset="map-ont", n_threads=8)
    os.environ["MM2AMD_DEVICE_FINISH"] = "0"  # the regions' last step (mm_update_extra) on the host ...
    try:
        plain = [[a.key() for a in h] for h in al.map_batch(rds)]
    finally:
        del os.environ["MM2AMD_DEVICE_FINISH"]
    os.environ["MM2AMD_DEVICE_FINISH"] = "1"  # ... and on the device: the same hits
    try:
        mm.profile_enable(True)
        whole = [[a.key() for a in h] for h in al.map_batch(rds)]
        prof = mm.profile_get()
        mm.profile_enable(False)
    finally:
        del os.environ["MM2AMD_DEVICE_FINISH"]
    assert whole == plain
    assert any(k.startswith("ksw_ext") for k in prof) and any(k.startswith("ksw_band_kernel") for k in prof) and "chain_fill_kernel" in prof and "region_finish_kernel" in prof
    assert any(k.startswith("ksw_stream_kernel") for k in prof)  # (the launch that takes the banded kernel's rejects)
    counts = {k: v for k, v in prof.items() if v["launches"] == 0}  # counts that go with a kernel without being a launch: the banded kernel's computed cells
    assert all(k.startswith("band_cells_computed") and v["units"] > 0 for k, v in counts.items()) and counts
    assert all(v["ms"] > 0 for v in prof.values() if v["launches"] >= 1)
    os.environ["MM2AMD_SUBBATCH_BASES"] = "50000"  # several sub-batches must give the same answer as one
    try:
        parts = [[a.key() for a in h] for h in al.map_batch(rds)]
    finally:
        del os.environ["MM2AMD_SUBBATCH_BASES"]
    al.close()
    assert parts == whole


def test_long_join_rechain_on_the_device():
    """map.c:283-292 on the GPU: a repeat-bearing reference (every 6 kb stretch exists twice) gives every read more than one chain; the
    re-chaining runs on the device (rechain_gather_kernel, anchor_sort_kernel, chain_rmq_kernel, chain_backtrack_kernel), the share is in
    mm2amd_last_stats, and the hits equal the compiled reference's and the host tree's (MM2AMD_LONG_JOIN_ON_HOST=1)"""
    import minimap2_amd as mm
    rng = np.random.default_rng(56)
    contig = synth.gen_duplicated_reference(rng)
    refs = [synth.ACGT[contig].tobytes(), synth.ACGT[rng.integers(0, 4, 200000, dtype=np.uint8)].tobytes()]
    reads = synth.gen_reads(rng, [contig], 200, 9000, 3000, 0.08)
    rds = [("rep%d" % i, synth.ACGT[r].tobytes()) for i, r in enumerate(reads)]
    al = mm.Aligner(refs, preset="map-ont", n_threads=8)
    try:
        mm.profile_enable(True)
        got = [[a.key() for a in h] for h in al.map_batch(rds)]
        prof = mm.profile_get()
        mm.profile_enable(False)
        st = al.last_stats()
        os.environ["MM2AMD_LONG_JOIN_ON_HOST"] = "1"
        try:
            on_host = [[a.key() for a in h] for h in al.map_batch(rds)]
            st_host = al.last_stats()
        finally:
            del os.environ["MM2AMD_LONG_JOIN_ON_HOST"]
    finally:
        al.close()
    assert st["n_long_join_dev"] + st["n_long_join_host"] >= 0.1 * len(rds) and st["n_long_join_dev"] > 0
    assert st_host["n_long_join_dev"] == 0 and st_host["n_long_join_host"] == st["n_long_join_dev"] + st["n_long_join_host"]
    assert "chain_rmq_kernel[long-join]" in prof and "rechain_gather_kernel" in prof
    assert on_host == got
    if os.path.exists(reflib.REF_SO):
        assert got == reflib.ref_map_reads(refs, rds, "map-ont")


def test_two_bucket_partition_closed_form(tmp_path):
    """anchor_sort_kernel's replay of the reference's unstable radix sort: partitions with exactly two buckets (the strand bit at the top) by
    the block-wide closed form instead of the one-thread walk; reads with a tandem duplication (pairs of equal anchor keys): the anchors in
    order (MM2AMD_SEED_DUMP) equal with and without it (MM2AMD_NO_TWO_BUCKET=1), the hits equal the compiled reference's"""
    import minimap2_amd as mm
    rng = np.random.default_rng(92)
    contigs = synth.gen_reference(rng, 2000000, 2)
    reads = synth.gen_tandem_reads(rng, contigs, 150, 9000, 0.08)
    refs = [synth.ACGT[c].tobytes() for c in contigs]
    rds = [("tan%d" % i, synth.ACGT[r].tobytes()) for i, r in enumerate(reads)]
    dumps = []
    for k, env in enumerate(({}, {"MM2AMD_NO_TWO_BUCKET": "1"})):
        dump = str(tmp_path / ("seeds%d.txt" % k))
        os.environ["MM2AMD_SEED_DUMP"] = dump
        os.environ.update(env)
        try:
            al = mm.Aligner(refs, preset="map-ont", names=["chr1", "chr2"], n_threads=8)
            got = [[a.key() for a in h] for h in al.map_batch(rds)]
            al.close()
        finally:
            del os.environ["MM2AMD_SEED_DUMP"]
            for e in env:
                del os.environ[e]
        dumps.append((open(dump).read(), got))
    assert dumps[0][0] == dumps[1][0] and dumps[0][0].count("SD\t") > 10000
    assert dumps[0][1] == dumps[1][1]
    if os.path.exists(reflib.REF_SO):
        assert dumps[0][1] == reflib.ref_map_reads(refs, rds, "map-ont")


def test_repeat_rich_reference_ties_and_long_anchor_lists():
    """a reference with many diverged copies of one element: thousands of anchors per read, equal-x ties, high-occurrence seeds"""
    import minimap2_amd as mm
    rng = np.random.default_rng(55)
    elem = rng.integers(0, 4, 1500, dtype=np.uint8)
    parts = []
    for c in range(60):
        e = elem.copy()
        mut = rng.random(len(e)) < 0.03
        e[mut] = (e[mut] + rng.integers(1, 4, int(mut.sum()), dtype=np.uint8)) % 4
        parts += [rng.integers(0, 4, int(rng.integers(200, 3000)), dtype=np.uint8), e]
    contig = np.concatenate(parts)
    refs = [synth.ACGT[contig].tobytes(), synth.ACGT[rng.integers(0, 4, 200000, dtype=np.uint8)].tobytes()]
    reads = synth.gen_reads(rng, [contig], 40, 6000, 2000, 0.08)
    rds = [("rep%d" % i, synth.ACGT[r].tobytes()) for i, r in enumerate(reads)]
    al = mm.Aligner(refs, preset="map-ont", n_threads=8)
    got = al.map_batch(rds)
    al.close()
    ref = reflib.RefMapper(refs, "map-ont")
    want = [ref.map(nm, s) for nm, s in rds]
    ref.close()
    for i in range(len(rds)):
        assert [a.key() for a in got[i]] == want[i], rds[i][0]


def test_tandem_array_many_anchors_with_ties():
    """reads across a tandem array of 8 near-identical copies: > 8192 anchors per read (the replay leaves LDS) and many equal keys"""
    import minimap2_amd as mm
    rng = np.random.default_rng(56)
    elem = rng.integers(0, 4, 1500, dtype=np.uint8)
    copies = []
    for c in range(8):
        e = elem.copy()
        mut = rng.random(len(e)) < 0.01
        e[mut] = (e[mut] + rng.integers(1, 4, int(mut.sum()), dtype=np.uint8)) % 4
        copies.append(e)
    contig = np.concatenate([rng.integers(0, 4, 30000, dtype=np.uint8)] + copies + [rng.integers(0, 4, 30000, dtype=np.uint8)])
    refs = [synth.ACGT[contig].tobytes(), synth.ACGT[rng.integers(0, 4, 100000, dtype=np.uint8)].tobytes()]
    rds = []
    for i in range(12):
        st = 24000 + 500 * i
        r = synth.mutate_read(rng, contig[st:st + 16000 + 300 * i], 0.03)
        if i % 2:
            r = synth.COMP[r[::-1]]
        rds.append(("arr%d" % i, synth.ACGT[r].tobytes()))
    al = mm.Aligner(refs, preset="map-ont", n_threads=8)
    got = al.map_batch(rds)
    al.close()
    ref = reflib.RefMapper(refs, "map-ont")
    want = [ref.map(nm, s) for nm, s in rds]
    ref.close()
    for i in range(len(rds)):
        assert [a.key() for a in got[i]] == want[i], rds[i][0]
