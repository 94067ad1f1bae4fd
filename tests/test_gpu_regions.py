"""The device's chains -> hits -> DP windows -> consume path (region_dev.hip, Backend::align_regions) against the compiled reference:
  * plain long reads are finished on the device (mm2amd_last_stats says so) and equal mm_map's hits field by field, CIGAR by CIGAR;
  * reads the device cannot decide alone -- a deleted / inserted / inverted stretch makes a gap fill trip the Z-drop test, so the region needs a second
    DP round, a split, maybe the inversion rescue -- are handed back to the host path and still equal the reference;
  * repeat-rich references (many chains per read, secondaries, equal scores) and every long-read preset;
  * MM2AMD_DEVICE_REGIONS=0 (the host's chains -> hits, planning and consumption) gives the same records."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import reflib  # noqa: E402
import synth  # noqa: E402

pytestmark = pytest.mark.gpu


def _codes_to_reads(reads):
    return [("read%d" % i, synth.ACGT[r].tobytes()) for i, r in enumerate(reads)]


def _map(refs, rds, preset, names):
    import minimap2_amd as mm
    al = mm.Aligner(refs, preset=preset, names=names, n_threads=4)
    hits = al.map_batch(rds)
    st = al.last_stats()
    al.close()
    return [[a.key() for a in h] for h in hits], st


def _sv_reads(rng, contigs, n, err):
    """reads with one rearranged stretch each: a deletion, an insertion of random bases, an inverted segment, or a tandem duplication"""
    out = []
    for i in range(n):
        c = contigs[int(rng.integers(0, len(contigs)))]
        L = int(rng.integers(4000, 9000))
        st = int(rng.integers(0, len(c) - L))
        s = c[st:st + L].copy()
        p = int(rng.integers(1200, L - 2400))
        w = int(rng.integers(300, 1200))
        kind = i % 4
        if kind == 0:
            s = np.concatenate([s[:p], s[p + w:]])
        elif kind == 1:
            s = np.concatenate([s[:p], rng.integers(0, 4, w, dtype=np.uint8), s[p:]])
        elif kind == 2:
            s = np.concatenate([s[:p], synth.COMP[s[p:p + w][::-1]], s[p + w:]])
        else:
            s = np.concatenate([s[:p + w], s[p:p + w], s[p + w:]])
        if rng.random() < 0.5:
            s = synth.COMP[s[::-1]]
        out.append(synth.mutate_read(rng, s, err))
    return out


@pytest.mark.parametrize("preset,err", [("map-ont", 0.12), ("map-hifi", 0.005), ("lr:hq", 0.02), ("asm20", 0.03)])
def test_plain_reads_are_finished_on_the_device(preset, err):
    rng = np.random.default_rng(31)
    contigs = synth.gen_reference(rng, 1500000, 3)
    reads = synth.gen_reads(rng, contigs, 48, 5000, 1500, err)
    refs, names = [synth.ACGT[c].tobytes() for c in contigs], ["chr1", "chr2", "chr3"]
    rds = _codes_to_reads(reads)
    got, st = _map(refs, rds, preset, names)
    assert got == reflib.ref_map_reads(refs, rds, preset, names=names)
    assert st["n_region_reads_dev"] + st["n_region_reads_host"] == len(rds)
    assert st["n_region_reads_dev"] >= 0.9 * len(rds), st  # (a read may still trip a Z-drop test by chance)


@pytest.mark.parametrize("preset,err", [("map-ont", 0.08), ("map-hifi", 0.005)])
def test_rearranged_reads_are_handed_back_and_identical(preset, err):
    rng = np.random.default_rng(32)
    contigs = synth.gen_reference(rng, 1000000, 2)
    reads = _sv_reads(rng, contigs, 48, err) + synth.gen_reads(rng, contigs, 16, 4000, 1000, err)
    refs, names = [synth.ACGT[c].tobytes() for c in contigs], ["chr1", "chr2"]
    rds = _codes_to_reads(reads)
    got, st = _map(refs, rds, preset, names)
    assert got == reflib.ref_map_reads(refs, rds, preset, names=names)
    assert st["n_region_reads_host"] >= 8, st   # the rearrangements need the host's rounds ...
    assert st["n_region_reads_dev"] >= 16, st   # ... the plain reads beside them in the same sub-batch do not


def test_rmq_reads_handed_back_by_the_chaining_kernel_keep_their_minimizers(monkeypatch):
    """with the chains left on the device, a read chain_rmq_kernel hands back (more anchors than one wavefront should walk -- a whole contig
    against its reference -- a tied range minimum, an over-full neighbourhood) is chained and finished by the host path, whose mm_est_err walks the read's
    minimizer positions: they must come to the host with the read's anchors.  MM2AMD_RMQ_DEV_MAX_ANCHORS lowers the kernel's limit so that every read is one."""
    monkeypatch.setenv("MM2AMD_RMQ_DEV_MAX_ANCHORS", "50")
    rng = np.random.default_rng(36)
    contigs = synth.gen_reference(rng, 900000, 2)
    reads = synth.gen_reads(rng, contigs, 24, 6000, 1500, 0.03)
    refs, names = [synth.ACGT[c].tobytes() for c in contigs], ["chr1", "chr2"]
    rds = _codes_to_reads(reads)
    got, st = _map(refs, rds, "asm20", names)
    assert got == reflib.ref_map_reads(refs, rds, "asm20", names=names)
    assert st["n_region_reads_host"] >= len(rds) - 2, st  # the hand-backs went through the host path
    assert all(len(h) >= 1 for h in got)


def test_repeat_rich_reference_many_chains():
    """a reference with multi-copy segments: reads get several chains, primaries with secondaries (mm_set_parent / mm_select_sub on the device),
    equal-scoring copies"""
    rng = np.random.default_rng(33)
    base = rng.integers(0, 4, 600000, dtype=np.uint8)
    unit = rng.integers(0, 4, 6000, dtype=np.uint8)
    for k in range(12):  # twelve copies of one 6 kb segment, 1-3 % diverged
        p = 20000 + k * 45000
        cp = unit.copy()
        m = rng.random(len(cp)) < 0.01 * (1 + k % 3)
        cp[m] = (cp[m] + rng.integers(1, 4, int(m.sum()), dtype=np.uint8)) % 4
        base[p:p + len(cp)] = cp
    contigs = [base]
    reads = synth.gen_reads(rng, contigs, 40, 5000, 1500, 0.06)
    for k in range(12):  # reads lying mostly inside a copy
        p = 20000 + k * 45000 + int(rng.integers(-1500, 1500))
        reads.append(synth.mutate_read(rng, base[p:p + 5000], 0.06))
    refs, names = [synth.ACGT[base].tobytes()], ["chr1"]
    rds = _codes_to_reads(reads)
    got, st = _map(refs, rds, "map-ont", names)
    assert got == reflib.ref_map_reads(refs, rds, "map-ont", names=names)
    assert max(len(h) for h in got) >= 2  # secondaries were kept
    assert st["n_region_reads_dev"] >= len(rds) // 2, st


def test_host_regions_switch_gives_the_same_records(monkeypatch):
    rng = np.random.default_rng(34)
    contigs = synth.gen_reference(rng, 800000, 2)
    reads = synth.gen_reads(rng, contigs, 24, 4000, 1000, 0.1) + _sv_reads(rng, contigs, 8, 0.1)
    refs, names = [synth.ACGT[c].tobytes() for c in contigs], ["chr1", "chr2"]
    rds = _codes_to_reads(reads)
    a, st_a = _map(refs, rds, "map-ont", names)
    monkeypatch.setenv("MM2AMD_DEVICE_REGIONS", "0")
    b, st_b = _map(refs, rds, "map-ont", names)
    assert a == b
    assert st_a["n_region_reads_dev"] > 0 and st_b["n_region_reads_dev"] == 0


def test_unsupported_configurations_stay_on_the_host():
    """spliced alignment is outside the device path's rules: every read goes through the host path, results as the reference's"""
    rng = np.random.default_rng(35)
    contigs = synth.gen_reference(rng, 600000, 1)
    reads = synth.gen_transcripts(rng, contigs, 12, 0.03)
    refs, names = [synth.ACGT[c].tobytes() for c in contigs], ["chr1"]
    rds = _codes_to_reads(reads)
    got, st = _map(refs, rds, "splice", names)
    assert got == reflib.ref_map_reads(refs, rds, "splice", names=names)
    assert st["n_region_reads_dev"] == 0


def test_long_reads_beyond_the_finish_kernels_lds_are_handed_back():
    """a 60 kb read's stitched CIGAR has more operations than region_finish_kernel stages in LDS (MM2AMD_FIN_MAX_OPS): the device hands the read back
    and the host path finishes it; shorter reads beside it stay on the device"""
    rng = np.random.default_rng(36)
    contigs = synth.gen_reference(rng, 1200000, 2)
    reads = synth.gen_reads(rng, contigs, 3, 60000, 2000, 0.1, min_len=50000) + synth.gen_reads(rng, contigs, 12, 4000, 1000, 0.1)
    refs, names = [synth.ACGT[c].tobytes() for c in contigs], ["chr1", "chr2"]
    rds = _codes_to_reads(reads)
    got, st = _map(refs, rds, "map-ont", names)
    assert got == reflib.ref_map_reads(refs, rds, "map-ont", names=names)
    assert st["n_region_reads_host"] >= 2 and st["n_region_reads_dev"] >= 10, st


def test_homopolymer_compressed_index_on_the_device():
    """map-pb (an HPC index): the window boundaries sit at homopolymer-run starts (mm_adjust_minier, align.c:418-428) and mm_est_err's mean span is a sum
    over the read's minimizers -- both on the device since round 5; the reads must be finished there and equal the reference's hits"""
    rng = np.random.default_rng(37)
    contigs = synth.gen_reference(rng, 1000000, 2)
    for c in contigs:  # homopolymer runs, so that compressed and plain coordinates differ
        for p in rng.integers(0, len(c) - 20, 4000):
            c[p:p + int(rng.integers(2, 9))] = c[p]
    reads = synth.gen_reads(rng, contigs, 40, 5000, 1500, 0.1)
    refs, names = [synth.ACGT[c].tobytes() for c in contigs], ["chr1", "chr2"]
    rds = _codes_to_reads(reads)
    got, st = _map(refs, rds, "map-pb", names)
    assert got == reflib.ref_map_reads(refs, rds, "map-pb", names=names)
    assert st["n_region_reads_dev"] >= 0.8 * len(rds), st
